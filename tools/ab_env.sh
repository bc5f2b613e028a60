# interleaved A/B of environment switches on the default bench (GPU box):  bash tools/ab_env.sh
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms_per_step']; print('$tag', d['value'], d['ms_per_step'], 'nt', k['gemm_nt'], 'rowops', k['rowops'], 'tn', k['gemm_tn'])"; }
for r in 1 2 3; do
run tn_batch A=1
run tn_each NVIT_TN_BATCH=0
done
