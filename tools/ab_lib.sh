# interleaved A/B of builds of the library on the default bench (GPU box):  bash tools/ab_lib.sh
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms_per_step']; print('$tag', d['value'], d['ms_per_step'], 'nt', k['gemm_nt'], 'rowops', k['rowops'], 'tn', k['gemm_tn'])"; }
for r in 1 2; do
run product A=1
for v in 1 2 3; do run variant$v NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.var$v; done
done
