# interleaved A/B of environment switches on the default bench (GPU box):  bash tools/ab_env.sh
# (edit the `run` lines: each is `run <tag> VAR=value ...`; the round-3 record: gpurun_out/ab_env*.log)
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms_per_step']; print('$tag', d['value'], d['ms_per_step'], 'nt', k['gemm_nt'], 'rowops', k['rowops'], 'tn', k['gemm_tn'])"; }
for r in 1 2 3; do
run round3_default A=1
run round3_switches_off NVIT_LO_DGRAD=0 NVIT_Y_BF16=0 NVIT_PART_BLOCKS=1024
done
