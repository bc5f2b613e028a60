# interleaved A/B of builds of the library on the default bench (GPU box):  bash tools/ab_lib.sh [variant tags...]
# (variants: make -C nvit_amd/csrc BUILD=build_var OUT=../libnvit_hip.so.<tag> EXTRA="...")
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms_per_step']; print('$tag', d['value'], d['ms_per_step'], {n: k[n] for n in ('gemm_nt','gemm_tn','gemm_swiglu','gemm_swiglu_bwd','gemm_qknorm','attn_fwd','attn_bwd','rowops')})"; }
for r in 1 2; do
run product A=1
for v in "$@"; do run $v NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.$v; done
done
