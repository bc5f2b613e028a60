# interleaved A/B of the fused-epilogue GEMMs between the product library and a variant build (GPU box): bash tools/epi5_ab.sh <tag>
for r in 1 2 3; do
  echo "== product (round $r)"; python3 tools/gemm_fused_bench.py 2>/dev/null
  echo "== $1 (round $r)"; NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.$1 python3 tools/gemm_fused_bench.py 2>/dev/null
done
