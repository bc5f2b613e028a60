# interleaved A/B of the row kernels between the product library and a variant build (GPU box): bash tools/rowops_ab.sh <tag>
for r in 1 2 3; do
  echo "== product (round $r)"; python3 tools/rowops_bench.py
  echo "== $1 (round $r)"; NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.$1 python3 tools/rowops_bench.py
done
