# interleaved A/B of library variants on tools/attn_bench.py (GPU box):  bash tools/attn_ab.sh tag1[:ENV=V] tag2 ...  ("product" = the product build)
mkdir -p gpurun_out
for r in 1 2; do
for spec in "$@"; do
  t=${spec%%:*}; envs=""; [ "$spec" != "$t" ] && envs=${spec#*:}
  echo "== $spec (round $r)"
  if [ "$t" = product ]; then env $envs python tools/attn_bench.py 2>/dev/null | grep -E "fwd bounded, q|bwd fused,"; else env $envs NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.$t python tools/attn_bench.py 2>/dev/null | grep -E "fwd bounded, q|bwd fused,"; fi
done
done
