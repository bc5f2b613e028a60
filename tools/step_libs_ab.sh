# interleaved A/B of the default bench between the product library and variant builds nvit_amd/libnvit_hip.so.<tag> (GPU box):
#   bash tools/step_libs_ab.sh tag1 tag2 ...      prints ms/step and the rowops / gemm_nt / attn_bwd / gemm_tn families
one() {
  python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=j['kernel_ms_per_step']; print('$1', j['ms_per_step'], 'rowops', k['rowops'], 'gemm_nt', k['gemm_nt'], 'attn_bwd', k['attn_bwd'], 'gemm_tn', k['gemm_tn'])"
}
for r in 1 2 3; do
  one product
  for t in "$@"; do NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.$t one $t; done
done
