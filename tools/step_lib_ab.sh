# interleaved A/B of the default bench between the product library and nvit_amd/libnvit_hip.so.olds (GPU box): ms/step, rowops, gemm_nt
for r in 1 2 3; do
  python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('new', j['ms_per_step'], j['kernel_ms_per_step']['rowops'], j['kernel_ms_per_step']['gemm_nt'])"
  NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.olds python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('old', j['ms_per_step'], j['kernel_ms_per_step']['rowops'], j['kernel_ms_per_step']['gemm_nt'])"
done
