# interleaved A/B of the GEMM micro-benchmarks between the product library and nvit_amd/libnvit_hip.so.olds (GPU box)
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_fullshape.py -m gpu -x -q -k "gemm or tn or wgrad or fullshape" > gpurun_out/gs_tests3.log 2>&1 || exit 1
for r in 1 2 3; do
  echo "== new $r"; timeout -k 10 100 python3 tools/gemm_bench.py 2>/dev/null | grep -E "NT|TN"; timeout -k 10 100 python3 tools/gemm_fused_bench.py 2>/dev/null | grep EPI
  echo "== old $r"; NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.olds timeout -k 10 100 python3 tools/gemm_bench.py 2>/dev/null | grep -E "NT|TN"; NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.olds timeout -k 10 100 python3 tools/gemm_fused_bench.py 2>/dev/null | grep EPI
done > gpurun_out/gs_ab3.log 2>&1
tail -2 gpurun_out/gs_tests3.log; cat gpurun_out/gs_ab3.log
