import sys, os
sys.path.insert(0, os.getcwd())
import torch
from nvit_amd import ops
dev = "cuda:0"
M = 100352
for N, K in ((768, 6144), (768, 2304)):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    out = torch.zeros(M, N, device=dev)
    for acc in (False, True):
        fn = lambda: ops.gemm_nt(A, W, M, N, K, out=out, accumulate=acc)
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        print(f"N={N} K={K} accumulate={acc}: {min(ts):7.1f} us  {2.0 * M * N * K / min(ts) / 1e6:7.1f} TF/s")
