"""256x256 vs 256x128 persistent NT tiles on shapes whose 256-wide tile count quantises badly (run once per
NVIT_GEMM_NT_TILE setting: the library reads the variable once)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops
from gemm_bench import bench

dev = "cuda:0"
for (M, N, K, name) in [(50176, 1024, 1024, "large o"), (50176, 1024, 4096, "large p"), (50176, 1024, 3072, "large dqkv"),
                        (50176, 1024, 8192, "large dfc"), (100352, 768, 768, "base o"), (100352, 768, 3072, "base p"),
                        (50176, 768, 768, "base64 o"), (50176, 768, 3072, "base64 p")]:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    res = []
    for dt in (torch.bfloat16, torch.float32):
        out = torch.empty(M, N, device=dev, dtype=dt)
        ms, tf = bench(lambda: ops.gemm_nt(A, B, M, N, K, out=out), 2.0 * M * N * K, iters=20)
        res.append((ms, tf))
    print(f"{name:12s} M={M} N={N} K={K}: bf16-out {res[0][0]:.3f} ms {res[0][1]:7.1f} TF/s | f32-out {res[1][0]:.3f} ms {res[1][1]:7.1f} TF/s", flush=True)
