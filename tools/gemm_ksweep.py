"""Per-tile cost model of the persistent NT GEMM: time vs K at a tile count that divides the CU count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops
from gemm_bench import bench  # noqa

dev = "cuda:0"
import os
CUS = int(os.environ.get("NVIT_GEMM_CUS", "256"))
TPC = int(os.environ.get("TILES_PER_CU", "3"))
M, N = 256 * CUS * TPC // 3, 768
for K in (256, 512, 768, 1536, 3072, 6144):
    A = torch.randn(M, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    res = []
    for dt in (torch.bfloat16, torch.float32):
        out = torch.empty(M, N, device=dev, dtype=dt)
        ms, tf = bench(lambda: ops.gemm_nt(A, B, M, N, K, out=out), 2.0 * M * N * K, iters=30)
        res.append((ms, tf))
    print(f"K={K:5d}: bf16-out {res[0][0]*1e3/TPC:8.2f} us/tile {res[0][1]:7.1f} TF/s | f32-out {res[1][0]*1e3/TPC:8.2f} us/tile {res[1][1]:7.1f} TF/s", flush=True)
