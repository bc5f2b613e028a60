"""wgrad TN GEMM: the 2 x 64 KiB ring (mode 5, product) against the 4 x 32 KiB ring (mode 3), interleaved in one process.  python tools/tn_modes_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from nvit_amd import ops, _lib
from gemm_bench import bench
lib = _lib.load()
M = 100352; dev = "cuda:0"
for (N, K, name) in [(768, 768, "o"), (2304, 768, "qkv"), (6144, 768, "fc"), (768, 3072, "p")]:
    A = torch.randn(M, N, device=dev).bfloat16(); B = torch.randn(M, K, device=dev).bfloat16(); G = torch.empty(N, K, device=dev)
    for rnd in range(3):
        for mode in (5, 3):
            lib.nvit_set_tn_order(mode)
            ms, tf = bench(lambda: ops.gemm_tn(A, B, G, M, N, K), 2.0 * M * N * K)
            print(f"TN {name:4s} mode {mode}: {ms:.3f} ms {tf:7.1f} TF/s")
lib.nvit_set_tn_order(5)
