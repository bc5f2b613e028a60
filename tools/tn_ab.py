"""A/B of the weight-gradient GEMM variants on the four block shapes of Base B=128, interleaved rounds in one process
(guide rule 24): mode 5 = XCD-contiguous items + 2 x 64 KiB ring (round-1 ring), mode 3 = XCD-contiguous + 4 x 32 KiB
ring, mode 4 = round-robin items + 2 x 64 KiB ring (round 1).  python tools/tn_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops, _lib

dev = torch.device("cuda:0")
M, C = 100352, 768
shapes = [("qkv", 3 * C, C, 0), ("o", C, C, 0), ("fc", 8 * C, C, 1), ("p", C, 4 * C, 0)]
bufs = {}
for name, N, K, perm in shapes:
    A = (torch.randn(M, N, device=dev) * 0.05).bfloat16()
    B = torch.randn(M, K, device=dev).bfloat16()
    G = torch.empty(N, K, device=dev)
    bufs[name] = (A, B, G)
lib = _lib.load()
res = {}
MODES = (4, 5, 3)
for rnd in range(6):
    for order in MODES:
        lib.nvit_set_tn_order(order)
        for name, N, K, perm in shapes:
            A, B, G = bufs[name]
            ops.gemm_tn(A, B, G, M, N, K, perm=perm)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm_tn(A, B, G, M, N, K, perm=perm)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault((name, order), []).append(e0.elapsed_time(e1) / 5)
lib.nvit_set_tn_order(3)
tot = {m: 0.0 for m in MODES}
for name, N, K, perm in shapes:
    fl = 2.0 * M * N * K
    for order in MODES:
        t = sorted(res[(name, order)])
        med = t[len(t) // 2]
        tot[order] += med
        print(f"{name:4s} N={N:5d} K={K:5d} mode={order}: median {med * 1e3:8.1f} us  min {t[0] * 1e3:8.1f} us  "
              f"{fl / med / 1e9:7.1f} TF/s  (splits {ops.tn_splits(M, N, K, 1)})")
print("per block [ms]: " + ", ".join(f"mode {m}: {tot[m]:.3f} (x12 = {12 * tot[m]:.2f})" for m in MODES))
