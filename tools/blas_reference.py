"""Known-good reference for the GEMM shapes of the Base step (guide rule 10: no ceiling claims without one): the vendor
library (hipBLASLt / rocBLAS through torch.matmul) against the hand-written persistent kernels, same random bf16 data,
interleaved rounds in one process.  NT: C[M,N] = A[M,K] B[N,K]^T (bf16 out and fp32-accumulate-free);  TN: G[N,K] = A[M,N]^T B[M,K].
  python tools/blas_reference.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops

dev = torch.device("cuda:0")
M, C = 100352, 768
torch.manual_seed(0)


def timed(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("NT shapes (M = 100352): torch.matmul (vendor library) vs nvit_gemm_nt, bf16 in, bf16 out")
for name, N, K in (("o-proj / o-dgrad", C, C), ("mlp_c_proj", C, 4 * C), ("qkv dgrad", C, 3 * C), ("c_fc dgrad", C, 8 * C),
                   ("qkv", 3 * C, C), ("c_fc", 8 * C, C), ("p dgrad", 4 * C, C)):
    A = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    B = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    Bt = B.t()
    res = {"lib": [], "ours": []}
    for _ in range(4):
        res["lib"].append(timed(lambda: torch.matmul(A, Bt, out=out)))
        res["ours"].append(timed(lambda: ops.gemm_nt(A, B, M, N, K, out_dtype=torch.bfloat16)))
    fl = 2.0 * M * N * K
    l, o = sorted(res["lib"])[1], sorted(res["ours"])[1]
    print(f"  {name:18s} N={N:5d} K={K:5d}: library {l:8.1f} us {fl / l / 1e6:7.1f} TF/s | ours {o:8.1f} us {fl / o / 1e6:7.1f} TF/s | ours/library time {o / l:.3f}")
    del A, B, out

print("TN shapes (reduction over M = 100352): torch.matmul(A^T, B) fp32 out vs nvit_gemm_tn")
for name, N, K in (("o wgrad", C, C), ("qkv wgrad", 3 * C, C), ("c_fc wgrad", 8 * C, C), ("p wgrad", C, 4 * C)):
    A = (torch.rand(M, N, device=dev) * 2 - 1).bfloat16()
    B = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    G = torch.empty(N, K, device=dev)
    Gb = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
    At = A.t()
    res = {"lib": [], "ours": []}
    for _ in range(4):
        res["lib"].append(timed(lambda: torch.matmul(At, B, out=Gb)))
        res["ours"].append(timed(lambda: ops.gemm_tn(A, B, G, M, N, K)))
    fl = 2.0 * M * N * K
    l, o = sorted(res["lib"])[1], sorted(res["ours"])[1]
    print(f"  {name:18s} N={N:5d} K={K:5d}: library {l:8.1f} us {fl / l / 1e6:7.1f} TF/s | ours {o:8.1f} us {fl / o / 1e6:7.1f} TF/s | ours/library time {o / l:.3f}")
    del A, B, G, Gb
