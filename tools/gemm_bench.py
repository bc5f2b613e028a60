"""Micro-benchmark of the MFMA GEMM kernels on the Base-config shapes (HIP-event timed)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops

def bench(fn, flops, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    return ms, flops / ms / 1e9

def main():
    M = 100352
    dev = "cuda:0"
    for (N, K, name) in [(768, 768, "o / o.Wt"), (2304, 768, "qkv"), (6144, 768, "fc"), (768, 3072, "p"),
                         (3072, 768, "p.Wt"), (768, 6144, "fc.Wt"), (768, 2304, "qkv.Wt")]:
        A = torch.randn(M, K, device=dev).bfloat16()
        B = torch.randn(N, K, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ms, tf = bench(lambda: ops.gemm_nt(A, B, M, N, K, out=out), 2.0 * M * N * K)
        out32 = torch.empty(M, N, device=dev, dtype=torch.float32)
        ms2, tf2 = bench(lambda: ops.gemm_nt(A, B, M, N, K, out=out32), 2.0 * M * N * K)
        print(f"NT {name:8s} N={N:5d} K={K:5d}: bf16-out {ms:7.3f} ms {tf:7.1f} TF/s | f32-out {ms2:7.3f} ms {tf2:7.1f} TF/s")
    for (N, K, name) in [(768, 768, "o"), (2304, 768, "qkv"), (6144, 768, "fc"), (768, 3072, "p")]:
        A = torch.randn(M, N, device=dev).bfloat16()
        B = torch.randn(M, K, device=dev).bfloat16()
        G = torch.empty(N, K, device=dev)
        ms, tf = bench(lambda: ops.gemm_tn(A, B, G, M, N, K), 2.0 * M * N * K)
        print(f"TN {name:8s} N={N:5d} K={K:5d}: {ms:7.3f} ms {tf:7.1f} TF/s  splits={ops.tn_splits(M, N, K, 1)}")


if __name__ == "__main__":
    main()
