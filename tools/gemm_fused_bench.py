"""Fused-epilogue GEMMs of the Base config (SwiGLU forward, SwiGLU backward, q/k normalise), HIP-event timed."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops
from gemm_bench import bench

dev = "cuda:0"
M, C = 100352, 768
x = torch.randn(M, C, device=dev).bfloat16()
wfc = (torch.randn(8 * C, C, device=dev) * 0.03).bfloat16()
suv = torch.ones(8 * C, device=dev)
ms, tf = bench(lambda: ops.gemm_nt_swiglu(x, wfc, M, 4 * C, C, suv, math.sqrt(C)), 2.0 * M * 8 * C * C, iters=20)
print(f"EPI3 fc + SwiGLU        : {ms:.3f} ms {tf:7.1f} TF/s", flush=True)
uv, xm = ops.gemm_nt_swiglu(x, wfc, M, 4 * C, C, suv, math.sqrt(C))
wpt = (torch.randn(4 * C, C, device=dev) * 0.03).bfloat16()
ms, tf = bench(lambda: ops.gemm_nt_swiglu_bwd(x, wpt, uv, M, 4 * C, C, suv, math.sqrt(C)), 2.0 * M * 4 * C * C, iters=20)
print(f"EPI5 p.Wt + SwiGLU bwd  : {ms:.3f} ms {tf:7.1f} TF/s", flush=True)
wqkv = (torch.randn(3 * C, C, device=dev) * 0.03).bfloat16()
sqk = torch.full((C,), 1 / 32, device=dev)
try:
    B_, T_, H_ = 128, 784, 12
    bufs = ops.qk_buffers(ops.dt_of(x), B_, T_, H_, 64, x.device)
    fn = lambda: ops.gemm_nt_qknorm(x, wqkv, M, C, 3, 0, sqk, 32.0, B_, T_, H_, 64, bufs=bufs)
    fn()
    ms, tf = bench(fn, 2.0 * M * 3 * C * C, iters=20)
    print(f"EPI4 qkv + q/k normalise: {ms:.3f} ms {tf:7.1f} TF/s", flush=True)
except Exception as e:   # (signature drift of the wrapper must not hide the other two lines)
    print("EPI4: skipped:", e)
