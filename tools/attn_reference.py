"""Known-good reference for the attention kernels on the same hardware: torch's scaled_dot_product_attention (the
flash / memory-efficient kernels the ROCm build of PyTorch ships) at the benchmarked shape, forward and backward, bf16,
next to the hand-written kernels (tools/attn_bench.py numbers).  python tools/attn_reference.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
B, H, T, d = 128, 12, 784, 64
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = [torch.randn(B, H, T, d, generator=g, device=dev).bfloat16().requires_grad_(True) for _ in range(3)]
do = torch.randn(B, H, T, d, generator=g, device=dev).bfloat16()

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

from torch.nn.attention import SDPBackend, sdpa_kernel
fl = B * H * T * T * d
for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION)):
    try:
        with sdpa_kernel(be):
            fwd = lambda: F.scaled_dot_product_attention(q, k, v, scale=1.0 / math.sqrt(d))
            t_f = min(timed(fwd) for _ in range(3))
            o = fwd()
            bwd = lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True)
            t_b = min(timed(bwd) for _ in range(3))
        print(f"torch SDPA {name:10s}: forward {t_f * 1e3:7.1f} us ({4 * fl / t_f / 1e9:6.1f} TF/s)   backward {t_b * 1e3:7.1f} us "
              f"({10 * fl / t_b / 1e9:6.1f} TF/s algorithmic)", flush=True)
    except Exception as e:   # a backend the build does not have
        print(f"torch SDPA {name}: not available ({type(e).__name__}: {str(e)[:100]})", flush=True)
