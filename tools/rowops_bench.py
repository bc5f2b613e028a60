"""LERP row kernels at the benchmarked shape (M = 100 352 rows of C = 768 fp32): HIP-event times and algorithmic TB/s.
Grid sizes come from NVIT_ROW_GRID (forward) / NVIT_PART_BLOCKS (backward) when set.  python tools/rowops_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops
from nvit_amd._lib import BF16

dev = torch.device("cuda:0")
M, C = 100352, 768
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
h, y, xs, dout = rn(M, C), rn(M, C), rn(M, C), rn(M, C)
yb = y.bfloat16()
alpha = torch.full((C,), 0.05, device=dev)
skip = torch.tensor([0.9], device=dev)
dh = torch.zeros(M, C, device=dev)
add = rn(M, C).bfloat16()

def t_of(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

cases = [
    ("lerp_fwd            ", lambda: ops.lerp_fwd(BF16, h, y, alpha, 1.6), 4 + 4 + 4 + 2),
    ("lerp_fwd + norm_skip", lambda: ops.lerp_fwd(BF16, h, y, alpha, 1.6, xs, skip), 4 + 4 + 4 + 4 + 2),
    ("lerp_bwd            ", lambda: ops.lerp_bwd(BF16, dout, h, y, alpha, 1.6, None, None, None, False, False, True), 12 + 4 + 2),
    ("lerp_bwd accumulate ", lambda: ops.lerp_bwd(BF16, dout, h, y, alpha, 1.6, None, None, dh, True, False, True), 12 + 8 + 2),
    ("lerp_bwd + norm_skip", lambda: ops.lerp_bwd(BF16, dout, h, y, alpha, 1.6, xs, skip, None, False, False, True), 16 + 4 + 4 + 2),
    # the two calls of a block's backward as the step makes them (bf16 y, bf16 addend of the previous data-gradient GEMM)
    ("bwd MLP half  (addend + norm_skip) ", lambda: ops.lerp_bwd(BF16, dout, h, yb, alpha, 1.6, xs, skip, None, False, False, True, dout_add=add), 4 + 2 + 4 + 2 + 4 + 4 + 4 + 2),
    ("bwd attn half (addend + accumulate)", lambda: ops.lerp_bwd(BF16, dout, h, yb, alpha, 1.6, None, None, dh, True, False, True, dout_add=add), 4 + 2 + 4 + 2 + 4 + 4 + 2),
]
print(f"NVIT_ROW_GRID={os.environ.get('NVIT_ROW_GRID', '-')} NVIT_PART_BLOCKS={os.environ.get('NVIT_PART_BLOCKS', '-')}")
for name, fn, bpe in cases:
    us = min(t_of(fn) for _ in range(3))
    print(f"  {name}: {us:7.1f} us  {M * C * bpe / us / 1e6:6.2f} TB/s ({bpe} B/element)")
