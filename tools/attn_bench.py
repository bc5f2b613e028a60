"""Attention kernels at the benchmarked shape (B*H = 1536, T = 784, d = 64): time forward (generic and bounded-score
path), backward (dq + dkv, fused epilogue variant) in interleaved rounds; print TFLOP/s.  python tools/attn_bench.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops
from nvit_amd._lib import BF16

dev = torch.device("cuda:0")
B, H, T, d = 128, 12, 784, 64
C, M = H * d, B * T
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
sqk = (1.0 / 32) * (1.0 + 0.05 * torch.tanh(rn(C)))
se = (sqk * 32.0).reshape(1, H, 1, d)
q = (se * torch.nn.functional.normalize(rn(B, H, T, d), dim=-1)).bfloat16()
k = (se * torch.nn.functional.normalize(rn(B, H, T, d), dim=-1)).bfloat16()
v = (rn(B, H, T, d) * 0.05).bfloat16()
gt = (rn(M, C) * 1e-3).bfloat16()
rq = 1.0 + rn(M, H).abs() * 0.1
rk = 1.0 + rn(M, H).abs() * 0.1
scale = math.sqrt(d)
o, lse = ops.attn_fwd(BF16, 1, q, k, v, scale, sqk, 32.0)
qpre = ops.attn_q_prescale(d)
qs = (q.float() * qpre).bfloat16()     # what GEMM epilogue 4 writes on the training path (q_prescale folded in)
dqkv = torch.empty(M, 3 * C, device=dev, dtype=torch.bfloat16)

def t_of(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

cases = {
    "fwd generic": (lambda: ops.attn_fwd(BF16, 1, q, k, v, scale), 4.0),
    "fwd bounded": (lambda: ops.attn_fwd(BF16, 1, q, k, v, scale, sqk, 32.0), 4.0),
    "bwd unfused": (lambda: ops.attn_bwd(BF16, 1, gt, q, k, v, o, lse, scale), 10.0),
    "bwd fused  ": (lambda: ops.attn_bwd_qknorm(gt, q, k, v, o, lse, scale, rq, rk, sqk, 32.0, dqkv, 3 * C, dqkv[:, C:],
                                                dqkv[:, 2 * C:], 3 * C), 10.0),
    "fwd bounded, q pre-scaled": (lambda: ops.attn_fwd(BF16, 1, qs, k, v, scale, sqk, 32.0, q_prescale=qpre), 4.0),
    "bwd fused,   q pre-scaled": (lambda: ops.attn_bwd_qknorm(gt, qs, k, v, o, lse, scale, rq, rk, sqk, 32.0, dqkv, 3 * C,
                                                              dqkv[:, C:], dqkv[:, 2 * C:], 3 * C, q_prescale=qpre), 10.0),
}
res = {n: [] for n in cases}
for rnd_ in range(5):
    for n, (fn, mult) in cases.items():
        res[n].append(t_of(fn))
for n, (fn, mult) in cases.items():
    ts = sorted(res[n]); med = ts[len(ts) // 2]
    fl = mult * B * H * T * T * d
    print(f"{n:26s}: median {med * 1e3:7.1f} us  min {ts[0] * 1e3:7.1f} us  {fl / med / 1e9:7.1f} TF/s algorithmic")
