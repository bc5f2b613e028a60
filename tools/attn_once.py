"""A few launches of the attention kernels at the benchmarked shape, for rocprofv3 PMC passes: python tools/attn_once.py"""
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
qpre = ops.attn_q_prescale(d)
q = (se * torch.nn.functional.normalize(rn(B, H, T, d), dim=-1) * qpre).bfloat16()
k = (se * torch.nn.functional.normalize(rn(B, H, T, d), dim=-1)).bfloat16()
v = (rn(B, H, T, d) * 0.05).bfloat16()
gt = (rn(M, C) * 1e-3).bfloat16()
rq = 1.0 + rn(M, H).abs() * 0.1
rk = 1.0 + rn(M, H).abs() * 0.1
scale = math.sqrt(d)
dqkv = torch.empty(M, 3 * C, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    o, lse = ops.attn_fwd(BF16, 1, q, k, v, scale, sqk, 32.0, q_prescale=qpre)
    ops.attn_bwd_qknorm(gt, q, k, v, o, lse, scale, rq, rk, sqk, 32.0, dqkv, 3 * C, dqkv[:, C:], dqkv[:, 2 * C:], 3 * C,
                        q_prescale=qpre)
torch.cuda.synchronize()
print("done")
