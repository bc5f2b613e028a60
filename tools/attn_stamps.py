"""In-kernel cycle stamps of the dk/dv kernel (diagnostic library built with -DNVIT_PROBE_ATTN_STAMPS): per-tile segment
cycles averaged over the active waves.  NVIT_LIB=.../libnvit_hip.so.stamps python tools/attn_stamps.py"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops, _lib
from nvit_amd._lib import BF16

dev = torch.device("cuda:0")
B, H, T, d = 128, 12, 784, 64
C_, M = H * d, B * T
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
sqk = (1.0 / 32) * (1.0 + 0.05 * torch.tanh(rn(C_)))
se = (sqk * 32.0).reshape(1, H, 1, d)
qpre = ops.attn_q_prescale(d)
q = (se * torch.nn.functional.normalize(rn(B, H, T, d), dim=-1) * qpre).bfloat16()
k = (se * torch.nn.functional.normalize(rn(B, H, T, d), dim=-1)).bfloat16()
v = (rn(B, H, T, d) * 0.05).bfloat16()
gt = (rn(M, C_) * 1e-3).bfloat16()
rq = 1.0 + rn(M, H).abs() * 0.1
rk = 1.0 + rn(M, H).abs() * 0.1
scale = math.sqrt(d)
dqkv = torch.empty(M, 3 * C_, device=dev, dtype=torch.bfloat16)
o, lse = ops.attn_fwd(BF16, 1, q, k, v, scale, sqk, 32.0, q_prescale=qpre)
lib = _lib.load()
fn = lib.nvit_attn_stamps_read
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
def run():
    ops.attn_bwd_qknorm(gt, q, k, v, o, lse, scale, rq, rk, sqk, 32.0, dqkv, 3 * C_, dqkv[:, C_:], dqkv[:, 2 * C_:], 3 * C_,
                        q_prescale=qpre)
    torch.cuda.synchronize()
for _ in range(2): run()
fn(buf, 1)
n = 3
for _ in range(n): run()
fn(buf, 1)
waves = buf[15]
tiles = 13   # tile bodies per wave (the last one is the ragged path and only carries stamps 0 and 7)
names = ["DMA issue", "s2=0 reads+wait", "s2=0 S/dP+softmax", "s2=0 dV/dK", "s2=1 reads+wait", "s2=1 S/dP+softmax", "s2=1 dV/dK",
         "vmcnt wait (+masked tile body)", "barrier"]
tot = 0
for i, nm in enumerate(names):
    c = buf[i] / waves / tiles
    tot += c
    print(f"{nm:36s} {c:8.1f} cycles per tile per wave")
print(f"{'sum':36s} {tot:8.1f}   (waves {waves // n} per launch)")
