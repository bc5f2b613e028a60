"""Times the fused attention backward (dq + dkv kernels of one layer) at the benchmarked shape with whatever library
NVIT_LIB names - for same-box comparisons of builds (e.g. the previous round's library).  python tools/attn_bwd_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from nvit_amd import _lib
if os.environ.get("NVIT_LIB"):   # an older build may lack newer entry points: bind what it has
    _l = ctypes.CDLL(os.environ["NVIT_LIB"])
    for _n, _a in _lib.SIGNATURES.items():
        if hasattr(_l, _n):
            getattr(_l, _n).argtypes = _a
            getattr(_l, _n).restype = _lib._RESTYPES.get(_n, ctypes.c_int)
    _lib._lib = _l
from nvit_amd import ops
from attn_dkv_asm_ab import make, dev, t_of

c = make(128, 12, 784)
dqkv = torch.empty(c["M"], 3 * c["C"], device=dev, dtype=torch.bfloat16)


def bwd():
    ops.attn_bwd_qknorm(c["gt"], c["qs"], c["k"], c["v"], c["o"], c["lse"], c["scale"], c["rq"], c["rk"], c["sqk"], 32.0, dqkv,
                        3 * c["C"], dqkv[:, c["C"]:], dqkv[:, 2 * c["C"]:], 3 * c["C"], q_prescale=c["qpre"])


ts = sorted(t_of(bwd, n=12) for _ in range(15))
print(f"{os.environ.get('NVIT_LIB', 'product library')}: backward (dq + dkv) median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us")
