"""Can an HBM-bound row kernel run UNDER an MFMA-bound persistent GEMM (different pipes) when both fit a CU?

The persistent GEMMs take one 512-thread workgroup per CU; whether a row-kernel wave can be co-resident is decided by
the register file: 2 GEMM waves per SIMD x their VGPR allocation + the row kernel's allocation <= 512.  The weight-
gradient GEMM has a 193-register variant (4 x 32 KiB ring, mode 3: 2 x 200 allocated -> 112 left) and a 230-register one
(2 x 64 KiB ring, mode 5: 48 left); lerp_fwd at C=768 takes 72, lerp_bwd 160.  Timed on the Base shapes, interleaved:
GEMM alone, row kernels alone, both back to back on one stream, both on two streams.   python tools/overlap_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops, _lib
from nvit_amd._lib import BF16

dev = torch.device("cuda:0")
M, C = 100352, 768
lib = _lib.load()
A = (torch.randn(M, 8 * C, device=dev) * 0.05).bfloat16()
B = torch.randn(M, C, device=dev).bfloat16()
G = torch.empty(8 * C, C, device=dev)
h = torch.nn.functional.normalize(torch.randn(M, C, device=dev), dim=-1)
y = torch.randn(M, C, device=dev)
dout = torch.randn(M, C, device=dev)
alpha = torch.full((C,), 1 / 32, device=dev)
s2 = torch.cuda.Stream()
NROW = 4


def gemm():
    ops.gemm_tn(A, B, G, M, 8 * C, C, perm=1)


def rows_fwd():
    for _ in range(NROW):
        ops.lerp_fwd(BF16, h, y, alpha, 1.6, want_lo=True)


yb = y.bfloat16()
add = torch.randn(M, C, device=dev).bfloat16()
xs = torch.nn.functional.normalize(torch.randn(M, C, device=dev), dim=-1)
skip = torch.tensor([0.9], device=dev)
dh_acc = torch.zeros(M, C, device=dev)


def rows_bwd():   # the two backward row kernels of a block as the step calls them
    ops.lerp_bwd(BF16, dout, h, yb, alpha, 1.6, xs, skip, None, False, False, True, dout_add=add)
    ops.lerp_bwd(BF16, dout, h, yb, alpha, 1.6, None, None, dh_acc, True, False, True, dout_add=add)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def both_streams(rows):
    def f():
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        gemm()
        with torch.cuda.stream(s2):
            rows()
        cur.wait_stream(s2)
    return f


for rows, tag in ((rows_fwd, f"{NROW} x lerp_fwd (72 VGPRs)"), (rows_bwd, "lerp_bwd MLP half + attention half")):
    for mode, mtag in ((5, "TN 2x64KiB ring, 230 VGPRs"), (3, "TN 4x32KiB ring, 193 VGPRs")):
        lib.nvit_set_tn_order(mode)
        res = {k: [] for k in ("gemm", "rows", "serial", "two_streams")}
        for rnd in range(4):
            res["gemm"].append(timed(gemm))
            res["rows"].append(timed(rows))
            res["serial"].append(timed(lambda: (gemm(), rows())))
            res["two_streams"].append(timed(both_streams(rows)))
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        print(f"{mtag:28s} + {tag:38s}: gemm {med['gemm']:7.1f} us, rows {med['rows']:7.1f} us, one stream {med['serial']:7.1f} us, "
              f"two streams {med['two_streams']:7.1f} us  (hidden: {med['serial'] - med['two_streams']:6.1f} us of {med['rows']:.0f})")
lib.nvit_set_tn_order(5)
