"""Where the time of the dQ kernel goes: s_memtime stamps at entry / tile-loop start / tile-loop end / exit (GPU box).
    NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.dq_stamps python tools/dq_stamps.py [B H T]"""
import ctypes as C
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nvit_amd import _lib
from attn_dkv_asm_ab import make, dev

lib = _lib.load()
fn = getattr(lib, "nvit_probe_attn_dq_stamps")
vp, ci, cf = C.c_void_p, C.c_int, C.c_float
fn.argtypes = [vp, vp, vp, vp, vp, vp, vp, cf, vp, vp, cf, cf, vp, ci, vp, ci, ci, ci, ci, vp, vp]
fn.restype = ci


def main(B, H, T):
    c = make(B, H, T, seed=1)
    Cc, M = c["C"], c["M"]
    p = lambda t: t.data_ptr()
    dqkv = torch.zeros(M, 3 * Cc, device=dev, dtype=torch.bfloat16)
    part_q = torch.empty((B * math.ceil(T / 128), Cc), device=dev)
    delta = torch.empty((2, B, H, T), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    nwg = math.ceil(T / 128) * B * H
    stamps = torch.zeros(nwg * 4, 16, device=dev, dtype=torch.int32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    spans = []
    for it in range(7):
        stamps.zero_()
        e0.record()
        rc = fn(p(c["gt"]), p(c["qs"]), p(c["k"]), p(c["v"]), p(c["o"]), p(c["lse"]), p(delta), c["scale"], p(c["rq"]), p(c["sqk"]), 32.0,
                c["qpre"], p(dqkv), 3 * Cc, p(part_q), B, H, T, T, p(stamps), st)
        e1.record()
        assert rc == 0
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        s = s[s[:, 13] != 0]
        key = s[:, 9] & 0xFFF00
        sp = []
        for k in np.unique(key):
            m = key == k
            base = s[m, 12][0]
            ent = ((s[m, 12] - base + 2 ** 31) & 0xFFFFFFFF) - 2 ** 31
            ex = ((s[m, 13] - base + 2 ** 31) & 0xFFFFFFFF) - 2 ** 31
            sp.append(ex.max() - ent.min())
        spans.append(float(np.median(sp)))
    us = e0.elapsed_time(e1) * 1e3
    span = float(np.median(spans[2:]))
    print(f"KERNEL CYCLES (busy span of a CU, median over CUs): per launch {[int(x) for x in spans]} -> median {int(span)}; last launch {us:.1f} us")
    d = lambda a, b: (a - b) & 0xFFFFFFFF
    seg = {"prologue (entry -> tile loop)": d(s[:, 2], s[:, 0]), "tile loop": d(s[:, 3], s[:, 2]), "epilogue (loop end -> exit)": d(s[:, 5], s[:, 3]),
           "whole wave": d(s[:, 5], s[:, 0])}
    for k, v in seg.items():
        print(f"  {k:32s} mean {v.mean():9.0f} cycles   p10 {np.percentile(v, 10):9.0f}   p90 {np.percentile(v, 90):9.0f}")
    nt = math.ceil(T / 64)
    print(f"  tiles {nt}; loop cycles per tile {seg['tile loop'].mean() / nt:7.0f} (48 MFMA x 16 = 768 cycles of matrix pipe per wave)")
    print(f"  workgroups per CU at a time: {len(s) / 4 / len(np.unique(key)) / (span / seg['whole wave'].mean()):.2f}")


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else [128, 12, 784]
    main(*a)
