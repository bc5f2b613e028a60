"""Where the time of the hand-placed dK/dV kernel goes: s_memtime stamps from inside the generated loop (GPU box).

Needs the stamped variant library (bash tools/dkv_stamps.sh in the build container); loads it through NVIT_LIB, runs the
stamped kernel (tools/probes/attn_dkv_stamps.hip) at the benchmarked shape and prints, per wave and averaged over the
workgroups: C++ prologue, loop prologue (K/V fragments + first two tiles landing), first barrier, the tile loop, the
vmcnt(0) and s_barrier waits inside it, accumulator dump, C++ epilogue; then how the two workgroups of a CU overlap.
    NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.dkv_stamps python tools/dkv_stamps.py [B H T]"""
import ctypes as C
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nvit_amd import ops, _lib
from nvit_amd._lib import BF16
from attn_dkv_asm_ab import make, dev

lib = _lib.load()
fn = getattr(lib, "nvit_probe_attn_dkv_stamps")
vp, ci, cf = C.c_void_p, C.c_int, C.c_float
fn.argtypes = [vp, vp, vp, vp, vp, cf, vp, vp, cf, cf, vp, vp, ci, vp, ci, ci, ci, ci, vp, vp]
fn.restype = ci


def main(B, H, T):
    c = make(B, H, T, seed=1)
    Cc, M = c["C"], c["M"]
    p = lambda t: t.data_ptr()
    dqkv = torch.zeros(M, 3 * Cc, device=dev, dtype=torch.bfloat16)
    part_q = torch.empty((B * math.ceil(T / 128), Cc), device=dev)
    part_k = torch.empty((B * math.ceil(T / 128), Cc), device=dev)
    delta = torch.empty((2, B, H, T), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.nvit_attn_bwd_qknorm(BF16, p(c["gt"]), p(c["qs"]), p(c["k"]), p(c["v"]), p(c["o"]), p(c["lse"]), c["scale"], p(c["rq"]),
                                  p(c["rk"]), p(c["sqk"]), 32.0, c["qpre"], p(dqkv), 3 * Cc, p(dqkv[:, Cc:]), p(dqkv[:, 2 * Cc:]),
                                  3 * Cc, p(part_q), p(part_k), p(delta), B, H, T, T, 64, st)
    assert rc == 0
    torch.cuda.synchronize()
    ref = dqkv.clone()
    nwg = math.ceil(T / 128) * B * H
    stamps = torch.zeros(nwg * 4, 16, device=dev, dtype=torch.int32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def cu_span(sn):
        sn = sn[sn[:, 5] != 0]
        key = sn[:, 9] & 0xFFF00
        sp = []
        for k in np.unique(key):
            m = key == k
            base = sn[m, 12][0]
            ent = ((sn[m, 12] - base + 2 ** 31) & 0xFFFFFFFF) - 2 ** 31
            ex = ((sn[m, 13] - base + 2 ** 31) & 0xFFFFFFFF) - 2 ** 31
            sp.append(ex.max() - ent.min())
        return float(np.median(sp))

    spans_it = []
    for it in range(9):
        stamps.zero_()
        e0.record()
        rc = fn(p(c["gt"]), p(c["qs"]), p(c["k"]), p(c["v"]), p(delta), c["scale"], p(c["rk"]), p(c["sqk"]), 32.0, c["qpre"],
                p(dqkv[:, Cc:]), p(dqkv[:, 2 * Cc:]), 3 * Cc, p(part_k), B, H, T, T, p(stamps), st)
        e1.record()
        assert rc == 0
        torch.cuda.synchronize()
        spans_it.append(cu_span(stamps.cpu().numpy().astype(np.int64) & 0xFFFFFFFF))
    us = e0.elapsed_time(e1) * 1e3
    print(f"KERNEL CYCLES (busy span of a CU, median over CUs; the clock-independent figure): per launch "
          f"{[int(x) for x in spans_it]} -> median {int(np.median(spans_it[2:]))}")
    print(f"stamped kernel: {us:.1f} us; results identical to the product: {bool((dqkv == ref).all().item())}")
    s = stamps.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    act = s[:, 5] != 0                       # waves that ran the loop (a workgroup's waves past the last key only feed)
    s = s[act]
    d = lambda a, b: (a - b) & 0xFFFFFFFF   # 32-bit wrap
    # the counters of different XCDs are not aligned: calibrate the tick on the busy span of each CU (first entry -> last exit)
    key = s[:, 9] & 0xFFF00
    spans = []
    for k in np.unique(key):
        m = key == k
        base = s[m, 12][0]
        ent, ex = d(s[m, 12], base).astype(np.int64), d(s[m, 13], base).astype(np.int64)
        ent = np.where(ent > 2 ** 31, ent - 2 ** 32, ent)
        ex = np.where(ex > 2 ** 31, ex - 2 ** 32, ex)
        spans.append(ex.max() - ent.min())
    span = float(np.median(spans))
    tick_ns = us * 1e3 / span
    print(f"{act.sum()} loop waves; busy span of a CU {span:.0f} ticks (median) -> one tick = {tick_ns:.3f} ns ({1 / tick_ns:.3f} GHz)")
    seg = {
        "C++ prologue (entry -> loop statement)": d(s[:, 0], s[:, 12]),
        "loop prologue (K/V + two tiles landed)": d(s[:, 1], s[:, 0]),
        "first barrier": d(s[:, 2], s[:, 1]),
        "tile loop (loop entry -> drained)": d(s[:, 3], s[:, 2]),
        "   of it: vmcnt(0) waits": s[:, 6],
        "   of it: s_barrier waits": s[:, 7],
        "last barrier": d(s[:, 4], s[:, 3]),
        "accumulator dump": d(s[:, 5], s[:, 4]),
        "C++ epilogue (dump -> exit)": d(s[:, 13], s[:, 5]),
        "whole wave": d(s[:, 13], s[:, 12]),
    }
    for k, v in seg.items():
        print(f"  {k:42s} mean {v.mean() * tick_ns / 1e3:7.3f} us   p10 {np.percentile(v, 10) * tick_ns / 1e3:7.3f}   "
              f"p90 {np.percentile(v, 90) * tick_ns / 1e3:7.3f}")
    nt = int(s[:, 8].max()) + 1
    print(f"  tiles per wave {nt}; loop time per tile {seg['tile loop (loop entry -> drained)'].mean() * tick_ns / nt:7.1f} ns "
          f"({seg['tile loop (loop entry -> drained)'].mean() / nt:7.0f} ticks; 64 MFMA x 16 cycles x 2 waves/SIMD = 2048 cycles)")
    # how the workgroups sharing a CU overlap: for every wave-0 record, what fraction of ITS loop ran while another workgroup
    # of the same CU (HW_ID: SE, CU, XCC fields) was outside its loop
    hw = s[:, 9]
    cu_key = hw & np.int64(0xFFF00)                    # XCC 19:16, SE 15:13, SH 12, CU 11:8 (wave 3:0, SIMD 5:4 dropped)
    order = {}
    first = {}
    for i in range(len(s)):
        if (hw[i] >> 4) & 3 != 0:
            continue                                    # one wave per workgroup (SIMD 0)
        k = int(cu_key[i])
        t0 = first.setdefault(k, int(s[i, 12]))
        rel = lambda x: ((int(x) - t0 + 2 ** 31) & 0xFFFFFFFF) - 2 ** 31
        order.setdefault(k, []).append((rel(s[i, 12]), rel(s[i, 2]), rel(s[i, 3]), rel(s[i, 13])))
    tot = covered = 0
    for k, v in order.items():
        v.sort()
        for a in v:
            lo, hi = a[1], a[2]
            tot += hi - lo
            for b_ in v:
                if b_ is a:
                    continue
                o = max(0, min(hi, b_[2]) - max(lo, b_[1]))   # both in their loops
                covered += o
    print(f"  CUs seen {len(order)}; share of a workgroup's loop time during which another workgroup of the CU is also in its loop: "
          f"{covered / max(tot, 1):.3f}")
    k0 = sorted(order)[0]
    print("  one CU, first workgroups (us: entry, loop entry, drained, exit):")
    m0 = min(a[0] for a in order[k0])
    for a in [tuple(x - m0 for x in a) for a in order[k0][:10]]:
        print("    " + "  ".join(f"{x * tick_ns / 1e3:8.2f}" for x in a))


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else [128, 12, 784]
    main(*a)
