#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/, scratch) into the small summaries committed under profiles/.

  python tools/summarize_profile.py <round-tag> --trace gpurun_out/prof_x --fetch gpurun_out/pmc_fetch \
         --write gpurun_out/pmc_write [--cmd "python3 bench.py ..."]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim columns, top 40),
profiles/<tag>_pmc.csv (per-kernel average FETCH_SIZE / WRITE_SIZE per launch) and profiles/<tag>_pmc.json
(gemm_nt family: HBM-side traffic per launch, gfx950 correction applied: FETCH_SIZE counts 64 B per 128-B
request on wide coalesced reads -> x2; both counters are in KiB; MI355X_MICROARCH.md §HBM)."""
import argparse, collections, csv, glob, json, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for key in ("gemm_nt_persistent", "gemm_tn_persistent", "gemm_nt_kernel", "gemm_tn_kernel", "attn_fwd_mfma",
                "attn_bwd_dq_mfma", "attn_bwd_dkv_mfma", "attn_bwd_dkv_asm32", "attn_fwd_ref", "attn_delta", "lerp_fwd", "lerp_bwd", "qknorm_fwd",
                "qknorm_bwd", "swiglu_fwd", "swiglu_bwd", "colsum_reduce", "colsum_kernel", "slab_reduce", "adamw_renorm", "grad_sqnorm", "renorm",
                "shadow", "patch_embed", "im2col", "pool", "recon", "cast_kernel", "scale_cols", "FusedAdam", "multi_tensor",
                "elementwise", "rocclr", "reduce_kernel", "softmax", "nll_loss"):
        if key in name:
            if key.startswith("gemm_nt_persistent"):
                import re
                m = re.search(r"Li(\d)ELi(\d)E", name)
                return f"gemm_nt_persistent<FM={m.group(1)},EPI={m.group(2)}>" if m else key
            return key
    return name[:60]


def _db(d):
    """rocprofv3 writes either CSV files or one rocpd SQLite database, depending on its default output format."""
    f = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return f[0] if f else None


def kernel_stats(d):
    """Rows {Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs}, longest total first."""
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        return list(csv.DictReader(open(f[0])))
    import sqlite3
    c = sqlite3.connect(_db(d))
    agg = collections.OrderedDict()
    for name, dur in c.execute("select name, duration from kernels"):
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values()) or 1
    rows = [{"Name": n, "Calls": a[0], "TotalDurationNs": a[1], "AverageNs": round(a[1] / a[0], 6),
             "Percentage": round(100.0 * a[1] / tot, 4), "MinNs": a[2], "MaxNs": a[3]} for n, a in agg.items()]
    rows.sort(key=lambda r: -r["TotalDurationNs"])
    return rows


def counter_rows(d, counter):
    """(kernel name, counter value summed over its instances) per dispatch."""
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if f:
        for r in csv.DictReader(open(f[0])):
            yield r["Kernel_Name"], float(r["Counter_Value"])
        return
    import sqlite3
    c = sqlite3.connect(_db(d))
    q = "select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name = ? group by dispatch_id, kernel_name"
    for _, name, value in c.execute(q, (counter,)):
        yield name, float(value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--trace")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--mfma", help="PMC pass with SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE")
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    if a.trace:
        rows = kernel_stats(a.trace)
        with open(os.path.join(out, f"{a.tag}_kernel_stats.csv"), "w") as w:
            w.write(f"# rocprofv3 --kernel-trace --stats -- {a.cmd}\n")
            cw = csv.writer(w)
            cw.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows[:40]:
                cw.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                             r["MinNs"], r["MaxNs"]])
    if a.mfma:
        # MFMA pipe utilisation per kernel: busy cycles summed over the 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs * 1024)
        busy = collections.defaultdict(float)
        act = collections.defaultdict(float)
        n = collections.Counter()
        for name, v in counter_rows(a.mfma, "SQ_VALU_MFMA_BUSY_CYCLES"):
            busy[short(name)] += v
        for name, v in counter_rows(a.mfma, "GRBM_GUI_ACTIVE"):
            act[short(name)] += v
            n[short(name)] += 1
        with open(os.path.join(out, f"{a.tag}_mfma_util.csv"), "w") as w:
            w.write(f"# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- {a.cmd}; "
                    "util = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)\n")
            cw = csv.writer(w)
            cw.writerow(["Kernel", "Launches", "GRBM_GUI_ACTIVE_total", "MFMA_busy_fraction"])
            for k in sorted(act, key=lambda k: -act[k]):
                if busy[k] > 0:
                    cw.writerow([k, n[k], int(act[k]), round(busy[k] / (act[k] / 8 * 1024), 4)])
    pm = {}
    for kind, d in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
        if not d:
            continue
        agg = collections.defaultdict(lambda: [0, 0.0])
        for name, value in counter_rows(d, kind):
            k = short(name)
            agg[k][0] += 1
            agg[k][1] += value
        pm[kind] = agg
    if pm:
        names = sorted(set().union(*[set(v) for v in pm.values()]))
        with open(os.path.join(out, f"{a.tag}_pmc.csv"), "w") as w:
            w.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- {a.cmd}; values in KiB per launch (raw)\n")
            cw = csv.writer(w)
            cw.writerow(["Kernel", "Launches", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg"])
            for n in names:
                fe = pm.get("FETCH_SIZE", {}).get(n, [0, 0.0])
                wr = pm.get("WRITE_SIZE", {}).get(n, [0, 0.0])
                cw.writerow([n, max(fe[0], wr[0]), round(fe[1] / max(1, fe[0]), 1), round(wr[1] / max(1, wr[0]), 1)])
        # the family bench.py's roofline is quoted on: bf16 NT GEMMs with the plain epilogue (EPI 0/1/2)
        fam = [n for n in names if n.startswith("gemm_nt") and not any(f"EPI={e}" in n for e in (3, 4, 5))]
        nl = sum(pm["FETCH_SIZE"][n][0] for n in fam) if "FETCH_SIZE" in pm else 0
        fetch = sum(pm["FETCH_SIZE"][n][1] for n in fam) if "FETCH_SIZE" in pm else 0.0
        write = sum(pm["WRITE_SIZE"][n][1] for n in fam) if "WRITE_SIZE" in pm else 0.0
        # every GEMM family bench.py lists (same correction): bytes per launch
        def fam_traffic(pred):
            ks = [n for n in names if pred(n)]
            nl_ = sum(pm["FETCH_SIZE"][n][0] for n in ks) if "FETCH_SIZE" in pm else 0
            fe_ = sum(pm["FETCH_SIZE"][n][1] for n in ks) if "FETCH_SIZE" in pm else 0.0
            wr_ = sum(pm["WRITE_SIZE"][n][1] for n in ks) if "WRITE_SIZE" in pm else 0.0
            return (2.0 * fe_ + wr_) * 1024 / nl_ if nl_ else None
        families = {"gemm_nt_plain": fam_traffic(lambda n: n in fam),
                    "gemm_nt_epi3": fam_traffic(lambda n: n.startswith("gemm_nt") and "EPI=3" in n),
                    "gemm_nt_epi4": fam_traffic(lambda n: n.startswith("gemm_nt") and "EPI=4" in n),
                    "gemm_nt_epi5": fam_traffic(lambda n: n.startswith("gemm_nt") and "EPI=5" in n),
                    "gemm_tn": fam_traffic(lambda n: n.startswith("gemm_tn_persistent")),
                    "attn_fwd": fam_traffic(lambda n: n.startswith("attn_fwd")),
                    "attn_bwd_dq": fam_traffic(lambda n: n.startswith("attn_bwd_dq")),
                    "attn_bwd_dkv": fam_traffic(lambda n: n.startswith("attn_bwd_dkv")),
                    "lerp_fwd": fam_traffic(lambda n: n.startswith("lerp_fwd")),
                    "lerp_bwd": fam_traffic(lambda n: n.startswith("lerp_bwd"))}
        families = {k: v for k, v in families.items() if v}
        js = {"family": "gemm_nt (plain epilogue)", "families": families, "launches": nl, "fetch_bytes_per_launch_corrected": 2.0 * fetch * 1024 / max(1, nl),
              "write_bytes_per_launch": write * 1024 / max(1, nl),
              "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024 / max(1, nl),
              "note": "L2<->fabric bytes (Infinity-Cache hits are included by these counters); FETCH_SIZE x2 per the gfx950 correction",
              "cmd": a.cmd}
        json.dump(js, open(os.path.join(out, f"{a.tag}_pmc.json"), "w"), indent=1)
        print(js)


if __name__ == "__main__":
    main()
