#!/usr/bin/env python3
"""Control-flow interpreter for the generated attention-backward loops: runs the scalar instructions that steer branches
(s_mov / s_add / s_sub / s_lshl / s_cmp_* / s_bitcmp* / s_cselect / s_cbranch_scc* / s_branch) of one wave with given
operands and counts what it executes - s_barrier above all: every wave of a workgroup must execute the same number of
barriers whatever its role (computing wave, wave without keys, half without a work item, first / second half), or the
workgroup hangs.  Everything else (vector, LDS, memory instructions) is counted, not executed.
    python3 tools/probes/asm_barrier_sim.py nvit_amd/csrc/attn_dkv_pp_asm.inc"""
import re, sys, itertools


def load(path):
    ins = []
    for line in open(path):
        m = re.match(r'\s*"(.*?)\\n\\t" \\', line)
        if m:
            s = m.group(1).strip()
            if s and not s.startswith(";"):
                ins.append(s)
    return ins


def run(ins, ops, limit=2_000_000):
    labels = {s[:-1]: i for i, s in enumerate(ins) if s.endswith(":")}
    reg, scc, pc = {}, 0, 0
    count = {"s_barrier": 0, "mfma": 0, "dma": 0, "ds_read": 0, "steps": 0}

    def val(tok):
        tok = tok.strip()
        if tok.startswith("%"):
            return ops.get(int(tok[1:]), 0)
        if re.fullmatch(r"s\d+", tok):
            return reg.get(tok, 0)
        if tok.startswith("0x"):
            return int(tok, 16)
        if re.fullmatch(r"-?\d+", tok):
            return int(tok)
        return 0   # pairs, m0, exec, vector registers: not modelled

    while pc < len(ins):
        count["steps"] += 1
        if count["steps"] > limit:
            raise RuntimeError("no termination")
        s = ins[pc]
        pc += 1
        if s.endswith(":"):
            continue
        op, _, rest = s.partition(" ")
        a = [x.strip() for x in rest.split(",")] if rest else []
        if op == "s_barrier":
            count["s_barrier"] += 1
        elif op.startswith("v_mfma"):
            count["mfma"] += 1
        elif op.startswith("global_load_lds"):
            count["dma"] += 1
        elif op.startswith("ds_read"):
            count["ds_read"] += 1
        elif op == "s_mov_b32" and re.fullmatch(r"s\d+", a[0]):
            reg[a[0]] = val(a[1]) & 0xFFFFFFFF
        elif op in ("s_add_u32", "s_sub_u32", "s_lshl_b32", "s_and_b32", "s_or_b32") and re.fullmatch(r"s\d+", a[0]):
            x, y = val(a[1]), val(a[2])
            r = {"s_add_u32": x + y, "s_sub_u32": x - y, "s_lshl_b32": x << (y & 31), "s_and_b32": x & y, "s_or_b32": x | y}[op]
            reg[a[0]] = r & 0xFFFFFFFF
            scc = int(r != (r & 0xFFFFFFFF)) if op in ("s_add_u32", "s_sub_u32") else int((r & 0xFFFFFFFF) != 0)
        elif op.startswith("s_cmp_"):
            x, y = val(a[0]), val(a[1])
            scc = int({"lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y, "eq": x == y, "lg": x != y}[op.split("_")[2]])
        elif op == "s_bitcmp1_b32":
            scc = (val(a[0]) >> val(a[1])) & 1
        elif op == "s_bitcmp0_b32":
            scc = 1 - ((val(a[0]) >> val(a[1])) & 1)
        elif op == "s_cselect_b32":
            reg[a[0]] = val(a[1]) if scc else val(a[2])
        elif op == "s_cbranch_scc1":
            if scc:
                pc = labels[a[0]]
        elif op == "s_cbranch_scc0":
            if not scc:
                pc = labels[a[0]]
        elif op == "s_branch":
            pc = labels[a[0]]
        elif op.startswith("s_cbranch"):
            raise RuntimeError("unmodelled branch " + s)
    return count


if __name__ == "__main__":
    ins = load(sys.argv[1])
    pp = "pp" in sys.argv[1]
    bad = 0
    for T in (16, 49, 64, 65, 96, 128, 200, 784, 832):
        nt = (T + 63) // 64
        nvl = T - (nt - 1) * 64
        res = {}
        roles = [(a, h, w) for a in ((3, 2, 0) if pp else (1, 0)) for h in ((0, 1) if pp else (0,)) for w in range(4)]
        for act, half, w in roles:
            ops = {6: nt, 9: nvl, 10: act | (half << 2), 11: w * 1024, 8: 0x10000 * half, 7: 1536}
            res[(act, half, w)] = run(ins, ops)
        bars = {k: v["s_barrier"] for k, v in res.items()}
        ok = len(set(bars.values())) == 1
        bad += not ok
        k0 = (3, 0, 0) if pp else (1, 0, 0)
        print(f"T={T:4d} nt={nt:2d} nvalid_last={nvl:2d}: barriers {sorted(set(bars.values()))} {'OK' if ok else 'MISMATCH ' + str(bars)}; "
              f"computing wave: {res[k0]['mfma']} MFMA, {res[k0]['dma']} DMA, {res[k0]['ds_read']} LDS reads")
    sys.exit(1 if bad else 0)
