#!/usr/bin/env python3
"""Generator of the hand-placed main loop of the attention backward dK/dV kernel (gfx950), TWO waves per SIMD form.

Writes attn_dkv32_asm.inc: ONE inline-asm string (the whole tile loop of attn_bwd_dkv_asm32_kernel in attn_mfma.hip) plus
its clobber list.  Reference semantics: the dK/dV half of the backward of F.scaled_dot_product_attention as the reference
calls it (/root/reference/nvit/model.py:121-124); arithmetic, operand layouts and accumulation order are those of the
compiler-built attn_bwd_dkv_mfma_kernel, against which this one is bit-exact.

Same geometry as the compiler-built kernel (a workgroup = 4 waves x 32 keys, two workgroups per CU, so one workgroup's
prologue / epilogue runs under the other's tile loop), but the tile loop is a software pipeline placed by this script:
the work of a 64-query tile is cut into 4 groups g = (32-query half, key fragment); every step issues M1(g+1) [S and dP
products, 8 MFMA], V(g) [exp2, p*(dP-delta), bf16 packs: 24 VALU] and M2(g-1) [dV and dK products, 8 MFMA], so every
dependency crosses a step boundary.  Fragment registers are single-buffered (256 registers per wave): the row fragments of
the next half are requested right behind the last MFMA that reads the current ones, the transposed fragments likewise,
and land under the rest of the step.  128 registers per wave sit in the accumulation half (dK / dV accumulators, K / V
fragments, nothing the compiler ever sees); the accumulators return through LDS.

usage: python3 gen_attn_dkv32_asm.py > ../attn_dkv32_asm.inc
"""
import os

PROBE = set(filter(None, os.environ.get("GEN_PROBE", "").split(",")))
# placement options (results stay exact).  Product = ring4,spreadtr,earlyrows,prio: the first three -1.6 % kernel cycles
# together against none of them (profiles/r04_dkv_cycles_ring4_spread.log), prio (s_setprio 1 for the tile loop: it
# outranks a partner wave that is in its prologue / epilogue) another -1.8 % (profiles/r04_dkv_cycles_prio.log);
# GEN_OPT=none builds without; vsched (one transcendental per MFMA gap): +-0
OPT = set(filter(None, (os.environ.get("GEN_OPT") or "ring4,spreadtr,earlyrows,prio").split(","))) - {"none"}

SLOT = 2 * 8192 + 512      # Q tile | dO tile | -lse[64] | -delta[64]   (= DKV_SLOT of attn_mfma.hip)
RING4 = "ring4" in OPT     # 4-slot ring: the fetch of tile t+2 is spread over steps 0..2 of tile t (it may overwrite the slot of
                           # tile t-2 before this tile's barrier), instead of bunched into step 3 behind the barrier
NSLOT = 4 if RING4 else 3
NDMA = 6                   # DMA wave-instructions per wave per tile

OP = dict(qbase=0, gbase=1, lbase=2, dbase=3, kbase=4, vbase=5, nt=6, ldg=7, ring=8, nvalid_last=9, active=10, wofs=11,
          voff_q0=12, voff_g0=13, rows_last=14, chunk16=15, lane4=16, kvoff0=17, lds_a0=19, lds_a1=20, lds_rn=21, lds_tr0=22,
          dump=26)

S_Q, S_G, S_L, S_D = 40, 42, 44, 46
S_NT, S_LDG, S_RING, S_NVL = 48, 49, 50, 51
S_T, S_TD, S_SLOTC, S_SLOTD = 52, 53, 54, 55
S_TMP, S_TMP2 = 56, 57
S_EXEC = 58
S_P32, S_T64, S_WOFS, S_RINGEND = 60, 61, 62, 63
S_SAVE = 64
S_FLAGS, S_DW, S_M0, S_SLOTT = 66, 67, 68, 69     # S_SLOTT: LDS base of the tile whose transposed fragments are being read

# per-lane operands (voff_q0, voff_g0, rows_last, chunk16, lane4, kvoff, the seven LDS lane offsets, dump) are read straight from the
# statement's input registers: a wave has 256 registers, 96 of them in the accumulation half, 16 left to the compiler
V_RA0, V_RA1, V_RN, V_RT = 16, 17, 18, 19      # absolute LDS addresses: rows (may already point at the next tile) / transposed
V_TMP, V_TMP2 = 23, 24
V_ROW = 32                                     # 48: a[2][2] (16) gg (16) nl[2] (8) nd[2] (8)
V_Z = 80                                       # two sets of z[2] (8) + w[2] (8)
V_P = 112                                      # two sets of pb (4) + sb (4)
V_TR = 128                                     # ga[4] (16) qa[4] (16): dO^T / Q^T fragments
V_END = 160
S_B2 = 70                                      # 70:71 second-piece base
A_DK, A_DV, A_KF, A_VF, A_END = 0, 32, 64, 80, 96

# GEN_PROBE=stamps (tools/probes/attn_dkv_stamps.hip only): s_memtime stamps kept in s72..s82 and written out by lane 0
# behind the accumulator dump, 16 dwords per wave at operand 24.  The stamps of the barrier block are consumed at the next
# lgkmcnt(0) the loop has anyway (scalar-memory returns count on lgkmcnt), so the loop's waits stay what they are.
STAMPS = "stamps" in PROBE
INLOOP = STAMPS and "inloop" in PROBE      # also stamp the per-tile waits (three more s_memtime per tile)
S_SB = 72            # 72..77: three 64-bit stamps in flight (prologue: entry / landed / loop entry; barrier block: before
                     # vmcnt / after vmcnt / after s_barrier; end: drained / last barrier / dumped)
S_T0 = 78            # 78..80: low words of the prologue stamps
S_ACC_VM, S_ACC_BAR = 81, 82
S_STAMP_END = 83


def stamp(i):
    if STAMPS:
        e(f"s_memtime s[{S_SB + 2 * (i % 3)}:{S_SB + 2 * (i % 3) + 1}]")

out = []


def e(s):
    out.append(s)


def vr(b, n=4):
    return f"v[{b}:{b + n - 1}]" if n > 1 else f"v{b}"


def ar(b, n=4):
    return f"a[{b}:{b + n - 1}]" if n > 1 else f"a{b}"


def mfma(d, a, b, c):
    return f"v_mfma_f32_16x16x32_bf16 {d}, {a}, {b}, {c}"


MFMA32 = "mfma32" in PROBE   # TIMING ONLY (garbage results): half as many v_mfma_f32_32x32x16_bf16 in place of the 16x16x32


def m1_atoms(fn, zbuf):
    R, Z = V_ROW, V_Z + 16 * zbuf
    if MFMA32:
        return [f"v_mfma_f32_32x32x16_bf16 {vr(Z, 16)}, {vr(R + 4 * i)}, {ar(A_KF + (fn * 2 + i % 2) * 4)}, {vr(Z, 16)}" for i in range(4)]
    ks0, ks1 = [], []
    for qq in (0, 1):
        z, w = vr(Z + qq * 4), vr(Z + 8 + qq * 4)
        ks0.append(mfma(z, vr(R + (qq * 2 + 0) * 4), ar(A_KF + (fn * 2 + 0) * 4), vr(R + 32 + qq * 4)))
        ks0.append(mfma(w, vr(R + 16 + (qq * 2 + 0) * 4), ar(A_VF + (fn * 2 + 0) * 4), vr(R + 40 + qq * 4)))
        ks1.append(mfma(z, vr(R + (qq * 2 + 1) * 4), ar(A_KF + (fn * 2 + 1) * 4), z))
        ks1.append(mfma(w, vr(R + 16 + (qq * 2 + 1) * 4), ar(A_VF + (fn * 2 + 1) * 4), w))
    return ks0 + ks1


def m2_atoms(fp, pbuf):
    P = V_P + 8 * pbuf
    if MFMA32:
        return [f"v_mfma_f32_32x32x16_bf16 {ar((A_DV if i % 2 == 0 else A_DK) + 16 * fp, 16)}, {vr(V_TR + 4 * i)}, {vr(P + 4 * (i % 2))}, "
                f"{ar((A_DV if i % 2 == 0 else A_DK) + 16 * fp, 16)}" for i in range(4)]
    res = []
    for df in range(4):
        dv = ar(A_DV + (df * 2 + fp) * 4)
        dk = ar(A_DK + (df * 2 + fp) * 4)
        res.append(mfma(dv, vr(V_TR + df * 4), vr(P), dv))
        res.append(mfma(dk, vr(V_TR + 16 + df * 4), vr(P + 4), dk))
    return res


def v_atoms(zbuf, pbuf):
    Z, P = V_Z + 16 * zbuf, V_P + 8 * pbuf
    res = []
    for qq in (0, 1):
        z, w = Z + qq * 4, Z + 8 + qq * 4
        for r in range(4):
            res.append(f"v_exp_f32_e32 v{z + r}, v{z + r}" if "noexp" not in PROBE else f"v_mov_b32_e32 v{z + r}, v{z + r}")
        for r in range(4):
            res.append(f"v_mul_f32_e32 v{w + r}, v{z + r}, v{w + r}")
        res.append(f"v_cvt_pk_bf16_f32 v{P + qq * 2}, v{z}, v{z + 1}")
        res.append(f"v_cvt_pk_bf16_f32 v{P + qq * 2 + 1}, v{z + 2}, v{z + 3}")
        res.append(f"v_cvt_pk_bf16_f32 v{P + 4 + qq * 2}, v{w}, v{w + 1}")
        res.append(f"v_cvt_pk_bf16_f32 v{P + 4 + qq * 2 + 1}, v{w + 2}, v{w + 3}")
    return res


def v_sched(zbuf, pbuf):
    """The same 24 instructions as v_atoms, as 16 per-gap lists for a 16-MFMA step (GEN_OPT=vsched): at most one
    transcendental per gap (an MFMA leaves 8 of its 16 cycles of vector issue; v_exp_f32 takes 8, the others 4-5), the two
    32-query halves' chains interleaved so that a gap pairs an exp with a multiply or a multiply with a pack; the last two
    gaps stay empty (the next step's first MFMAs read the packs)."""
    Z, P = V_Z + 16 * zbuf, V_P + 8 * pbuf
    ex = lambda qq, r: f"v_exp_f32_e32 v{Z + qq * 4 + r}, v{Z + qq * 4 + r}"
    mu = lambda qq, r: f"v_mul_f32_e32 v{Z + 8 + qq * 4 + r}, v{Z + qq * 4 + r}, v{Z + 8 + qq * 4 + r}"
    cp = lambda qq, i: f"v_cvt_pk_bf16_f32 v{P + qq * 2 + i}, v{Z + qq * 4 + 2 * i}, v{Z + qq * 4 + 2 * i + 1}"
    cs = lambda qq, i: f"v_cvt_pk_bf16_f32 v{P + 4 + qq * 2 + i}, v{Z + 8 + qq * 4 + 2 * i}, v{Z + 8 + qq * 4 + 2 * i + 1}"
    g = [[ex(0, r)] for r in range(4)]
    g += [[ex(1, r), mu(0, r)] for r in range(4)]
    g += [[mu(1, 0), cp(0, 0)], [mu(1, 1), cp(0, 1)], [mu(1, 2), cs(0, 0)], [mu(1, 3), cs(0, 1)]]
    g += [[cp(1, 0), cp(1, 1)], [cs(1, 0), cs(1, 1)], [], []]
    return g


def row_reads(s2):
    R = V_ROW
    res = []
    for qq in (0, 1):
        qfi = 2 * s2 + qq
        res.append(f"ds_read_b128 {vr(R + (qq * 2 + 0) * 4)}, v{V_RA0} offset:{qfi * 2048}")
        res.append(f"ds_read_b128 {vr(R + (qq * 2 + 1) * 4)}, v{V_RA1} offset:{qfi * 2048}")
        res.append(f"ds_read_b128 {vr(R + 16 + (qq * 2 + 0) * 4)}, v{V_RA0} offset:{8192 + qfi * 2048}")
        res.append(f"ds_read_b128 {vr(R + 16 + (qq * 2 + 1) * 4)}, v{V_RA1} offset:{8192 + qfi * 2048}")
        res.append(f"ds_read_b128 {vr(R + 32 + qq * 4)}, v{V_RN} offset:{16384 + qfi * 64}")
        res.append(f"ds_read_b128 {vr(R + 40 + qq * 4)}, v{V_RN} offset:{16640 + qfi * 64}")
    return res


def tr_reads(s2):
    T = V_TR
    res = []
    for df in range(4):
        res.append(f"ds_read_b64_tr_b16 {vr(T + df * 4, 2)}, v{V_RT + df} offset:{8192 + s2 * 4096}")
        res.append(f"ds_read_b64_tr_b16 {vr(T + df * 4 + 2, 2)}, v{V_RT + df} offset:{8192 + s2 * 4096 + 2048}")
        res.append(f"ds_read_b64_tr_b16 {vr(T + 16 + df * 4, 2)}, v{V_RT + df} offset:{s2 * 4096}")
        res.append(f"ds_read_b64_tr_b16 {vr(T + 16 + df * 4 + 2, 2)}, v{V_RT + df} offset:{s2 * 4096 + 2048}")
    return res


def dma_atoms(last):
    P = OP
    atoms = [[f"s_add_u32 s{S_DW}, s{S_SLOTD}, s{S_WOFS}"]]
    if not last:
        atoms.append([f"s_mov_b32 m0, s{S_DW}", "s_nop 0", f"global_load_lds_dwordx4 %{P['voff_q0']}, s[{S_Q}:{S_Q + 1}]"])
        atoms.append([f"s_add_u32 s{S_B2}, s{S_Q}, 4096", f"s_addc_u32 s{S_B2 + 1}, s{S_Q + 1}, 0",
                      f"s_add_u32 s{S_TMP}, s{S_DW}, 4096", f"s_mov_b32 m0, s{S_TMP}", "s_nop 0",
                      f"global_load_lds_dwordx4 %{P['voff_q0']}, s[{S_B2}:{S_B2 + 1}]"])
        atoms.append([f"s_add_u32 s{S_TMP}, s{S_DW}, 8192", f"s_mov_b32 m0, s{S_TMP}", "s_nop 0",
                      f"global_load_lds_dwordx4 %{P['voff_g0']}, s[{S_G}:{S_G + 1}]"])
        atoms.append([f"s_add_u32 s{S_B2}, s{S_G}, s{S_P32}", f"s_addc_u32 s{S_B2 + 1}, s{S_G + 1}, 0",
                      f"s_add_u32 s{S_TMP}, s{S_DW}, {8192 + 4096}", f"s_mov_b32 m0, s{S_TMP}", "s_nop 0",
                      f"global_load_lds_dwordx4 %{P['voff_g0']}, s[{S_B2}:{S_B2 + 1}]"])
        vl = f"%{P['lane4']}"
        pre_l = []
    else:   # ragged last tile: rows past the end re-read the last valid row (offsets formed on the fly)
        for i in range(2):
            atoms.append([f"v_bfe_u32 v{V_TMP2}, %{P['rows_last']}, {8 * i}, 8", f"v_lshlrev_b32_e32 v{V_TMP}, 7, v{V_TMP2}",
                          f"v_add_u32_e32 v{V_TMP}, v{V_TMP}, %{P['chunk16']}",
                          f"s_add_u32 s{S_TMP}, s{S_DW}, {i * 4096}", f"s_mov_b32 m0, s{S_TMP}", "s_nop 0",
                          f"global_load_lds_dwordx4 v{V_TMP}, s[{S_Q}:{S_Q + 1}]"])
        for i in range(2):
            atoms.append([f"v_bfe_u32 v{V_TMP2}, %{P['rows_last']}, {8 * i}, 8", f"v_mul_lo_u32 v{V_TMP}, v{V_TMP2}, s{S_LDG}",
                          f"v_add_u32_e32 v{V_TMP}, v{V_TMP}, %{P['chunk16']}",
                          f"s_add_u32 s{S_TMP}, s{S_DW}, {8192 + i * 4096}", f"s_mov_b32 m0, s{S_TMP}", "s_nop 0",
                          f"global_load_lds_dwordx4 v{V_TMP}, s[{S_G}:{S_G + 1}]"])
        vl = f"v{V_TMP}"
        pre_l = [f"s_sub_u32 s{S_TMP2}, s{S_NVL}, 1", f"s_lshl_b32 s{S_TMP2}, s{S_TMP2}, 2",
                 f"v_min_u32_e32 v{V_TMP}, s{S_TMP2}, %{P['lane4']}"]
    # the two 256-byte rows of row constants: wave 0 fetches -lse, wave 1 -delta (the waits are vmcnt(0), so the waves need
    # not issue the same number of loads)
    uid = len(out) * 1000 + len(atoms) + (500 if last else 0)
    sel_l = [f"s_bitcmp0_b32 s{S_WOFS}, 10", f"s_cbranch_scc0 .Lnol_{uid}_%="] if RING4 else \
            [f"s_cmp_lg_u32 s{S_WOFS}, 0", f"s_cbranch_scc1 .Lnol_{uid}_%="]
    sel_d = [f"s_bitcmp1_b32 s{S_WOFS}, 10", f"s_cbranch_scc0 .Lnod_{uid}_%="] if RING4 else \
            [f"s_cmp_lg_u32 s{S_WOFS}, 1024", f"s_cbranch_scc1 .Lnod_{uid}_%="]
    atoms.append(sel_l + pre_l +
                 [f"s_add_u32 s{S_TMP}, s{S_SLOTD}, 16384", f"s_mov_b32 m0, s{S_TMP}", "s_nop 0",
                  f"global_load_lds_dword {vl}, s[{S_L}:{S_L + 1}]", f".Lnol_{uid}_%=:"])
    atoms.append(sel_d + pre_l +
                 [f"s_add_u32 s{S_TMP}, s{S_SLOTD}, 16640", f"s_mov_b32 m0, s{S_TMP}", "s_nop 0",
                  f"global_load_lds_dword {vl}, s[{S_D}:{S_D + 1}]", f".Lnod_{uid}_%=:"])
    atoms.append([f"s_add_u32 s{S_Q}, s{S_Q}, 8192", f"s_addc_u32 s{S_Q + 1}, s{S_Q + 1}, 0",
                  f"s_add_u32 s{S_G}, s{S_G}, s{S_T64}", f"s_addc_u32 s{S_G + 1}, s{S_G + 1}, 0"])
    atoms.append([f"s_add_u32 s{S_L}, s{S_L}, 256", f"s_addc_u32 s{S_L + 1}, s{S_L + 1}, 0",
                  f"s_add_u32 s{S_D}, s{S_D}, 256", f"s_addc_u32 s{S_D + 1}, s{S_D + 1}, 0"])
    atoms.append([f"s_add_u32 s{S_TD}, s{S_TD}, 1", f"s_add_u32 s{S_SLOTD}, s{S_SLOTD}, {SLOT}",
                  f"s_cmp_ge_u32 s{S_SLOTD}, s{S_RINGEND}", f"s_cselect_b32 s{S_SLOTD}, s{S_RING}, s{S_SLOTD}"])
    return atoms


def emit_dma(tag):
    e(f"s_add_u32 s{S_TMP2}, s{S_TD}, 1")
    e(f"s_cmp_eq_u32 s{S_TMP2}, s{S_NT}")
    e(f"s_cbranch_scc1 .Ldma_last_{tag}_%=")
    for a in dma_atoms(False):
        for i in a:
            e(i)
    e(f"s_branch .Ldma_done_{tag}_%=")
    e(f".Ldma_last_{tag}_%=:")
    for a in dma_atoms(True):
        for i in a:
            e(i)
    e(f".Ldma_done_{tag}_%=:")


def emit_fixup(tag, plus):
    e(f"s_add_u32 s{S_TMP2}, s{S_T}, {plus + 1}")
    e(f"s_cmp_eq_u32 s{S_TMP2}, s{S_NT}")
    e(f"s_cbranch_scc0 .Lfix_skip_{tag}_%=")
    e(f"s_cmp_lt_u32 s{S_NVL}, 64")
    e(f"s_cbranch_scc0 .Lfix_skip_{tag}_%=")
    e(f"v_lshrrev_b32_e32 v{V_TMP}, 2, %{OP['lane4']}")
    e(f"v_cmp_ge_u32_e64 s[{S_EXEC}:{S_EXEC + 1}], v{V_TMP}, s{S_NVL}")
    e(f"s_and_saveexec_b64 s[{S_SAVE}:{S_SAVE + 1}], s[{S_EXEC}:{S_EXEC + 1}]")
    e(f"v_add_u32_e32 v{V_TMP}, s{S_TMP}, %{OP['lane4']}")
    e(f"v_mov_b32_e32 v{V_TMP2}, 0xff800000")
    e(f"ds_write_b32 v{V_TMP}, v{V_TMP2} offset:16384")
    e(f"s_mov_b64 exec, s[{S_SAVE}:{S_SAVE + 1}]")
    e("s_waitcnt lgkmcnt(0)")
    e(f".Lfix_skip_{tag}_%=:")


def next_slot(dst, src):
    e(f"s_add_u32 s{dst}, s{src}, {SLOT}")
    e(f"s_cmp_ge_u32 s{dst}, s{S_RINGEND}")
    e(f"s_cselect_b32 s{dst}, s{S_RING}, s{dst}")


def set_row_addresses():
    e(f"v_add_u32_e32 v{V_RA0}, s{S_SLOTC}, %{OP['lds_a0']}")
    e(f"v_add_u32_e32 v{V_RA1}, s{S_SLOTC}, %{OP['lds_a1']}")
    e(f"v_add_u32_e32 v{V_RN}, s{S_SLOTC}, %{OP['lds_rn']}")


def tr_address_atoms():
    return [f"v_add_u32_e32 v{V_RT + i}, s{S_SLOTT}, %{OP['lds_tr0'] + i}" for i in range(4)]


def place(mf, va, after=None, dma=(), v_from=0, v_keep=3, dma_keep=0):
    """Emit the MFMAs of a step with its VALU work spread over the gaps from MFMA v_from on; after[i] = instructions that
    follow MFMA i at once (fragment reads behind the last MFMA that uses the registers they overwrite); DMA atoms spread
    over all gaps."""
    after = after or {}
    nm = max(len(mf), 1)
    sched = None
    if va and isinstance(va[0], list):     # per-gap schedule for 16 MFMAs (v_sched); fewer MFMAs: neighbouring gaps merged
        if len(mf) == 16:
            sched = va
        elif len(mf) == 8:
            sched = [va[2 * i] + va[2 * i + 1] for i in range(8)]
        va = [] if sched else [x for gp in va for x in gp]
    if MFMA32:
        v_keep = 1
    nv = max(nm - v_from - v_keep, 1)     # the last v_keep gaps stay free: the next step's first MFMAs read what V packs
    vi = di = 0
    dma = list(dma)
    for i, m in enumerate(mf):
        e(m)
        for ins in after.get(i, []):
            e(ins)
        if sched:
            for ins in sched[i]:
                e(ins)
        if i >= v_from:
            want = min(len(va), (len(va) * (i - v_from + 1) + nv - 1) // nv)
            while vi < want:
                e(va[vi])
                vi += 1
        wd = min(len(dma), (len(dma) * (i + 1) + nm - dma_keep - 1) // max(nm - dma_keep, 1)) if dma_keep else (len(dma) * (i + 1)) // nm
        while di < wd:
            for ins in dma[di]:
                e(ins)
            di += 1
    if vi < len(va):   # fewer MFMAs than planned (probe builds, drained steps): the rest follows, padded against the packs' readers
        while vi < len(va):
            e(va[vi])
            vi += 1
        e("s_nop 3")
    while di < len(dma):
        for ins in dma[di]:
            e(ins)
        di += 1
    for k in sorted(after):
        if k >= len(mf):
            for ins in after[k]:
                e(ins)


def stamp_sums():
    return [f"s_sub_u32 s{S_TMP2}, s{S_SB + 2}, s{S_SB}", f"s_add_u32 s{S_ACC_VM}, s{S_ACC_VM}, s{S_TMP2}",
            f"s_sub_u32 s{S_TMP2}, s{S_SB + 4}, s{S_SB + 2}", f"s_add_u32 s{S_ACC_BAR}, s{S_ACC_BAR}, s{S_TMP2}"]


def step(j, do_m1=True, do_m2=True, row_next=None, tr_this=None, dma=(), pre_rows=(), dma_after=None, sums=False):
    """Step j (0..3) of a tile: group (half = j // 2, key fragment f = j % 2).  Order inside a step: M2(g-1) first - its last
    MFMA frees the transposed fragments, whose successors are requested at once - then M1(g+1), behind whose last MFMA the
    next row fragments are requested; V(g) fills the gaps from the 4th MFMA on (its inputs come from the M1 products at the
    end of the previous step).
    row_next: half index (0/1) whose row fragments are requested behind M1 (None: none);  tr_this: half whose transposed
    fragments are requested behind M2;  pre_rows: instructions between M1 and the row reads (the barrier block)."""
    f = j % 2
    m1 = m1_atoms((j + 1) % 2, (j + 1) % 2) if (do_m1 and "nom1" not in PROBE) else []
    m2 = m2_atoms((j - 1) % 2, (j - 1) % 2) if (do_m2 and "nom2" not in PROBE) else []
    va = [] if "novalu" in PROBE else (v_sched(j % 2, j % 2) if "vsched" in OPT else v_atoms(j % 2, j % 2))
    nolds = "nolds" in PROBE
    mf = m2 + m1
    after = {}
    if f == 1:
        # fragment reads of the previous step: 16 transposed (needed by M2 now), then 12 rows (needed by M1)
        if m2:
            e("s_waitcnt lgkmcnt(12)")
        if m1:
            extra = stamp_sums() if (sums and INLOOP) else []
            if m2:
                after[len(m2) - 1] = ["s_waitcnt lgkmcnt(0)"] + extra
            else:
                e("s_waitcnt lgkmcnt(0)")
                for ins in extra:
                    e(ins)
        for k, atom in (dma_after or {}).items():
            kk = min(k, len(mf) - 1) if mf else 0
            after[kk] = after.get(kk, []) + list(atom)
        place(mf, va, after, dma, v_from=1)
        return
    k2 = len(m2) - 1
    if tr_this is not None and not nolds and "spreadtr" in OPT and k2 == 7:
        # the transposed fragments of d-block df are free once M2's MFMA pair df has issued: request their successors there
        # instead of all 16 behind the last pair (same order, so the counted waits of the next step hold)
        rd = tr_reads(tr_this)
        if tr_this == 0:       # (both halves of a tile read the same slot: the addresses of step 0 serve step 2)
            after[0] = after.get(0, []) + tr_address_atoms()
        for df in range(4):
            after[2 * df + 1] = after.get(2 * df + 1, []) + rd[4 * df:4 * df + 4]
    elif tr_this is not None and not nolds:
        blk = (tr_address_atoms() if tr_this == 0 else []) + tr_reads(tr_this)
        if k2 >= 0:
            after[k2] = after.get(k2, []) + blk
        else:
            for ins in blk:
                e(ins)
    blk = list(pre_rows)
    if row_next is not None and not nolds:
        rr = row_reads(row_next)
        if "earlyrows" in OPT and not pre_rows and m1:
            # fragments that only the ks0 products read (a[qq][0], gg[qq][0], -lse, -delta) go out behind the 4th M1 MFMA
            early = [r for i, r in enumerate(rr) if i % 6 in (0, 2, 4, 5)]
            late = [r for i, r in enumerate(rr) if i % 6 in (1, 3)]
            ke = len(m2) + 3
            after[ke] = after.get(ke, []) + early
            blk += late
        else:
            blk += rr
    k1 = len(mf) - 1
    for k, atom in (dma_after or {}).items():
        kk = min(k, len(mf) - 1) if mf else 0
        after[kk] = after.get(kk, []) + list(atom)
    if blk:
        if k1 >= 0:
            after[k1] = after.get(k1, []) + blk
        else:
            for ins in blk:
                e(ins)
    place(mf, va, after, dma, v_from=1, dma_keep=1 if (pre_rows and dma) else 0)


def barrier_block(tag, inflight=0):
    """Tile t+1 has landed (all vector-memory work but the `inflight` youngest operations - the fetch of tile t+2 with the
    4-slot ring - is its DMA) and every wave is done with tile t-1."""
    save = out[:]
    del out[:]
    if INLOOP:
        e(f"s_memtime s[{S_SB}:{S_SB + 1}]")
    e(f"s_waitcnt vmcnt({inflight})")
    if INLOOP:
        e(f"s_memtime s[{S_SB + 2}:{S_SB + 3}]")
    next_slot(S_TMP, S_SLOTC)
    emit_fixup("loop" + tag, 1)
    e("s_barrier")
    if INLOOP:
        e(f"s_memtime s[{S_SB + 4}:{S_SB + 5}]")
    next_slot(S_SLOTC, S_SLOTC)
    set_row_addresses()
    blk = out[:]
    del out[:]
    out.extend(save)
    return blk


def second_half(variant):
    atoms = [] if variant == "N" else dma_atoms(variant == "L")
    e(f"; step 2 ({variant})")
    # the barrier sits behind the last MFMA of step 2; the fetch of tile t+2 overwrites the slot of tile t-1, so it goes out
    # behind it: all of it in the gaps of step 3
    step(2, row_next=0, tr_this=1, pre_rows=barrier_block(variant))
    e(f"; step 3 ({variant})")
    step(3, dma=atoms, sums=True)


def tile_ring4(variant):
    """One whole tile with the 4-slot ring: the fetch of tile t+2 (variant F: a full tile, L: the ragged last one, N: none left)
    goes out over steps 0..2 - ahead of this tile's barrier: its slot is that of tile t-2 - and the barrier waits for all but
    those NDMA_WAVE operations."""
    atoms = [] if variant == "N" else dma_atoms(variant == "L")
    n = len(atoms)
    c0, c1, c2 = atoms[:(n + 2) // 3], atoms[(n + 2) // 3:(2 * n + 2) // 3], atoms[(2 * n + 2) // 3:]
    e(f"; tile ({variant}) step 0")
    step(0, row_next=1, tr_this=0, dma=c0)
    e(f"; step 1")
    step(1, dma=c1)
    e(f"; step 2")
    step(2, row_next=0, tr_this=1, dma=c2, pre_rows=barrier_block(variant, 0 if variant == "N" else NDMA_WAVE))
    e(f"; step 3")
    step(3, sums=True)


NDMA_WAVE = 5   # vector-memory operations per wave per tile with ring4 (4 x 1 KiB of Q / dO + one row-constant load)


def emit():
    P = OP
    e(f"s_mov_b32 s{S_M0}, m0")
    stamp(0)
    if STAMPS:
        for r in (S_ACC_VM, S_ACC_BAR):
            e(f"s_mov_b32 s{r}, 0")
    e(f"s_mov_b64 s[{S_Q}:{S_Q + 1}], %{P['qbase']}")
    e(f"s_mov_b64 s[{S_G}:{S_G + 1}], %{P['gbase']}")
    e(f"s_mov_b64 s[{S_L}:{S_L + 1}], %{P['lbase']}")
    e(f"s_mov_b64 s[{S_D}:{S_D + 1}], %{P['dbase']}")
    e(f"s_mov_b32 s{S_NT}, %{P['nt']}")
    e(f"s_mov_b32 s{S_LDG}, %{P['ldg']}")
    e(f"s_mov_b32 s{S_RING}, %{P['ring']}")
    e(f"s_mov_b32 s{S_NVL}, %{P['nvalid_last']}")
    e(f"s_mov_b32 s{S_FLAGS}, %{P['active']}")
    e(f"s_mov_b32 s{S_WOFS}, %{P['wofs']}")
    e(f"s_lshl_b32 s{S_P32}, s{S_LDG}, 5")
    e(f"s_lshl_b32 s{S_T64}, s{S_LDG}, 6")
    e(f"s_add_u32 s{S_RINGEND}, s{S_RING}, {NSLOT * SLOT}")
    e(f"s_mov_b32 s{S_T}, 0")
    e(f"s_mov_b32 s{S_TD}, 0")
    e(f"s_mov_b32 s{S_SLOTC}, s{S_RING}")
    e(f"s_mov_b32 s{S_SLOTT}, s{S_RING}")
    e(f"s_mov_b32 s{S_SLOTD}, s{S_RING}")
    emit_dma("p0")
    for f in range(2):
        for ks in range(2):
            e(f"global_load_dwordx4 {ar(A_KF + (f * 2 + ks) * 4)}, %{P['kvoff0'] + f}, %{P['kbase']} offset:{ks * 64}")
            e(f"global_load_dwordx4 {ar(A_VF + (f * 2 + ks) * 4)}, %{P['kvoff0'] + f}, %{P['vbase']} offset:{ks * 64}")
    e(f"s_cmp_lt_u32 s{S_TD}, s{S_NT}")
    e("s_cbranch_scc0 .Lno_second_%=")
    emit_dma("p1")
    e(".Lno_second_%=:")
    # zero: accumulators; the transposed fragments and both packed sets (group -1 multiplies zeros)
    for i in range(64):
        e(f"v_accvgpr_write_b32 a{A_DK + i}, 0")
    for i in range(32):
        e(f"v_mov_b32_e32 v{V_TR + i}, 0")
    for i in range(16):
        e(f"v_mov_b32_e32 v{V_P + i}, 0")
    e("s_waitcnt vmcnt(0)")
    stamp(1)
    e(f"s_mov_b32 s{S_TMP}, s{S_RING}")
    emit_fixup("pro", 0)
    e("s_barrier")
    stamp(2)
    if STAMPS:
        e("s_waitcnt lgkmcnt(0)")
        for i in range(3):
            e(f"s_mov_b32 s{S_T0 + i}, s{S_SB + 2 * i}")
    e(f"s_cmp_eq_u32 s{S_FLAGS}, 0")
    e("s_cbranch_scc1 .Lfeed_only_%=")
    set_row_addresses()
    for r in row_reads(0):
        e(r)
    e("s_waitcnt lgkmcnt(0)")
    for m in m1_atoms(0, 0):
        e(m)
    if "prio" in OPT:
        e("s_setprio 1")       # the tile loop outranks a partner wave that is in its prologue / epilogue
    e(f"s_add_u32 s{S_TMP}, s{S_T}, 1")
    e(f"s_cmp_lt_u32 s{S_TMP}, s{S_NT}")
    e("s_cbranch_scc0 .Llast_tile_%=")
    e(".Ltile_loop_%=:")
    if RING4:
        e(f"s_cmp_lt_u32 s{S_TD}, s{S_NT}")
        e("s_cbranch_scc0 .Lt_none_%=")
        e(f"s_add_u32 s{S_TMP2}, s{S_TD}, 1")
        e(f"s_cmp_eq_u32 s{S_TMP2}, s{S_NT}")
        e("s_cbranch_scc1 .Lt_last_%=")
        tile_ring4("F")
        e("s_branch .Lh2_done_%=")
        e(".Lt_last_%=:")
        tile_ring4("L")
        e("s_branch .Lh2_done_%=")
        e(".Lt_none_%=:")
        tile_ring4("N")
        e(".Lh2_done_%=:")
    else:
        e("; step 0")
        step(0, row_next=1, tr_this=0)
        e("; step 1")
        step(1)
    if RING4:
        pass
    elif "nodma" in PROBE:
        second_half("N")
    else:
        e(f"s_cmp_lt_u32 s{S_TD}, s{S_NT}")
        e("s_cbranch_scc0 .Lh2_none_%=")
        e(f"s_add_u32 s{S_TMP2}, s{S_TD}, 1")
        e(f"s_cmp_eq_u32 s{S_TMP2}, s{S_NT}")
        e("s_cbranch_scc1 .Lh2_last_%=")
        second_half("F")
        e("s_branch .Lh2_done_%=")
        e(".Lh2_last_%=:")
        second_half("L")
        e("s_branch .Lh2_done_%=")
        e(".Lh2_none_%=:")
        second_half("N")
        e(".Lh2_done_%=:")
    e(f"s_mov_b32 s{S_SLOTT}, s{S_SLOTC}")        # the next tile's transposed fragments come from the slot just switched to
    e(f"s_add_u32 s{S_T}, s{S_T}, 1")
    e(f"s_add_u32 s{S_TMP}, s{S_T}, 1")
    e(f"s_cmp_lt_u32 s{S_TMP}, s{S_NT}")
    e("s_cbranch_scc1 .Ltile_loop_%=")
    e(".Llast_tile_%=:")
    e(f"s_cmp_le_u32 s{S_NVL}, 32")
    e("s_cbranch_scc0 .Llast_full_%=")
    e("; short last tile: its second half contributes exactly nothing (p = 0)")
    step(0, row_next=None, tr_this=0)
    step(1, do_m1=False)
    for m in m2_atoms(1, 1):
        e(m)
    e("s_branch .Ldrained_%=")
    e(".Llast_full_%=:")
    step(0, row_next=1, tr_this=0)
    step(1)
    step(2, row_next=None, tr_this=1)
    step(3, do_m1=False)
    for m in m2_atoms(1, 1):
        e(m)
    e(".Ldrained_%=:")
    if "prio" in OPT:
        e("s_setprio 0")
    stamp(3)
    e("s_nop 15")
    e("s_nop 15")
    e("s_barrier")
    stamp(4)
    for i in range(16):
        e(f"ds_write_b128 %{P['dump']}, {ar(i * 4)} offset:{i * 1024}")
    e("s_waitcnt lgkmcnt(0)")
    if STAMPS:
        stamp(5)
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_mov_b64 s[{S_SAVE}:{S_SAVE + 1}], exec")
        e("s_mov_b64 exec, 1")
        e(f"s_getreg_b32 s{S_TMP}, hwreg(HW_REG_HW_ID)")          # wave 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13
        e(f"s_getreg_b32 s{S_TMP2}, hwreg(HW_REG_XCC_ID)")
        e(f"s_and_b32 s{S_TMP}, s{S_TMP}, 0xffff")
        e(f"s_and_b32 s{S_TMP2}, s{S_TMP2}, 15")
        e(f"s_lshl_b32 s{S_TMP2}, s{S_TMP2}, 16")
        e(f"s_or_b32 s{S_TMP}, s{S_TMP}, s{S_TMP2}")
        srcs = [S_T0, S_T0 + 1, S_T0 + 2, S_SB, S_SB + 2, S_SB + 4, S_ACC_VM, S_ACC_BAR, S_T, S_TMP, S_SB + 1, S_SB + 5]
        for i, r in enumerate(srcs):
            e(f"v_mov_b32_e32 v{V_ROW + i}, s{r}")
        e(f"v_mov_b32_e32 v{V_ROW + 15}, 0")
        for i in range(3):
            e(f"global_store_dwordx4 v{V_ROW + 15}, {vr(V_ROW + 4 * i)}, %{OP['dump'] + 1} offset:{16 * i}")
        e("s_waitcnt vmcnt(0)")
        e(f"s_mov_b64 exec, s[{S_SAVE}:{S_SAVE + 1}]")
    e("s_branch .Lend_%=")
    e(".Lfeed_only_%=:")
    e(f"s_add_u32 s{S_TMP}, s{S_T}, 1")
    e(f"s_cmp_lt_u32 s{S_TMP}, s{S_NT}")
    e("s_cbranch_scc0 .Lfeed_done_%=")
    e(".Lfeed_loop_%=:")
    e("s_waitcnt vmcnt(0)")
    next_slot(S_TMP, S_SLOTC)
    emit_fixup("feed", 1)
    e("s_barrier")
    next_slot(S_SLOTC, S_SLOTC)
    e(f"s_cmp_lt_u32 s{S_TD}, s{S_NT}")
    e("s_cbranch_scc0 .Lfeed_nodma_%=")
    emit_dma("feed")
    e(".Lfeed_nodma_%=:")
    e(f"s_add_u32 s{S_T}, s{S_T}, 1")
    e(f"s_add_u32 s{S_TMP}, s{S_T}, 1")
    e(f"s_cmp_lt_u32 s{S_TMP}, s{S_NT}")
    e("s_cbranch_scc1 .Lfeed_loop_%=")
    e(".Lfeed_done_%=:")
    e("s_barrier")
    e(".Lend_%=:")
    e(f"s_mov_b32 m0, s{S_M0}")


emit()
print("// GENERATED by gen/gen_attn_dkv32_asm.py - do not edit (regenerate: make -C nvit_amd/csrc gen)")
NAME = "NVIT_ATTN_DKV32_STAMPS" if STAMPS else "NVIT_ATTN_DKV32_ASM"
print(f"#define {NAME}_BODY \\")
for line in out:
    print(f'  "{line}\\n\\t" \\')
print('  ""')
clob = [f'"v{i}"' for i in list(range(16, 25)) + list(range(V_ROW, V_END))] + [f'"a{i}"' for i in range(A_END)] + [f'"s{i}"' for i in range(40, S_STAMP_END if STAMPS else 72)] + ['"vcc"', '"memory"']
print(f"#define {NAME}_CLOBBERS " + ", ".join(clob))
print(f"// instructions: {sum(1 for l in out if not l.startswith(';') and not l.endswith(':'))}")
