#!/usr/bin/env python3
"""Generator of the hand-placed main loop of the attention backward dK/dV kernel (gfx950), PING-PONG form: a 512-thread
workgroup = two independent halves of 4 waves x 32 keys (each half its own (batch, head, key block), its own tile ring);
the two waves of a SIMD belong to different halves and alternate, separated by workgroup barriers, between a COMPUTE
segment (32 MFMA with the softmax-backward VALU work in their gaps, operands in registers) and a LOAD segment (the 28
fragment reads of the next half tile, the LDS-DMA pieces of the tile after next) - matrix work beside memory work.
Measured reason (in-kernel stamps): two matrix-heavy waves on one SIMD do not overlap - with two
independent 4-wave workgroups per CU the two tile loops take the time of running them one after the other.

Writes attn_dkv_pp_asm.inc: ONE inline-asm string (the whole tile loop of attn_bwd_dkv_pp_kernel in attn_mfma.hip) plus its
clobber list.  Arithmetic, operand layouts and accumulation order are those of the compiler-built attn_bwd_dkv_mfma_kernel
(reference semantics: the dK/dV half of the backward of F.scaled_dot_product_attention as the reference calls it,
/root/reference/nvit/model.py:121-124); bit-exact against it.

Program of a half (t = tile, h = 32-query half, f = key fragment; groups g = (h, f) in the order (0,0) (0,1) (1,0) (1,1)):
  step 0: M2(prev tile (1,1)) V(0,0) M1(0,1) | BAR | LOAD a: transposed(t,0) rows(t,1) 3 DMA pieces | BAR |
  step 1: M2(0,0) V(0,1) M1(1,0);  step 2: M2(0,1) V(1,0) M1(1,1);  wait for tile t+1 | BAR |
  LOAD b: transposed(t,1) rows(t+1,0) 2 DMA pieces | BAR | step 3: M2(1,0) V(1,1) M1(next (0,0))
The second half runs the same program one barrier later, so its compute segments (steps 3+0, steps 1+2) fall beside the
first half's load segments and vice versa.

MEASURED, NOT ADOPTED (round 4).  Bit-exact on every shape of the GPU test, but 1.37 M cycles per launch against the 1.14 M
of the product form (two independent 4-wave workgroups per CU): the loop runs 3 195 cycles per tile PAIR (MFMA alone 2 210,
VALU +692, LDS reads +379, DMA +158: profiles/r04_dkv_cycles_pingpong_probes.log) against 2 x 1 975 for two stand-alone
waves, i.e. the alternation buys 19 %, and the workgroup's prologue and epilogue (36 % of its life) are no longer covered by
a partner workgroup.  The kernel that used this loop (attn_bwd_dkv_pp_kernel, mode 2 of nvit_set_attn_dkv_asm) is in
commit 240ec07 of this repository: nvit_amd/csrc/attn_mfma.hip there; this generator wrote its attn_dkv_pp_asm.inc.

usage (at that commit): python3 gen_attn_dkv_pp_asm.py > ../../nvit_amd/csrc/attn_dkv_pp_asm.inc
"""
import os

PROBE = set(filter(None, os.environ.get("GEN_PROBE", "").split(",")))
OPT = set(filter(None, os.environ.get("GEN_OPT", "").split(",")))   # tuning experiments (results stay exact)

SLOT = 2 * 8192 + 512      # Q tile | dO tile | -lse[64] | -delta[64]   (= DKV_SLOT of attn_mfma.hip)
RING4 = True   # every wave issues exactly one of the two row-constant loads     # 4-slot ring: the fetch of tile t+2 is spread over steps 0..2 of tile t (it may overwrite the slot of
                           # tile t-2 before this tile's barrier), instead of bunched into step 3 behind the barrier
NSLOT = 3
NDMA = 6                   # DMA wave-instructions per wave per tile

OP = dict(qbase=0, gbase=1, lbase=2, dbase=3, kbase=4, vbase=5, nt=6, ldg=7, ring=8, nvalid_last=9, active=10, wofs=11,
          voff_q0=12, voff_g0=13, rows_last=14, chunk16=15, lane4=16, kvoff0=17, lds_pack0=19, dump=23)

S_Q, S_G, S_L, S_D = 40, 42, 44, 46
S_NT, S_LDG, S_RING, S_NVL = 48, 49, 50, 51
S_T, S_TD, S_SLOTC, S_SLOTD = 52, 53, 54, 55
S_TMP, S_TMP2 = 56, 57
S_EXEC = 58
S_P32, S_T64, S_WOFS, S_RINGEND = 60, 61, 62, 63
S_SAVE = 64
S_FLAGS, S_DW, S_M0, S_SLOTT = 66, 67, 68, 69     # S_SLOTT: LDS base of the tile whose transposed fragments are being read

# per-lane operands (voff_q0, voff_g0, rows_last, chunk16, lane4, kvoff, packed LDS offsets, dump) are read straight from the
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


def m1_atoms(fn, zbuf):
    R, Z = V_ROW, V_Z + 16 * zbuf
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



def emit_seq(atoms):
    for a in atoms:
        for i in a:
            e(i)


def emit_fixup(tag, plus):
    """-inf into the -lse entries of the rows past the end of the ragged last tile, if tile S_T + plus is that tile
    (slot base in S_TMP); every wave does it behind its own vmcnt wait, before the barrier that publishes the tile."""
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


S_SLOTN = S_SLOTC          # slot of tile t+1 (rows of the next tile's first half are read from it)
# S_SLOTT: slot of tile t


def row_addresses(slot):
    p0, p1 = OP['lds_pack0'], OP['lds_pack0'] + 1
    e(f"v_and_b32_e32 v{V_TMP}, 0xffff, %{p0}")
    e(f"v_add_u32_e32 v{V_RA0}, s{slot}, v{V_TMP}")
    e(f"v_lshrrev_b32_e32 v{V_TMP}, 16, %{p0}")
    e(f"v_add_u32_e32 v{V_RA1}, s{slot}, v{V_TMP}")
    e(f"v_and_b32_e32 v{V_TMP}, 0xffff, %{p1}")
    e(f"v_add_u32_e32 v{V_RN}, s{slot}, v{V_TMP}")


def tr_addresses():
    p1, p2, p3 = OP['lds_pack0'] + 1, OP['lds_pack0'] + 2, OP['lds_pack0'] + 3
    for ins in [f"v_lshrrev_b32_e32 v{V_TMP}, 16, %{p1}", f"v_add_u32_e32 v{V_RT}, s{S_SLOTT}, v{V_TMP}",
                f"v_and_b32_e32 v{V_TMP}, 0xffff, %{p2}", f"v_add_u32_e32 v{V_RT + 1}, s{S_SLOTT}, v{V_TMP}",
                f"v_lshrrev_b32_e32 v{V_TMP}, 16, %{p2}", f"v_add_u32_e32 v{V_RT + 2}, s{S_SLOTT}, v{V_TMP}",
                f"v_add_u32_e32 v{V_RT + 3}, s{S_SLOTT}, %{p3}"]:
        e(ins)


NOLDS = "nolds" in PROBE
NODMA = "nodma" in PROBE
N_A = 3    # vector-memory operations of a tile's fetch issued in load segment a (the other 2 in b)


def bar():
    e("s_barrier")


def compute_step(j, do_m1=True, do_m2=True):
    m1 = m1_atoms((j + 1) % 2, (j + 1) % 2) if (do_m1 and "nom1" not in PROBE) else []
    m2 = m2_atoms((j - 1) % 2, (j - 1) % 2) if (do_m2 and "nom2" not in PROBE) else []
    va = [] if "novalu" in PROBE else (v_sched(j % 2, j % 2) if "vsched" in OPT else v_atoms(j % 2, j % 2))
    place(m2 + m1, va, v_from=1)


def load_a(atoms, rows=True):
    """transposed fragments of (t, half 0); row fragments of (t, half 1); the first pieces of the fetch of tile t+2"""
    if not NOLDS:
        tr_addresses()
        rr = tr_reads(0)
        if rows:
            row_addresses(S_SLOTT)
            rr = rr + row_reads(1)
    else:
        rr = []
    interleave(rr, atoms)


def load_b(atoms, rows=True):
    """transposed fragments of (t, half 1); row fragments of (t+1, half 0); the rest of the fetch of tile t+2"""
    if not NOLDS:
        rr = tr_reads(1)          # (addresses: those of load segment a, same tile)
        if rows:
            row_addresses(S_SLOTN)
            rr = rr + row_reads(0)
    else:
        rr = []
    interleave(rr, atoms)


def interleave(reads, atoms):
    """LDS reads with the DMA atoms spread between them"""
    atoms = list(atoms)
    n, k = len(reads), len(atoms)
    di = 0
    for i, r in enumerate(reads):
        e(r)
        want = (k * (i + 1)) // max(n, 1)
        while di < want:
            for ins in atoms[di]:
                e(ins)
            di += 1
    while di < k:
        for ins in atoms[di]:
            e(ins)
        di += 1


def split_atoms(atoms):
    """(segment a, segment b): head + the first N_A vector-memory pieces | the rest + the cursor updates"""
    if not atoms:
        return [], []
    return atoms[:1 + N_A], atoms[1 + N_A:]


def wait_next_tile(tag, inflight):
    """end of steps 1+2: tile t+1 must have landed (this wave's pieces of it) before the barrier that lets load segment b
    read it; `inflight` = pieces of tile t+2 already issued behind them"""
    e(f"s_waitcnt vmcnt({inflight})")
    e(f"s_mov_b32 s{S_TMP}, s{S_SLOTN}")
    emit_fixup(tag, 1)


def tile(variant):
    atoms = [] if (variant == "N" or NODMA) else dma_atoms(variant == "L")
    ca, cb = split_atoms(atoms)
    e(f"; ---- tile ({variant}): step 0")
    compute_step(0)
    bar()
    load_a(ca)
    bar()
    e("s_waitcnt lgkmcnt(0)")
    compute_step(1)
    compute_step(2)
    wait_next_tile("t" + variant, N_A if atoms else 0)
    bar()
    load_b(cb)
    bar()
    e("s_waitcnt lgkmcnt(0)")
    compute_step(3)


def feed_tile(variant):
    atoms = [] if (variant == "N" or NODMA) else dma_atoms(variant == "L")
    ca, cb = split_atoms(atoms)
    bar()
    emit_seq(ca)
    bar()
    wait_next_tile("f" + variant, N_A if atoms else 0)
    bar()
    emit_seq(cb)
    bar()


def variants(fn, tag):
    """three copies of a tile body, by what is left to fetch: a full tile t+2 (F), the ragged last one (L), nothing (N)"""
    if NODMA:
        fn("N")
        return
    e(f"s_cmp_lt_u32 s{S_TD}, s{S_NT}")
    e(f"s_cbranch_scc0 .L{tag}_none_%=")
    e(f"s_add_u32 s{S_TMP2}, s{S_TD}, 1")
    e(f"s_cmp_eq_u32 s{S_TMP2}, s{S_NT}")
    e(f"s_cbranch_scc1 .L{tag}_last_%=")
    fn("F")
    e(f"s_branch .L{tag}_done_%=")
    e(f".L{tag}_last_%=:")
    fn("L")
    e(f"s_branch .L{tag}_done_%=")
    e(f".L{tag}_none_%=:")
    fn("N")
    e(f".L{tag}_done_%=:")


def emit_dma(tag):
    e(f"s_add_u32 s{S_TMP2}, s{S_TD}, 1")
    e(f"s_cmp_eq_u32 s{S_TMP2}, s{S_NT}")
    e(f"s_cbranch_scc1 .Ldma_last_{tag}_%=")
    emit_seq(dma_atoms(False))
    e(f"s_branch .Ldma_done_{tag}_%=")
    e(f".Ldma_last_{tag}_%=:")
    emit_seq(dma_atoms(True))
    e(f".Ldma_done_{tag}_%=:")


def advance_tile():
    e(f"s_mov_b32 s{S_SLOTT}, s{S_SLOTN}")
    next_slot(S_SLOTN, S_SLOTN)
    e(f"s_add_u32 s{S_T}, s{S_T}, 1")


def emit():
    P = OP
    e(f"s_mov_b32 s{S_M0}, m0")
    stamp(0)
    e(f"s_mov_b32 s{S_NT}, %{P['nt']}")
    e(f"s_mov_b32 s{S_FLAGS}, %{P['active']}")     # bit 0: this wave has keys; bit 1: this half has a work item; bit 2: second half
    e(f"s_bitcmp1_b32 s{S_FLAGS}, 1")
    e("s_cbranch_scc0 .Lidle_half_%=")
    e(f"s_mov_b64 s[{S_Q}:{S_Q + 1}], %{P['qbase']}")
    e(f"s_mov_b64 s[{S_G}:{S_G + 1}], %{P['gbase']}")
    e(f"s_mov_b64 s[{S_L}:{S_L + 1}], %{P['lbase']}")
    e(f"s_mov_b64 s[{S_D}:{S_D + 1}], %{P['dbase']}")
    e(f"s_mov_b32 s{S_LDG}, %{P['ldg']}")
    e(f"s_mov_b32 s{S_RING}, %{P['ring']}")
    e(f"s_mov_b32 s{S_NVL}, %{P['nvalid_last']}")
    e(f"s_mov_b32 s{S_WOFS}, %{P['wofs']}")
    e(f"s_lshl_b32 s{S_P32}, s{S_LDG}, 5")
    e(f"s_lshl_b32 s{S_T64}, s{S_LDG}, 6")
    e(f"s_add_u32 s{S_RINGEND}, s{S_RING}, {NSLOT * SLOT}")
    e(f"s_mov_b32 s{S_T}, 0")
    e(f"s_mov_b32 s{S_TD}, 0")
    e(f"s_mov_b32 s{S_SLOTT}, s{S_RING}")
    e(f"s_add_u32 s{S_SLOTN}, s{S_RING}, {SLOT}")
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
    bar()                                            # barrier 0: tiles 0 and 1 are in place
    stamp(2)
    if STAMPS:
        e("s_waitcnt lgkmcnt(0)")
        for i in range(3):
            e(f"s_mov_b32 s{S_T0 + i}, s{S_SB + 2 * i}")
    e(f"s_bitcmp1_b32 s{S_FLAGS}, 0")
    e("s_cbranch_scc0 .Lfeed_only_%=")
    row_addresses(S_SLOTT)
    for r in row_reads(0):
        e(r)
    e("s_waitcnt lgkmcnt(0)")
    for m in m1_atoms(0, 0):
        e(m)
    e(f"s_bitcmp1_b32 s{S_FLAGS}, 2")                # the second half runs one barrier behind the first
    e("s_cbranch_scc0 .Lno_shift_%=")
    bar()
    e(".Lno_shift_%=:")
    e(f"s_add_u32 s{S_TMP}, s{S_T}, 1")
    e(f"s_cmp_lt_u32 s{S_TMP}, s{S_NT}")
    e("s_cbranch_scc0 .Llast_tile_%=")
    e(".Ltile_loop_%=:")
    variants(tile, "t")
    advance_tile()
    e(f"s_add_u32 s{S_TMP}, s{S_T}, 1")
    e(f"s_cmp_lt_u32 s{S_TMP}, s{S_NT}")
    e("s_cbranch_scc1 .Ltile_loop_%=")
    e(".Llast_tile_%=:")
    e(f"s_cmp_le_u32 s{S_NVL}, 32")
    e("s_cbranch_scc0 .Llast_full_%=")
    e("; short last tile: its second half contributes exactly nothing (p = 0)")
    compute_step(0)
    bar()
    load_a([], rows=False)
    bar()
    e("s_waitcnt lgkmcnt(0)")
    compute_step(1, do_m1=False)
    for m in m2_atoms(1, 1):
        e(m)
    bar()
    bar()
    e("s_branch .Ldrained_%=")
    e(".Llast_full_%=:")
    compute_step(0)
    bar()
    load_a([])
    bar()
    e("s_waitcnt lgkmcnt(0)")
    compute_step(1)
    compute_step(2)
    bar()
    load_b([], rows=False)
    bar()
    e("s_waitcnt lgkmcnt(0)")
    compute_step(3, do_m1=False)
    for m in m2_atoms(1, 1):
        e(m)
    e(".Ldrained_%=:")
    stamp(3)
    e(f"s_bitcmp1_b32 s{S_FLAGS}, 2")                # the first half waits one barrier for the second
    e("s_cbranch_scc1 .Lno_tail_%=")
    bar()
    e(".Lno_tail_%=:")
    e("s_nop 15")
    e("s_nop 15")
    bar()                                            # every wave of the workgroup is done with its rings
    stamp(4)
    for i in range(16):
        e(f"ds_write_b128 %{P['dump']}, {ar(i * 4)} offset:{i * 1024}")
    e("s_waitcnt lgkmcnt(0)")
    if STAMPS:
        stamp(5)
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_mov_b64 s[{S_SAVE}:{S_SAVE + 1}], exec")
        e("s_mov_b64 exec, 1")
        e(f"s_getreg_b32 s{S_TMP}, hwreg(HW_REG_HW_ID)")
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
            e(f"global_store_dwordx4 v{V_ROW + 15}, {vr(V_ROW + 4 * i)}, %24 offset:{16 * i}")
        e("s_waitcnt vmcnt(0)")
        e(f"s_mov_b64 exec, s[{S_SAVE}:{S_SAVE + 1}]")
    e("s_branch .Lend_%=")
    # ---- a wave without keys in a half that has work: feeds the ring, keeps every barrier
    e(".Lfeed_only_%=:")
    e(f"s_bitcmp1_b32 s{S_FLAGS}, 2")
    e("s_cbranch_scc0 .Lfeed_no_shift_%=")
    bar()
    e(".Lfeed_no_shift_%=:")
    e(f"s_add_u32 s{S_TMP}, s{S_T}, 1")
    e(f"s_cmp_lt_u32 s{S_TMP}, s{S_NT}")
    e("s_cbranch_scc0 .Lfeed_last_%=")
    e(".Lfeed_loop_%=:")
    variants(feed_tile, "f")
    advance_tile()
    e(f"s_add_u32 s{S_TMP}, s{S_T}, 1")
    e(f"s_cmp_lt_u32 s{S_TMP}, s{S_NT}")
    e("s_cbranch_scc1 .Lfeed_loop_%=")
    e(".Lfeed_last_%=:")
    for i in range(4):
        bar()
    e("s_branch .Ltail_bars_%=")
    # ---- a half without a work item (odd number of items): keeps every barrier
    e(".Lidle_half_%=:")
    bar()                                            # barrier 0
    e(f"s_bitcmp1_b32 s{S_FLAGS}, 2")
    e("s_cbranch_scc0 .Lidle_no_shift_%=")
    bar()
    e(".Lidle_no_shift_%=:")
    e(f"s_mov_b32 s{S_T}, 0")
    e(".Lidle_loop_%=:")
    for i in range(4):
        bar()
    e(f"s_add_u32 s{S_T}, s{S_T}, 1")
    e(f"s_cmp_lt_u32 s{S_T}, s{S_NT}")
    e("s_cbranch_scc1 .Lidle_loop_%=")
    e(".Ltail_bars_%=:")
    e(f"s_bitcmp1_b32 s{S_FLAGS}, 2")
    e("s_cbranch_scc1 .Ltail_no_%=")
    bar()
    e(".Ltail_no_%=:")
    bar()
    e(".Lend_%=:")
    e(f"s_mov_b32 m0, s{S_M0}")


emit()
NAME = "NVIT_ATTN_DKV_PP_STAMPS" if STAMPS else "NVIT_ATTN_DKV_PP_ASM"
print("// GENERATED by gen/gen_attn_dkv_pp_asm.py - do not edit (regenerate: make -C nvit_amd/csrc gen)")
print(f"#define {NAME}_BODY \\")
for line in out:
    print(f'  "{line}\\n\\t" \\')
print('  ""')
clob = [f'"v{i}"' for i in list(range(16, 25)) + list(range(V_ROW, V_END))] + [f'"a{i}"' for i in range(A_END)] + [f'"s{i}"' for i in range(40, S_STAMP_END if STAMPS else 72)] + ['"vcc"', '"memory"']
print(f"#define {NAME}_CLOBBERS " + ", ".join(clob))
print(f"// instructions: {sum(1 for l in out if not l.startswith(';') and not l.endswith(':'))}")
