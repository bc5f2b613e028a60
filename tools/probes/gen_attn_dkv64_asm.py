#!/usr/bin/env python3
"""Generator of the hand-placed main loop of the attention backward dK/dV kernel (gfx950, one wave per SIMD).

Writes attn_dkv_asm.inc: ONE inline-asm string (the whole tile loop of nvit_amd/csrc/attn_bwd_asm.hip) plus its clobber
list.  Reference semantics: the dK/dV half of the backward of F.scaled_dot_product_attention as the reference calls it
(/root/reference/nvit/model.py:121-124); the arithmetic, operand layouts and accumulation order are those of the
compiler-built kernel attn_bwd_dkv_mfma_kernel (attn_mfma.hip), against which this one is bit-exact.

Structure: a workgroup = 2 waves, each wave owns 64 keys (4 key fragments of 16) and the whole
512-entry register file of its SIMD: dK/dV accumulators (128) and the K/V fragments (64) in the accumulation half,
transposed Q/dO fragments loaded by ds_read_b64_tr_b16 straight into it as well.  The work of a 64-query tile is cut
into 8 groups g = (32-query half, key fragment); a software pipeline runs M1(g+1) [S and dP products, 8 MFMA], V(g)
[exp2, p*(dP-delta), bf16 packs: 24 VALU] and M2(g-1) [dV and dK products, 8 MFMA] in every step, so each dependency
crosses a step boundary and every instruction of a step can sit in any MFMA gap; LDS fragment reads run one 32-query half
ahead into a second register set; the Q/dO tiles arrive by LDS-DMA two tiles ahead (one barrier per tile).

PROBE RECORD (round 4), not part of libnvit_hip.so: the wrapper kernel (attn_bwd_dkv_asm_kernel: 128 threads, 64 KiB + 1 KiB of
LDS, HIP prologue / epilogue around this statement) is in the tree at commit 13407bb.  Measured there (MI355X, B*H = 1536,
T = 784): bit-exact with the compiler-built kernel on every shape tried; tile loop 2 700 cycles per 64x64 tile against 4 660
for the compiler-built kernel's two 64x32 wave tiles (1.57x), but 6.8 us of prologue and 7.4 us of epilogue per workgroup
sit unhidden on a SIMD that hosts one wave (31 us per workgroup in all against 29.5), so the kernel is 7-13 % SLOWER than
the compiler-built one.  The two-waves-per-SIMD form (nvit_amd/csrc/gen/gen_attn_dkv32_asm.py) is the product.

usage: python3 gen_attn_dkv64_asm.py > attn_dkv_asm.inc
"""
import os
import sys

# timing probes (results are garbage by design; never the committed .inc): GEN_PROBE = comma list of
#   nodma (no LDS-DMA inside the tile loop), novalu (no exp / mul / pack), nolds (no fragment reads in the loop),
#   nom1 / nom2 (no S,dP / no dV,dK products), dmablock (all DMA of a tile in one block: the first version)
PROBE = set(filter(None, os.environ.get("GEN_PROBE", "").split(",")))

SLOT = 2 * 8192 + 512      # Q tile | dO tile | -lse[64] | -delta[64]   (= DKV_SLOT of attn_mfma.hip)
NSLOT = 3

# ---- operand numbers of the asm statement (attn_bwd_asm.hip passes them in this order)
OP = dict(qbase=0, gbase=1, lbase=2, dbase=3, kbase=4, vbase=5, nt=6, ldg=7, ring=8, nvalid_last=9, active=10, wofs=11,
          voff_q0=12, voff_g0=13, rows_last=14, chunk16=15, lane4=16, kvoff0=17, lds_pack0=21, dump=25)

# ---- fixed registers
S_Q, S_G, S_L, S_D = 40, 42, 44, 46          # DMA cursors (64-bit)
S_NT, S_LDG, S_RING, S_NVL = 48, 49, 50, 51
S_T, S_TD, S_SLOTC, S_SLOTD = 52, 53, 54, 55  # compute tile, dma tile, LDS base of the tile being read / being filled
S_TMP, S_TMP2 = 56, 57
S_EXEC = 58                                   # 58:59 cmp mask, 64:65 saved exec
S_P16, S_T64, S_WOFS, S_RINGEND = 60, 61, 62, 63
S_SAVE = 64
S_FLAGS = 66
S_DW = 67                                     # slot_d + wofs
S_M0 = 68                                     # m0 of the surrounding code (the compiler reserves it)

V_A0, V_A1, V_N, V_T = 16, 17, 18, 19         # slot-relative LDS offsets (V_T..V_T+3)
V_RA0, V_RA1, V_RN, V_RT = 23, 24, 25, 26     # same, absolute for the tile being read (V_RT..V_RT+3)
V_VQ, V_VG, V_VQL, V_VGL = 30, 34, 38, 42     # DMA per-lane offsets, 4 pieces each (full tile / clamped last tile)
V_LSE, V_LSEL = 46, 47
V_ROW = 48                                    # two fragment sets of 48 registers
V_Z = 144                                     # two sets of z[2] (8) + w[2] (8)
V_P = 176                                     # two sets of pb (4) + sb (4)
V_LANE, V_NINF, V_TMP = 208, 209, 210
A_DK, A_DV, A_KF, A_VF, A_TR = 0, 64, 128, 160, 192

out = []


def e(s):
    out.append(s)


def vr(b, n=4):
    return f"v[{b}:{b + n - 1}]" if n > 1 else f"v{b}"


def ar(b, n=4):
    return f"a[{b}:{b + n - 1}]" if n > 1 else f"a{b}"


def row_base(buf):
    return V_ROW + 48 * buf


def mfma(d, a, b, c):
    return f"v_mfma_f32_16x16x32_bf16 {d}, {a}, {b}, {c}"


# ---------------------------------------------------------------- atoms of one pipeline step
def m1_atoms(fn, half_buf, zbuf):
    """S and dP products of (32-query half in fragment set half_buf, key fragment fn) into z/w set zbuf: two lists (ks0, ks1)."""
    R, Z = row_base(half_buf), V_Z + 16 * zbuf
    ks0, ks1 = [], []
    for qq in (0, 1):
        z, w = vr(Z + qq * 4), vr(Z + 8 + qq * 4)
        ks0.append(mfma(z, vr(R + (qq * 2 + 0) * 4), ar(A_KF + (fn * 2 + 0) * 4), vr(R + 32 + qq * 4)))
        ks0.append(mfma(w, vr(R + 16 + (qq * 2 + 0) * 4), ar(A_VF + (fn * 2 + 0) * 4), vr(R + 40 + qq * 4)))
        ks1.append(mfma(z, vr(R + (qq * 2 + 1) * 4), ar(A_KF + (fn * 2 + 1) * 4), z))
        ks1.append(mfma(w, vr(R + 16 + (qq * 2 + 1) * 4), ar(A_VF + (fn * 2 + 1) * 4), w))
    return ks0, ks1


def m2_atoms(fp, tr_buf, pbuf):
    """dV and dK products of key fragment fp with the packed P / dS set pbuf and the transposed fragment set tr_buf."""
    T, P = A_TR + 32 * tr_buf, V_P + 8 * pbuf
    res = []
    for df in range(4):
        dv = ar(A_DV + (df * 4 + fp) * 4)
        dk = ar(A_DK + (df * 4 + fp) * 4)
        res.append(mfma(dv, ar(T + df * 4), vr(P), dv))            # dV^T[d][key] += dO^T P
        res.append(mfma(dk, ar(T + 16 + df * 4), vr(P + 4), dk))   # dK^T[d][key] += Q^T dS
    return res


def v_atoms(zbuf, pbuf):
    """p = exp2(S - lse), dS = p * (dP - delta), both packed to bf16 (z/w set zbuf -> pb/sb set pbuf)."""
    Z, P = V_Z + 16 * zbuf, V_P + 8 * pbuf
    res = []
    for qq in (0, 1):
        z, w = Z + qq * 4, Z + 8 + qq * 4
        for r in range(4):
            res.append(f"v_exp_f32_e32 v{z + r}, v{z + r}")
        for r in range(4):
            res.append(f"v_mul_f32_e32 v{w + r}, v{z + r}, v{w + r}")
        res.append(f"v_cvt_pk_bf16_f32 v{P + qq * 2}, v{z}, v{z + 1}")
        res.append(f"v_cvt_pk_bf16_f32 v{P + qq * 2 + 1}, v{z + 2}, v{z + 3}")
        res.append(f"v_cvt_pk_bf16_f32 v{P + 4 + qq * 2}, v{w}, v{w + 1}")
        res.append(f"v_cvt_pk_bf16_f32 v{P + 4 + qq * 2 + 1}, v{w + 2}, v{w + 3}")
    return res


def row_reads(s2, buf):
    """12 ds_read_b128: Q / dO row fragments and the -lse / -delta quads of the 32-query half s2 of the tile being read."""
    R = row_base(buf)
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


def tr_reads(s2, buf):
    """16 ds_read_b64_tr_b16: dO^T and Q^T fragments of the half s2 (tr_frag of attn_mfma.hip: lo rows, hi = +16 rows)."""
    T = A_TR + 32 * buf
    res = []
    for df in range(4):
        res.append(f"ds_read_b64_tr_b16 {ar(T + df * 4, 2)}, v{V_RT + df} offset:{8192 + s2 * 4096}")
        res.append(f"ds_read_b64_tr_b16 {ar(T + df * 4 + 2, 2)}, v{V_RT + df} offset:{8192 + s2 * 4096 + 2048}")
        res.append(f"ds_read_b64_tr_b16 {ar(T + 16 + df * 4, 2)}, v{V_RT + df} offset:{s2 * 4096}")
        res.append(f"ds_read_b64_tr_b16 {ar(T + 16 + df * 4 + 2, 2)}, v{V_RT + df} offset:{s2 * 4096 + 2048}")
    return res


def dma_atoms(last):
    """LDS-DMA of the tile S_TD into the slot S_SLOTD: 4 + 4 pieces of 1 KiB (this wave's half of Q and dO) + the two
    256-byte rows of -lse and -delta.  Each atom is a list of instructions that stay together."""
    vq, vg, vl = (V_VQL, V_VGL, V_LSEL) if last else (V_VQ, V_VG, V_LSE)
    atoms = [[f"s_add_u32 s{S_DW}, s{S_SLOTD}, s{S_WOFS}"]]
    for i in range(4):
        atoms.append([f"s_add_u32 s{S_TMP}, s{S_DW}, {i * 2048}", f"s_mov_b32 m0, s{S_TMP}", "s_nop 0",
                      f"global_load_lds_dwordx4 v{vq + i}, s[{S_Q}:{S_Q + 1}]"])
    for i in range(4):
        atoms.append([f"s_add_u32 s{S_TMP}, s{S_DW}, {8192 + i * 2048}", f"s_mov_b32 m0, s{S_TMP}", "s_nop 0",
                      f"global_load_lds_dwordx4 v{vg + i}, s[{S_G}:{S_G + 1}]"])
    atoms.append([f"s_add_u32 s{S_TMP}, s{S_SLOTD}, 16384", f"s_mov_b32 m0, s{S_TMP}", "s_nop 0",
                  f"global_load_lds_dword v{vl}, s[{S_L}:{S_L + 1}]"])
    atoms.append([f"s_add_u32 s{S_TMP}, s{S_SLOTD}, 16640", f"s_mov_b32 m0, s{S_TMP}", "s_nop 0",
                  f"global_load_lds_dword v{vl}, s[{S_D}:{S_D + 1}]"])
    atoms.append([f"s_add_u32 s{S_Q}, s{S_Q}, 8192", f"s_addc_u32 s{S_Q + 1}, s{S_Q + 1}, 0",
                  f"s_add_u32 s{S_G}, s{S_G}, s{S_T64}", f"s_addc_u32 s{S_G + 1}, s{S_G + 1}, 0"])
    atoms.append([f"s_add_u32 s{S_L}, s{S_L}, 256", f"s_addc_u32 s{S_L + 1}, s{S_L + 1}, 0",
                  f"s_add_u32 s{S_D}, s{S_D}, 256", f"s_addc_u32 s{S_D + 1}, s{S_D + 1}, 0"])
    atoms.append([f"s_add_u32 s{S_TD}, s{S_TD}, 1", f"s_add_u32 s{S_SLOTD}, s{S_SLOTD}, {SLOT}",
                  f"s_cmp_ge_u32 s{S_SLOTD}, s{S_RINGEND}", f"s_cselect_b32 s{S_SLOTD}, s{S_RING}, s{S_SLOTD}"])
    return atoms


def emit_dma(tag):
    """DMA of tile S_TD as one block: the clamped variant when it is the ragged last tile."""
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


def emit_fixup(tag, tile_expr_plus):
    """If the tile whose data has just landed (index S_T + tile_expr_plus) is the ragged last one: -lse of the query rows
    past the end becomes -inf, so that their probabilities are exactly 0 (the rows themselves are clamped copies)."""
    e(f"s_add_u32 s{S_TMP2}, s{S_T}, {tile_expr_plus + 1}")
    e(f"s_cmp_eq_u32 s{S_TMP2}, s{S_NT}")
    e(f"s_cbranch_scc0 .Lfix_skip_{tag}_%=")
    e(f"s_cmp_lt_u32 s{S_NVL}, 64")
    e(f"s_cbranch_scc0 .Lfix_skip_{tag}_%=")
    e(f"v_cmp_ge_u32_e64 s[{S_EXEC}:{S_EXEC + 1}], v{V_LANE}, s{S_NVL}")
    e(f"s_and_saveexec_b64 s[{S_SAVE}:{S_SAVE + 1}], s[{S_EXEC}:{S_EXEC + 1}]")
    e(f"v_add_u32_e32 v{V_TMP}, s{S_TMP}, v{V_LSE}")          # S_TMP = LDS base of that tile's slot (set by the caller)
    e(f"ds_write_b32 v{V_TMP}, v{V_NINF} offset:16384")
    e(f"s_mov_b64 exec, s[{S_SAVE}:{S_SAVE + 1}]")
    e("s_waitcnt lgkmcnt(0)")
    e(f".Lfix_skip_{tag}_%=:")


def next_slot(dst, src):
    e(f"s_add_u32 s{dst}, s{src}, {SLOT}")
    e(f"s_cmp_ge_u32 s{dst}, s{S_RINGEND}")
    e(f"s_cselect_b32 s{dst}, s{S_RING}, s{dst}")


def set_read_addresses():
    e(f"v_add_u32_e32 v{V_RA0}, s{S_SLOTC}, v{V_A0}")
    e(f"v_add_u32_e32 v{V_RA1}, s{S_SLOTC}, v{V_A1}")
    e(f"v_add_u32_e32 v{V_RN}, s{S_SLOTC}, v{V_N}")
    for i in range(4):
        e(f"v_add_u32_e32 v{V_RT + i}, s{S_SLOTC}, v{V_T + i}")


def interleave(mf, va, lds, dma=()):
    """One pipeline step: 16 (or fewer) MFMAs with the VALU / LDS instructions and the LDS-DMA atoms of the step spread
    over their gaps (a DMA atom = m0 set-up + the load, kept together)."""
    nm = max(len(mf), 1)
    vi = li = di = 0
    dma = list(dma)
    for i, m in enumerate(mf):
        e(m)
        want = (len(va) * (i + 1) + nm - 1) // nm
        while vi < want:
            e(va[vi])
            vi += 1
        wl = (len(lds) * (i + 1) + nm - 1) // nm
        while li < wl:
            e(lds[li])
            li += 1
        wd = (len(dma) * (i + 1)) // nm
        while di < wd:
            for ins in dma[di]:
                e(ins)
            di += 1
    while vi < len(va):
        e(va[vi])
        vi += 1
    while li < len(lds):
        e(lds[li])
        li += 1
    while di < len(dma):
        for ins in dma[di]:
            e(ins)
        di += 1


def step(j, do_m1=True, do_m2=True, reads=True, dma=()):
    """Step j (0..7) of a tile: group g = 8t + j, half = j // 4, key fragment f = j % 4."""
    f, half = j % 4, j // 4
    gn = j + 1                       # next group (may be 8 = group 0 of the next tile)
    m1 = ([], [])
    if do_m1 and "nom1" not in PROBE:
        m1 = m1_atoms(gn % 4, (gn // 4) % 2, gn % 2)
    m2 = m2_atoms((j - 1) % 4, ((j - 1) // 4) % 2, (j - 1) % 2) if (do_m2 and "nom2" not in PROBE) else []
    # MFMA order: M1 ks0 (4), M2 (2), M1 ks1 (4), M2 (6): a chain's second product is 6 MFMAs behind its first, and the
    # last S / dP result is 6 MFMAs + ~9 VALU ahead of the next step's first exp
    mf = m1[0] + m2[:2] + m1[1] + m2[2:]
    va = [] if "novalu" in PROBE else v_atoms(j % 2, j % 2)
    lds = []
    if reads and "nolds" not in PROBE:
        nh = half + 1                # fragments of the NEXT half: (nh % 2) selects the half inside the tile being read
        if f in (0, 1):
            rr = row_reads(nh % 2, nh % 2)
            lds = rr[:6] if f == 0 else rr[6:]
        else:
            tr = tr_reads(nh % 2, nh % 2)
            lds = tr[:8] if f == 2 else tr[8:]
    interleave(mf, va, lds, dma)


def second_half(variant):
    """Steps 4..7 of a steady tile.  variant: 'N' no tile left to fetch, 'F' fetch a full tile, 'L' fetch the ragged last
    tile (clamped rows).  The 15 DMA atoms of a tile are dealt over the four steps, each into an MFMA gap."""
    atoms = [] if variant == "N" else dma_atoms(variant == "L")
    per = {4: atoms[0:4], 5: atoms[4:7], 6: atoms[7:11], 7: atoms[11:]} if atoms else {4: [], 5: [], 6: [], 7: []}
    for j in range(4, 8):
        e(f"; step {j} ({variant})")
        if j == 5:
            e("s_waitcnt lgkmcnt(6)")    # transposed fragments of the second half
        if j == 7:
            e("s_waitcnt lgkmcnt(8)")    # row fragments of the next tile's first half
        step(j, dma=per[j])


def emit():
    P = OP
    e("; ---------------- prologue: fixed registers from the operands")
    e(f"s_mov_b32 s{S_M0}, m0")
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
    e(f"s_lshl_b32 s{S_P16}, s{S_LDG}, 4")
    e(f"s_lshl_b32 s{S_T64}, s{S_LDG}, 6")
    e(f"s_add_u32 s{S_RINGEND}, s{S_RING}, {NSLOT * SLOT}")
    e(f"s_mov_b32 s{S_T}, 0")
    e(f"s_mov_b32 s{S_TD}, 0")
    e(f"s_mov_b32 s{S_SLOTC}, s{S_RING}")
    e(f"s_mov_b32 s{S_SLOTD}, s{S_RING}")
    # per-lane LDS offsets (packed two per operand)
    e(f"v_and_b32_e32 v{V_A0}, 0xffff, %{P['lds_pack0']}")
    e(f"v_lshrrev_b32_e32 v{V_A1}, 16, %{P['lds_pack0']}")
    e(f"v_and_b32_e32 v{V_N}, 0xffff, %{P['lds_pack0'] + 1}")
    e(f"v_lshrrev_b32_e32 v{V_T}, 16, %{P['lds_pack0'] + 1}")
    e(f"v_and_b32_e32 v{V_T + 1}, 0xffff, %{P['lds_pack0'] + 2}")
    e(f"v_lshrrev_b32_e32 v{V_T + 2}, 16, %{P['lds_pack0'] + 2}")
    e(f"v_mov_b32_e32 v{V_T + 3}, %{P['lds_pack0'] + 3}")
    # DMA offsets: piece i of this wave = rows 16 i + 8 wid + r8
    e(f"v_mov_b32_e32 v{V_VQ}, %{P['voff_q0']}")
    e(f"v_mov_b32_e32 v{V_VG}, %{P['voff_g0']}")
    for i in range(1, 4):
        e(f"v_add_u32_e32 v{V_VQ + i}, {2048 * i}, v{V_VQ}")
        e(f"v_add_u32_e32 v{V_VG + i}, s{S_P16}, v{V_VG + i - 1}")
    for i in range(4):   # clamped rows of the ragged last tile
        e(f"v_bfe_u32 v{V_TMP}, %{P['rows_last']}, {8 * i}, 8")
        e(f"v_lshlrev_b32_e32 v{V_VQL + i}, 7, v{V_TMP}")
        e(f"v_add_u32_e32 v{V_VQL + i}, v{V_VQL + i}, %{P['chunk16']}")
        e(f"v_mul_lo_u32 v{V_VGL + i}, v{V_TMP}, s{S_LDG}")
        e(f"v_add_u32_e32 v{V_VGL + i}, v{V_VGL + i}, %{P['chunk16']}")
    e(f"v_mov_b32_e32 v{V_LSE}, %{P['lane4']}")
    e(f"s_sub_u32 s{S_TMP}, s{S_NVL}, 1")
    e(f"s_lshl_b32 s{S_TMP}, s{S_TMP}, 2")
    e(f"v_min_u32_e32 v{V_LSEL}, s{S_TMP}, v{V_LSE}")
    e(f"v_lshrrev_b32_e32 v{V_LANE}, 2, v{V_LSE}")
    e(f"v_mov_b32_e32 v{V_NINF}, 0xff800000")
    e(f"v_mov_b32_e32 v{V_TMP + 1}, %{P['dump']}")
    # zero: accumulators, both transposed fragment sets, both packed sets (group -1 multiplies zeros)
    for i in range(128):
        e(f"v_accvgpr_write_b32 a{A_DK + i}, 0")
    for i in range(64):
        e(f"v_accvgpr_write_b32 a{A_TR + i}, 0")
    for i in range(16):
        e(f"v_mov_b32_e32 v{V_P + i}, 0")
    # tile 0, the K / V fragments of this wave's 64 keys (-> accumulation registers: MFMA B operands for the whole kernel),
    # tile 1; only tile 1 may still be in flight when the first fragments are read
    emit_dma("p0")
    for f in range(4):
        for ks in range(2):
            e(f"global_load_dwordx4 {ar(A_KF + (f * 2 + ks) * 4)}, %{P['kvoff0'] + f}, %{P['kbase']} offset:{ks * 64}")
            e(f"global_load_dwordx4 {ar(A_VF + (f * 2 + ks) * 4)}, %{P['kvoff0'] + f}, %{P['vbase']} offset:{ks * 64}")
    e(f"s_cmp_lt_u32 s{S_TD}, s{S_NT}")
    e("s_cbranch_scc0 .Lno_second_%=")
    emit_dma("p1")
    e("s_waitcnt vmcnt(10)")
    e("s_branch .Lpro_waited_%=")
    e(".Lno_second_%=:")
    e("s_waitcnt vmcnt(0)")
    e(".Lpro_waited_%=:")
    e(f"s_mov_b32 s{S_TMP}, s{S_RING}")
    emit_fixup("pro", 0)
    e("s_barrier")
    e(f"s_cmp_eq_u32 s{S_FLAGS}, 0")
    e("s_cbranch_scc1 .Lfeed_only_%=")
    set_read_addresses()
    for r in row_reads(0, 0):
        e(r)
    e("s_waitcnt lgkmcnt(0)")
    m1 = m1_atoms(0, 0, 0)
    tr = tr_reads(0, 0)
    interleave(m1[0] + m1[1], [], tr)
    e("; ---------------- steady tiles: t = 0 .. nt-2")
    e(f"s_add_u32 s{S_TMP}, s{S_T}, 1")
    e(f"s_cmp_lt_u32 s{S_TMP}, s{S_NT}")
    e("s_cbranch_scc0 .Llast_tile_%=")
    e(".Ltile_loop_%=:")
    for j in range(4):
        e(f"; step {j}")
        if j == 1:
            e("s_waitcnt lgkmcnt(6)")    # transposed fragments of this half (issued in steps 6, 7 / the prologue)
        if j == 3:
            e("s_waitcnt lgkmcnt(8)")    # row fragments of the second half
        step(j)
    # tile t+1 has landed (its DMA is the only vector-memory work in flight); everybody is done with tile t-1
    e("s_waitcnt vmcnt(0)")
    next_slot(S_TMP, S_SLOTC)
    emit_fixup("loop", 1)
    e("s_barrier")
    next_slot(S_SLOTC, S_SLOTC)
    set_read_addresses()
    if "nodma" in PROBE:
        second_half("N")
    elif "dmablock" in PROBE:
        e(f"s_cmp_lt_u32 s{S_TD}, s{S_NT}")
        e("s_cbranch_scc0 .Lno_dma_%=")
        emit_dma("loop")
        e(".Lno_dma_%=:")
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
    e(f"s_add_u32 s{S_T}, s{S_T}, 1")
    e(f"s_add_u32 s{S_TMP}, s{S_T}, 1")
    e(f"s_cmp_lt_u32 s{S_TMP}, s{S_NT}")
    e("s_cbranch_scc1 .Ltile_loop_%=")
    e("; ---------------- last tile: nothing left to fetch or to read ahead past its second half")
    e(".Llast_tile_%=:")
    # a last tile with at most 32 valid queries: its second half contributes exactly nothing (p = 0), skip it
    e(f"s_cmp_le_u32 s{S_NVL}, 32")
    e("s_cbranch_scc0 .Llast_full_%=")
    for j in range(4):
        e(f"; short last-tile step {j}")
        if j == 1:
            e("s_waitcnt lgkmcnt(0)")
        step(j, do_m1=(j < 3), reads=False)
    for m in m2_atoms(3, 0, 1):
        e(m)
    e("s_branch .Ldrained_%=")
    e(".Llast_full_%=:")
    for j in range(8):
        e(f"; last-tile step {j}")
        if j == 1:
            e("s_waitcnt lgkmcnt(6)")
        if j == 3:
            e("s_waitcnt lgkmcnt(8)")
        if j == 5:
            e("s_waitcnt lgkmcnt(0)")
        step(j, do_m1=(j < 7), reads=(j < 4))
    e("; ---------------- drain: dV / dK products of the last group")
    for m in m2_atoms(3, 1, 1):
        e(m)
    e(".Ldrained_%=:")
    e("s_nop 15")
    e("s_nop 15")
    e(".Ldump_%=:")
    e("s_barrier")                       # the ring is idle in every wave: it becomes the accumulator hand-over area
    for i in range(32):
        e(f"ds_write_b128 v{V_TMP + 1}, {ar(i * 4)} offset:{i * 1024}")
    e("s_waitcnt lgkmcnt(0)")
    e("s_branch .Lend_%=")
    e("; ---------------- a wave without keys: feeds the ring and the barriers only")
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
print("// GENERATED by gen/gen_attn_dkv_asm.py - do not edit (regenerate: make -C nvit_amd/csrc gen)")
print("#define NVIT_ATTN_DKV_ASM_BODY \\")
for line in out:
    print(f'  "{line}\\n\\t" \\')
print('  ""')
clob = [f'"v{i}"' for i in range(16, 216)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(40, 72)] + ['"vcc"', '"memory"']
print("#define NVIT_ATTN_DKV_ASM_CLOBBERS " + ", ".join(clob))
print(f"// instructions: {sum(1 for l in out if not l.startswith(';') and not l.endswith(':'))}")
