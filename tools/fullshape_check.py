"""Numeric check of every GEMM shape / epilogue and of attention AT THE BENCHMARKED SIZE
(nViT-Base, B=128: M = B*T = 100 352 rows, B*H = 1536 heads of T = 784 tokens).

The oracle cannot run this size in reasonable time, so the reference here is fp32/fp64 torch math ON THE GPU of
the same bf16 operands (matmul + the epilogue formula), on rows sampled from EVERY 256-row tile (plus the whole
first and last tile), all columns.  That is the correctness evidence for the persistent tile walk (1 176 - 9 408
tiles on 256 workgroups), the XCD grouping, the LDS-DMA ring across tile boundaries and the row splits of the
weight-gradient kernel, none of which the small-shape op tests reach.

Used by tests/test_gpu_fullshape.py (-m gpu) and by `bench.py --check` (outside the timed region).
All device work goes through the C ABI (nvit_amd.ops); torch supplies the reference arithmetic only.
"""
from __future__ import annotations

import math
import os
import sys
from typing import Dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch


def _rnd(shape, seed, dev, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * scale).to(dtype)


def _rows(M: int, dev) -> torch.Tensor:
    """rows hitting every 256-row tile (stride 49 is coprime with 256) + the whole first and last tile."""
    r = torch.cat((torch.arange(0, M, 49), torch.arange(0, min(256, M)), torch.arange(max(0, M - 256), M)))
    return torch.unique(r).to(dev)


def _interleave(x: torch.Tensor, F: int) -> torch.Tensor:
    u, v = x[..., :F], x[..., F:]
    sh = x.shape[:-1]
    return torch.stack([u.reshape(*sh, F // 16, 16), v.reshape(*sh, F // 16, 16)], dim=-2).reshape(*sh, 2 * F)


def _perm_rows(N: int) -> torch.Tensor:
    """slab row s (interleaved order) -> natural row, as gemm_tn(perm=1) un-permutes."""
    s = torch.arange(N)
    q, w = s // 32, s % 32
    return torch.where(w < 16, q * 16 + w, N // 2 + q * 16 + (w - 16))


class Report:
    def __init__(self, verbose: bool):
        self.rows: Dict[str, dict] = {}
        self.verbose = verbose

    def add(self, name: str, err: float, tol: float):
        ok = bool(err <= tol) and math.isfinite(err)
        self.rows[name] = {"err": float(f"{err:.3e}"), "tol": float(f"{tol:.3e}"), "ok": ok}
        if self.verbose:
            print(f"[fullshape] {name:58s} err {err:.3e}  tol {tol:.3e}  {'ok' if ok else 'FAIL'}", flush=True)

    @property
    def ok(self) -> bool:
        return all(r["ok"] for r in self.rows.values())


def check_all(B: int = 128, T: int = 784, C: int = 768, H: int = 12, verbose: bool = True, dev=None) -> Report:
    from nvit_amd import ops
    from nvit_amd._lib import BF16
    dev = dev or torch.device("cuda", torch.cuda.current_device())
    rep = Report(verbose)
    M = B * T
    d = C // H
    rows = _rows(M, dev)
    bf = torch.bfloat16

    # ---------------------------------------------------------------- NT GEMMs, plain epilogues (EPI 1 / 2)
    # (N, K, out dtype, accumulate, extras) as launched by one Base train step
    nt_cases = [
        (C, C, torch.float32, False, ""),            # att_c_proj / out_proj forward
        (C, 4 * C, torch.float32, False, ""),        # mlp_c_proj forward
        (C, 8 * C, torch.float32, True, ""),         # c_fc data gradient, accumulated into dh1
        (C, 3 * C, torch.float32, True, ""),         # qkv data gradient, accumulated into dx
        (C, 2 * C, torch.float32, False, ""),        # cross k/v data gradient
        (C, C, bf, False, ""),                       # att_c_proj data gradient (bf16 out)
        (C, 2 * C, bf, False, ""),                   # cross proj data gradient (bf16 out)
        (192, C, torch.float32, False, "bias"),      # reconstruction head (256x128 tiles)
    ]
    for N, K, odt, acc, extra in nt_cases:
        A = _rnd((M, K), 11, dev)
        W = _rnd((N, K), 12, dev, scale=1.0 / math.sqrt(K))
        bias = _rnd((N,), 13, dev, dtype=torch.float32) if "bias" in extra else None
        pos = _rnd((T, N), 14, dev, dtype=torch.float32) if "pos" in extra else None
        base = _rnd((M, N), 15, dev, dtype=odt) if acc else None
        out = base.clone() if acc else None
        out = ops.gemm_nt(A, W, M, N, K, out=out, out_dtype=odt, bias=bias, rowadd=pos,
                          rowadd_period=T if pos is not None else 0, accumulate=acc)
        ref = A[rows].float() @ W.float().t()
        if bias is not None:
            ref = ref + bias
        if pos is not None:
            ref = ref + pos[rows % T]
        if acc:
            ref = ref + base[rows].float()
        err = (out[rows].float() - ref).abs().max().item()
        mag = ref.abs().max().item()
        tol = (2e-5 if odt == torch.float32 else 2.0 ** -8) * max(1.0, mag)
        rep.add(f"gemm_nt M={M} N={N} K={K} out={'f32' if odt == torch.float32 else 'bf16'}"
                f"{' +=' if acc else ''} {extra}", err, tol)
        del A, W, out, ref, base

    # ---------------------------------------------------------------- fused dual patch embedding (gather + split-operand MFMA)
    # reference: fp32 unfold of the (reflect-padded) image on the GPU, fp64 product of the un-rounded operands, rows
    # sampled from every 256-token tile.  Only for the square geometry T = G*G tokens of 8x8 / 16x16 patches.
    G = math.isqrt(T)
    if G * G == T:
        import torch.nn.functional as Fn
        ch, Pl, Pg = 3, 8, 16
        S = G * Pl
        pad = (Pg - Pl) // 2
        img = _rnd((B, ch, S, S), 16, dev, dtype=torch.float32)
        sh, ws = [], []
        for K, seed in ((ch * Pl * Pl, 17), (ch * Pg * Pg, 18)):
            w = _rnd((C, K), seed, dev, scale=1.0 / math.sqrt(K), dtype=torch.float32)
            Kp = ops.patch_kp(K)
            img_w = torch.zeros((C, 2 * Kp), device=dev, dtype=bf)
            t_, n_ = ops.shadow_table([(w, img_w, 2 * Kp, K, None, 0, 0, 2)], dev)
            ops.shadow_weights(t_, n_, BF16)
            sh.append(img_w)
            ws.append(w)
        b_l, b_g = (_rnd((C,), s_, dev, scale=0.1, dtype=torch.float32) for s_ in (19, 20))
        p_l, p_g = (_rnd((T, C), s_, dev, scale=0.02, dtype=torch.float32) for s_ in (24, 25))
        loc, glo, a_l, a_g, lo_l, lo_g = ops.patch_embed_fwd(img, sh[0], b_l, p_l, sh[1], b_g, p_g, Pl, Pg, C, twins=True)
        A_l = Fn.unfold(img, Pl, stride=Pl).transpose(1, 2).reshape(M, -1)[rows]
        A_g = Fn.unfold(Fn.pad(img, (pad,) * 4, mode="reflect"), Pg, stride=Pl).transpose(1, 2).reshape(M, -1)[rows]
        for name, out, A, w, b, pos, a_hi in (("local", loc, A_l, ws[0], b_l, p_l, a_l), ("global", glo, A_g, ws[1], b_g, p_g, a_g)):
            ref = A.double() @ w.double().t() + b.double() + pos[rows % T].double()
            err = (out[rows].double() - ref).abs().max().item()
            rep.add(f"patch_embed {name} M={M} K={A.shape[1]} (fused gather, hi/lo split)", err, 1e-5 * max(1.0, ref.abs().max().item()))
            e_rows = (a_hi[rows][:, :A.shape[1]].float() - A.to(bf).float()).abs().max().item()
            rep.add(f"patch_embed {name} saved bf16 patch rows (exact)", e_rows, 1e-30)
        e_tw = max((lo_l[rows].float() - loc[rows].to(bf).float()).abs().max().item(),
                   (lo_g[rows].float() - glo[rows].to(bf).float()).abs().max().item())
        rep.add("patch_embed bf16 twins of the outputs (exact)", e_tw, 1e-30)
        del img, loc, glo, a_l, a_g, lo_l, lo_g, A_l, A_g, sh, ws

    # ---------------------------------------------------------------- EPI 3: SwiGLU forward (c_fc, cross proj)
    for F, use_gs in ((4 * C, True), (C, False)):
        K = C
        A = _rnd((M, K), 21, dev)
        W = _rnd((2 * F, K), 22, dev, scale=1.0 / math.sqrt(K))       # rows already in the interleaved order
        gs = (1.0 + _rnd((2 * F,), 23, dev, scale=0.1, dtype=torch.float32)) if use_gs else None
        gscale = math.sqrt(C) if use_gs else 1.0
        assert ops.fusable(BF16, M, 2 * F, K)
        uv, xm = ops.gemm_nt_swiglu(A, W, M, F, K, gs, gscale)
        acc = A[rows].float() @ W.float().t()
        e_uv = (uv[rows].float() - acc).abs().max().item()
        z = acc * (gs * gscale) if use_gs else acc
        zz = z.reshape(-1, F // 16, 2, 16)
        want = (zz[:, :, 0] * (zz[:, :, 1] * torch.sigmoid(zz[:, :, 1]))).reshape(-1, F)
        e_x = (xm[rows].float() - want).abs().max().item()
        rep.add(f"gemm_nt_swiglu (EPI 3) F={F}: raw uv", e_uv, 2.0 ** -8 * max(1.0, acc.abs().max().item()))
        rep.add(f"gemm_nt_swiglu (EPI 3) F={F}: gated x", e_x, 2.0 ** -7 * max(1.0, want.abs().max().item()))
        # ------------------------------------------------------------ EPI 5: SwiGLU backward on the same uv
        dy = _rnd((M, C), 24, dev, scale=0.05)
        Wt = _rnd((F, C), 25, dev, scale=1.0 / math.sqrt(C))          # [F, K=C] transposed shadow of the down proj
        gs_nat = None
        if use_gs:   # natural-order suv matching the interleaved gs above
            gs_nat = torch.empty_like(gs)
            gi = gs.reshape(F // 16, 2, 16)
            gs_nat[:F] = gi[:, 0].reshape(F)
            gs_nat[F:] = gi[:, 1].reshape(F)
        duv, part = ops.gemm_nt_swiglu_bwd(dy, Wt, uv, M, F, C, gs_nat, gscale)
        dx = (dy[rows].float() @ Wt.float().t()).to(bf).float()       # the kernel packs dx to bf16 first
        uvr = uv[rows].float().reshape(-1, F // 16, 2, 16)
        ur, vr = uvr[:, :, 0].reshape(-1, F), uvr[:, :, 1].reshape(-1, F)
        gu = gs_nat[:F] * gscale if use_gs else 1.0
        gv = gs_nat[F:] * gscale if use_gs else 1.0
        u, v = ur * gu, vr * gv
        sg = torch.sigmoid(v)
        du = dx * v * sg
        dv = dx * u * sg * (1.0 + v * (1.0 - sg))
        want_duv = _interleave(torch.cat((du * gu, dv * gv), dim=1), F)
        e_d = (duv[rows].float() - want_duv).abs().max().item()
        rep.add(f"gemm_nt_swiglu_bwd (EPI 5) F={F}: duv", e_d, 2.0 ** -6 * max(1e-3, want_duv.abs().max().item()))
        if use_gs:
            # d(suv): full-column sums; reference over ALL rows in chunks (fp32 on device)
            ds = torch.empty(2 * F, device=dev)
            ops.colsum_reduce(part, ds, False)
            ref = torch.zeros(2 * F, device=dev, dtype=torch.float64)
            for r0 in range(0, M, 12544):
                sl = slice(r0, min(M, r0 + 12544))
                dxc = (dy[sl].float() @ Wt.float().t()).to(bf).float()
                uvc = uv[sl].float().reshape(-1, F // 16, 2, 16)
                urc, vrc = uvc[:, :, 0].reshape(-1, F), uvc[:, :, 1].reshape(-1, F)
                uc, vc = urc * gu, vrc * gv
                sgc = torch.sigmoid(vc)
                ref[:F] += (dxc * vc * sgc * urc).double().sum(0) * gscale
                ref[F:] += (dxc * uc * sgc * (1.0 + vc * (1.0 - sgc)) * vrc).double().sum(0) * gscale
            e_s = (ds.double() - ref).abs().max().item()
            rep.add(f"gemm_nt_swiglu_bwd (EPI 5) F={F}: d(suv) column sums", e_s, 2e-3 * max(1.0, ref.abs().max().item()))
        del A, W, uv, xm, dy, Wt, duv

    # ---------------------------------------------------------------- EPI 4: q/k/v projection + normalise + head split
    for nparts, part0 in ((3, 0), (1, 0), (2, 1)):
        A = _rnd((M, C), 31, dev)
        W = _rnd((nparts * C, C), 32, dev, scale=1.0 / math.sqrt(C))
        sqk = (1.0 / 32 + _rnd((C,), 33, dev, scale=0.003, dtype=torch.float32))
        c_q = 32.0
        bufs = ops.qk_buffers(BF16, B, T, H, d, dev)
        for t in bufs:
            t.fill_(float("nan"))
        qh, kh, vh, rq, rk = ops.gemm_nt_qknorm(A, W, M, C, nparts, part0, sqk, c_q, B, T, H, d, bufs)
        acc = A[rows].float() @ W.float().t()
        b_i, t_i = rows // T, rows % T
        worst, worst_r = 0.0, 0.0
        for j in range(nparts):
            part = part0 + j
            z = acc[:, j * C:(j + 1) * C].reshape(-1, H, d)
            got = (qh, kh, vh)[part][b_i, :, t_i, :].float()                  # [rows, H, d]
            if part < 2:
                nrm = z.norm(dim=-1, keepdim=True)
                want = z / nrm * (sqk * c_q).reshape(1, H, d)
                rn = (rq, rk)[part][rows]                                      # [rows, H]
                worst_r = max(worst_r, (rn * nrm.squeeze(-1) - 1.0).abs().max().item())
            else:
                want = z
            worst = max(worst, ((got - want).abs().max() / max(1.0, want.abs().max().item())).item())
        rep.add(f"gemm_nt_qknorm (EPI 4) nparts={nparts} part0={part0}: q/k/v heads", worst, 2.0 ** -7)
        if part0 < 2:
            rep.add(f"gemm_nt_qknorm (EPI 4) nparts={nparts} part0={part0}: 1/norm", worst_r, 1e-5)
        del A, W, bufs, acc

    # ---------------------------------------------------------------- TN (weight-gradient) GEMMs, full output vs fp64
    tn_cases = [(C, 4 * C, 0), (8 * C, C, 1), (C, C, 0), (3 * C, C, 0), (2 * C, C, 1), (C, 192, 0), (192, C, 0)]
    for N, K, perm in tn_cases:
        A = _rnd((M, N), 41, dev, scale=0.05)
        Bm = _rnd((M, K), 42, dev)
        G = torch.full((N, K), float("nan"), device=dev)
        ops.gemm_tn(A, Bm, G, M, N, K, perm=perm)
        ref = torch.zeros((N, K), device=dev, dtype=torch.float64)
        for r0 in range(0, M, 25088):
            sl = slice(r0, min(M, r0 + 25088))
            ref += A[sl].double().t() @ Bm[sl].double()
        if perm:
            full = torch.empty_like(ref)
            full[_perm_rows(N).to(dev)] = ref
            ref = full
        err = (G.double() - ref).abs().max().item()
        tol = 1e-6 * M * A.float().abs().mean().item() * Bm.float().abs().mean().item()
        rep.add(f"gemm_tn Mred={M} N={N} K={K} perm={perm} (splits {ops.tn_splits(M, N, K, BF16)})", err, tol)
        G2 = G.clone()
        ops.gemm_tn(A, Bm, G2, M, N, K, perm=perm, accumulate=True)
        rep.add(f"gemm_tn Mred={M} N={N} K={K} perm={perm} accumulate", (G2.double() - 2 * ref).abs().max().item(), 2 * tol)
        del A, Bm, G, G2, ref

    # ---------------------------------------------------------------- attention forward / backward at B*H = 1536
    scale = math.sqrt(d)
    q = (1.2 * torch.nn.functional.normalize(_rnd((B, H, T, d), 51, dev, dtype=torch.float32), dim=-1)).to(bf)
    k = (1.2 * torch.nn.functional.normalize(_rnd((B, H, T, d), 52, dev, dtype=torch.float32), dim=-1)).to(bf)
    v = _rnd((B, H, T, d), 53, dev)
    g_tok = _rnd((M, C), 54, dev)
    o, lse = ops.attn_fwd(BF16, 1, q, k, v, scale)
    dq, dk, dv = ops.attn_bwd(BF16, 1, g_tok, q, k, v, o, lse, scale)
    sample = [(0, 0), (0, H - 1), (B // 2, H // 2), (B - 1, 0), (B - 1, H - 1), (77 % B, 5 % H), (B - 2, 3 % H), (1, 1)]
    e_o = e_l = 0.0
    e_g = {"dq": 0.0, "dk": 0.0, "dv": 0.0}
    for (b, h) in sample:
        qf = q[b, h].float().requires_grad_(True)
        kf = k[b, h].float().requires_grad_(True)
        vf = v[b, h].float().requires_grad_(True)
        s = (qf @ kf.t()) * scale
        o_ref = torch.softmax(s, dim=-1) @ vf
        lse_ref = torch.logsumexp(s, dim=-1)
        gg = g_tok.reshape(B, T, H, d)[b, :, h].float()
        o_ref.backward(gg)
        e_o = max(e_o, (o.reshape(B, T, H, d)[b, :, h].float() - o_ref.detach()).abs().max().item())
        e_l = max(e_l, (lse[b, h] - lse_ref.detach()).abs().max().item())
        for name, got, ref in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
            e_g[name] = max(e_g[name], ((got[b, h].float() - ref).abs().max() / max(1.0, ref.abs().max().item())).item())
    rep.add(f"attn_fwd B*H={B * H} T={T}: O", e_o, 1e-2)
    rep.add(f"attn_fwd B*H={B * H} T={T}: lse", e_l, 1e-4)
    for name in ("dq", "dk", "dv"):
        rep.add(f"attn_bwd B*H={B * H} T={T}: {name} (relative to max)", e_g[name], 3e-2)
    # fused variant (q/k-normalise backward in the epilogues) agrees with unfused + qknorm_bwd on every row
    sqk = (1.0 / 32 + _rnd((C,), 55, dev, scale=0.003, dtype=torch.float32))
    rq = 1.0 + _rnd((M, H), 56, dev, scale=0.1, dtype=torch.float32).abs()
    rk = 1.0 + _rnd((M, H), 57, dev, scale=0.1, dtype=torch.float32).abs()
    dqkv = torch.full((M, 3 * C), float("nan"), device=dev, dtype=bf)
    pq, pk = ops.attn_bwd_qknorm(g_tok, q, k, v, o, lse, scale, rq, rk, sqk, 32.0, dqkv, 3 * C, dqkv[:, C:], dqkv[:, 2 * C:],
                                 3 * C)
    dqkv_u = torch.empty_like(dqkv)
    part = ops.qknorm_bwd(BF16, dq, dk, dv, q, k, rq, rk, sqk, 32.0, dqkv_u, 3 * C, dqkv_u[:, C:], 3 * C,
                          dqkv_u[:, 2 * C:], 3 * C, B, T, H, d)
    e_f = (dqkv.float() - dqkv_u.float()).abs().max().item()
    rep.add("attn_bwd_qknorm (fused epilogues) vs attn_bwd + qknorm_bwd: dqkv", e_f,
            2.0 ** -6 * max(1e-3, dqkv_u.float().abs().max().item()))
    ds_f = torch.empty(C, device=dev)
    ops.colsum_reduce(pq, ds_f, False, kind=0, scale=32.0)
    ops.colsum_reduce(pk, ds_f, True, kind=0, scale=32.0)
    ds_u = torch.empty(C, device=dev)
    ops.colsum_reduce(part, ds_u, False, kind=0, scale=32.0)
    rep.add("attn_bwd_qknorm: d(sqk)", (ds_f - ds_u).abs().max().item(), 2e-2 * max(1e-3, ds_u.abs().max().item()))
    del q, k, v, o, dq, dk, dv, dqkv, dqkv_u

    # ---------------------------------------------------------------- LERP (+norm_skip) forward at full M
    h = torch.nn.functional.normalize(_rnd((M, C), 61, dev, dtype=torch.float32), dim=-1)
    y = _rnd((M, C), 62, dev, scale=0.3, dtype=torch.float32)
    xs = torch.nn.functional.normalize(_rnd((M, C), 63, dev, dtype=torch.float32), dim=-1)
    alpha = 1.0 / 32 + _rnd((C,), 64, dev, scale=0.003, dtype=torch.float32)
    skip = torch.tensor([0.9], device=dev)
    out, out_lo = ops.lerp_fwd(BF16, h, y, alpha, 1.6, skip_x=xs, skip=skip, want_lo=True)
    lam = (alpha * 1.6).abs()
    a_, b_ = h[rows], torch.nn.functional.normalize(y[rows], dim=-1)
    r = torch.nn.functional.normalize(a_ + lam * (b_ - a_), dim=-1)
    r = torch.nn.functional.normalize(r * 0.9 + xs[rows], dim=-1)
    rep.add(f"lerp_fwd + norm_skip M={M}", (out[rows] - r).abs().max().item(), 2e-6)
    rep.add(f"lerp_fwd bf16 twin M={M}", (out_lo[rows].float() - out[rows]).abs().max().item(), 2.0 ** -8 * 0.2)
    torch.cuda.synchronize()
    return rep


if __name__ == "__main__":
    rep = check_all()
    print("ALL OK" if rep.ok else "FAILURES: " + ", ".join(k for k, r in rep.rows.items() if not r["ok"]))
    sys.exit(0 if rep.ok else 1)
