"""GPU parity tests, op level: every C-ABI entry point against fp32 torch math of the same op
(the oracle's building blocks).  Run on the MI355X box: pytest -m gpu."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nvit_oracle as O


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def ops_():
    from nvit_amd import ops
    return ops


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 192), (777, 1000, 128), (64, 10, 64), (513, 384, 768),
                                   (16500, 1000, 128), (33000, 520, 64), (25088, 768, 192)])  # last 3: persistent kernel
def test_gemm_nt(dtype, M, N, K):
    ops = ops_()
    A = rnd(M, K, seed=1).to(dtype)
    B = rnd(N, K, seed=2).to(dtype)
    bias = rnd(N, seed=3)
    cs = rnd(N, seed=4)
    period = 7
    radd = rnd(period, N, seed=5)
    ref = (A.float() @ B.float().t() + bias) * cs + radd[torch.arange(M) % period]
    out = ops.gemm_nt(A.to(dev()), B.to(dev()), M, N, K, bias=bias.to(dev()), colscale=cs.to(dev()),
                      rowadd=radd.to(dev()), rowadd_period=period)
    err = (out.cpu() - ref).abs().max().item()
    assert err < TOL[dtype] * math.sqrt(K) * 4, err
    # accumulate + bf16 output
    base = rnd(M, N, seed=6)
    out2 = base.to(dev()).clone()
    ops.gemm_nt(A.to(dev()), B.to(dev()), M, N, K, out=out2, accumulate=True)
    ref2 = A.float() @ B.float().t() + base
    assert (out2.cpu() - ref2).abs().max().item() < TOL[dtype] * math.sqrt(K) * 4
    out3 = ops.gemm_nt(A.to(dev()), B.to(dev()), M, N, K, out_dtype=torch.bfloat16)
    assert (out3.float().cpu() - A.float() @ B.float().t()).abs().max().item() < 0.02 * math.sqrt(K) * 4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Mred,N,K,perm", [(256, 128, 128, 0), (1000, 192, 72, 0), (5000, 256, 64, 1), (130, 64, 200, 0),
                                           (5000, 256, 256, 0), (9001, 512, 256, 1), (20000, 256, 768, 0),
                                           (37, 40, 16, 0)])
def test_gemm_tn(dtype, Mred, N, K, perm):
    ops = ops_()
    A = rnd(Mred, N, seed=1).to(dtype)
    B = rnd(Mred, K, seed=2).to(dtype)
    ref = A.float().t() @ B.float()
    if perm:
        idx = torch.tensor([(s // 32) * 16 + s % 32 if s % 32 < 16 else N // 2 + (s // 32) * 16 + (s % 32 - 16)
                            for s in range(N)])
        full = torch.zeros_like(ref)
        full[idx] = ref
        ref = full
    G = torch.full((N, K), 7.0, device=dev())
    ops.gemm_tn(A.to(dev()), B.to(dev()), G, Mred, N, K, perm=perm)
    tol = TOL[dtype] * math.sqrt(Mred) * 4
    assert (G.cpu() - ref).abs().max().item() < tol
    ops.gemm_tn(A.to(dev()), B.to(dev()), G, Mred, N, K, perm=perm, accumulate=True)
    assert (G.cpu() - 2 * ref).abs().max().item() < 2 * tol


def test_renorm_and_shadow():
    ops = ops_()
    from nvit_amd._lib import BF16, F32
    ws = [rnd(96, 64, seed=1), rnd(64, 256, seed=2), rnd(512, 64, seed=3), rnd(768, 100, seed=4), rnd(50, 36, seed=5)]
    dims = [1, 0, 1, 0, 1]
    dws = [w.to(dev()).contiguous() for w in ws]
    table, items = ops.renorm_table(list(zip(dws, dims)), dev())
    ops.renorm_weights(table, items)
    for w, d, dw in zip(ws, dims, dws):
        ref = w / w.norm(dim=d, keepdim=True)
        assert (dw.cpu() - ref).abs().max().item() < 2e-7
    # shadows: plain + transpose (+pad), SwiGLU interleave
    src = rnd(64, 40, seed=7).to(dev())
    for dt, td in ((F32, torch.float32), (BF16, torch.bfloat16)):
        dst = torch.full((64, 48), 9.0, device=dev(), dtype=td)
        dstT = torch.full((40, 128), 9.0, device=dev(), dtype=td)
        t, n = ops.shadow_table([(src, dst, 48, 48, dstT, 128, 100, 1)], dev())
        ops.shadow_weights(t, n, dt)
        perm = torch.tensor([(s // 32) * 16 + s % 32 if s % 32 < 16 else 32 + (s // 32) * 16 + (s % 32 - 16)
                             for s in range(64)])
        want = src.cpu()[perm].to(td).float()
        assert torch.equal(dst.float().cpu()[:, :40], want)
        assert torch.equal(dst.float().cpu()[:, 40:], torch.zeros(64, 8))
        assert torch.equal(dstT.float().cpu()[:, :64], want.t())
        assert torch.equal(dstT.float().cpu()[:, 64:100], torch.zeros(40, 36))
        assert torch.equal(dstT.float().cpu()[:, 100:], torch.full((40, 28), 9.0))


def _split_image(w):
    """host restatement of the perm=2 shadow layout: per 32 columns one [hi32 | lo32] slice, zero padded"""
    rows, K = w.shape
    Kp = (K + 31) // 32 * 32
    hi = w.bfloat16()
    lo = (w - hi.float()).bfloat16()
    img = torch.zeros(rows, 2 * Kp, dtype=torch.bfloat16)
    for j in range(Kp // 32):
        n = min(32, K - 32 * j)
        img[:, 64 * j: 64 * j + n] = hi[:, 32 * j: 32 * j + n]
        img[:, 64 * j + 32: 64 * j + 32 + n] = lo[:, 32 * j: 32 * j + n]
    return img, hi, lo


def test_shadow_split_precision_image():
    """perm=2: per 32 patch elements a [hi32 | lo32] bf16 slice of a patch-embedding weight; hi + lo reproduces the
    fp32 value to 2^-16; the padding of the last slice is left as allocated (zero)."""
    ops = ops_()
    from nvit_amd._lib import BF16
    w = rnd(100, 72, seed=9, scale=0.1)
    Kp = ops.patch_kp(72)
    assert Kp == 96
    dst = torch.zeros((100, 2 * Kp), device=dev(), dtype=torch.bfloat16)
    t, n = ops.shadow_table([(w.to(dev()), dst, 2 * Kp, 72, None, 0, 0, 2)], dev())
    ops.shadow_weights(t, n, BF16)
    want, hi, lo = _split_image(w)
    assert torch.equal(dst.cpu().float(), want.float())
    assert (hi.float() + lo.float() - w).abs().max().item() < 2.0 ** -16 * w.abs().max().item()


@pytest.mark.parametrize("B,ch,S,Pl,Pg,C", [
    (2, 3, 224, 8, 16, 768),     # Base geometry (T=784, K=192/768), M = 1568 -> a partial 256-token tile
    (3, 3, 224, 16, 32, 1024),   # 16/32 patches, C=1024
    (5, 3, 32, 4, 8, 192),       # pad = 2: runs straddle the border (scalar gather), K=48 -> zero-padded stage, C < 256
    (2, 3, 64, 8, 32, 320),      # pad = 12 > patch: deep reflection, C not a multiple of 256
    (2, 1, 48, 8, 8, 64),        # one channel, global window = local window
])
def test_fused_patch_embed(B, ch, S, Pl, Pg, C):
    """nvit_patch_embed_fwd (gather -> LDS -> split-operand MFMA, bias + pos epilogue) against fp64 im2col GEMMs of
    the un-rounded operands (reference model.py:286-304,407-415), and its saved bf16 patch rows bit for bit."""
    ops = ops_()
    from nvit_amd._lib import BF16
    d = dev()
    T = (S // Pl) ** 2
    M = B * T
    Kl, Kg = ch * Pl * Pl, ch * Pg * Pg
    img = rnd(B, ch, S, S, seed=1)
    wl, wg = rnd(C, Kl, seed=2, scale=Kl ** -0.5), rnd(C, Kg, seed=3, scale=Kg ** -0.5)
    bl, bg = rnd(C, seed=4, scale=0.1), rnd(C, seed=5, scale=0.1)
    pl, pg = rnd(T, C, seed=6, scale=0.02), rnd(T, C, seed=7, scale=0.02)
    sh = []
    ent = []
    for w, K in ((wl, Kl), (wg, Kg)):
        Kp = ops.patch_kp(K)
        sh.append(torch.zeros((C, 2 * Kp), device=d, dtype=torch.bfloat16))
        ent.append((w.to(d), sh[-1], 2 * Kp, K, None, 0, 0, 2))
    t, n = ops.shadow_table(ent, d)
    ops.shadow_weights(t, n, BF16)
    loc, glo, a_l, a_g, lo_l, lo_g = ops.patch_embed_fwd(img.to(d), sh[0], bl.to(d), pl.to(d), sh[1], bg.to(d), pg.to(d),
                                                         Pl, Pg, C, twins=True)
    # the bf16 twins are the rounded fp32 outputs, bit for bit
    assert torch.equal(lo_l.cpu(), loc.cpu().bfloat16()) and torch.equal(lo_g.cpu(), glo.cpu().bfloat16())
    Al = O.im2col(img, Pl, Pl, 0).reshape(M, Kl)
    Ag = O.im2col(img, Pg, Pl, (Pg - Pl) // 2).reshape(M, Kg)
    for out, A, w, b, pos, a_hi, K in ((loc, Al, wl, bl, pl, a_l, Kl), (glo, Ag, wg, bg, pg, a_g, Kg)):
        ref = (A.double() @ w.double().t() + b.double()).reshape(B, T, C) + pos.double()
        err = (out.cpu().double().reshape(B, T, C) - ref).abs().max().item()
        # missing lo*lo term: 2^-16 per product, random signs over K terms; fp32 accumulation
        assert err < 1e-5 * max(1.0, ref.abs().max().item()), (err, K)
        rows = a_hi.cpu()[:M]
        assert torch.equal(rows[:, :K].float(), A.bfloat16().float())
        assert torch.equal(rows[:, K:].float(), torch.zeros(M, rows.shape[1] - K))
    # without bias and without the saved rows
    loc2, glo2, a2, _, lo2, _ = ops.patch_embed_fwd(img.to(d), sh[0], None, pl.to(d), sh[1], None, pg.to(d), Pl, Pg, C,
                                                    save_rows=False)
    assert a2 is None and lo2 is None
    assert (loc2.cpu() + bl - loc.cpu()).abs().max().item() < 1e-6 and (glo2.cpu() + bg - glo.cpu()).abs().max().item() < 1e-6


@pytest.mark.parametrize("C", [64, 192, 768, 1024])
@pytest.mark.parametrize("with_skip", [False, True])
def test_lerp_fwd_bwd(C, with_skip):
    ops = ops_()
    from nvit_amd._lib import F32
    M = 333
    h = rnd(M, C, seed=1).requires_grad_(True)
    y = rnd(M, C, seed=2, scale=0.3).requires_grad_(True)
    alpha = (rnd(C, seed=3, scale=0.01) + 1 / 32).requires_grad_(True)
    xs = rnd(M, C, seed=4).requires_grad_(True)
    skip = torch.tensor([0.9], requires_grad=True)
    c_a = 0.05 * 32
    out = O.lerp(h, y, alpha, c_a)
    if with_skip:
        out = O.nrm(out * skip + xs)
    g = rnd(M, C, seed=5)
    out.backward(g)
    d = dev()
    got, got_lo = ops.lerp_fwd(F32, h.detach().to(d), y.detach().to(d), alpha.detach().to(d), c_a,
                               skip_x=xs.detach().to(d) if with_skip else None,
                               skip=skip.detach().to(d) if with_skip else None, want_lo=True)
    assert (got.cpu() - out.detach()).abs().max().item() < 1e-6
    assert torch.equal(got_lo, got)
    dh, dy, dy_lo, dxs, part, pskip = ops.lerp_bwd(F32, g.to(d), h.detach().to(d), y.detach().to(d),
                                                   alpha.detach().to(d), c_a,
                                                   xs.detach().to(d) if with_skip else None,
                                                   skip.detach().to(d) if with_skip else None, None, False, True, True)
    assert (dh.cpu() - h.grad).abs().max().item() < 2e-6 * max(1.0, h.grad.abs().max().item())
    assert (dy.cpu() - y.grad).abs().max().item() < 2e-6 * max(1.0, y.grad.abs().max().item())
    da = torch.empty(C, device=d)
    ops.colsum_reduce(part, da, False, kind=1, ref=alpha.detach().to(d), scale=c_a)
    assert (da.cpu() - alpha.grad).abs().max().item() < 1e-4 * max(1.0, alpha.grad.abs().max().item())
    if with_skip:
        assert (dxs.cpu() - xs.grad).abs().max().item() < 2e-6
        ds = torch.empty(1, device=d)
        ops.colsum_reduce(pskip, ds, False)
        assert abs(ds.item() - skip.grad.item()) < 1e-4 * max(1.0, abs(skip.grad.item()))


@pytest.mark.parametrize("C", [128, 768, 1024, 1280])
@pytest.mark.parametrize("with_skip", [False, True])
def test_lerp_bwd_bf16_addend_and_accumulate(C, with_skip):
    """The backward row kernel as the bf16 step calls it: y stored in bf16, the incoming gradient = an fp32 tensor + a
    bf16 addend (the data-gradient GEMM's output), dh accumulated onto an existing tensor, dy written in bf16 only; W =
    ceil(C / 256) waves share a row (1, 3, 4 and 5 here), ragged row count, against fp64 autograd on the same values."""
    ops = ops_()
    from nvit_amd._lib import BF16
    M = 517
    d = dev()
    h32 = rnd(M, C, seed=11)
    yb = rnd(M, C, seed=12, scale=0.3).bfloat16()
    alpha32 = rnd(C, seed=13, scale=0.01) + 1 / 32
    xs32, g32 = rnd(M, C, seed=14), rnd(M, C, seed=15)
    addb = rnd(M, C, seed=16, scale=0.5).bfloat16()
    old = rnd(M, C, seed=17)
    skip32 = torch.tensor([0.9])
    c_a = 0.05 * 32
    h, y, alpha, xs, skip = (t.double().requires_grad_(True) for t in (h32, yb.float(), alpha32, xs32, skip32))
    out = O.lerp(h, y, alpha, c_a)
    if with_skip:
        out = O.nrm(out * skip + xs)
    out.backward(g32.double() + addb.double())
    dh0 = old.clone().to(d)
    dh, dy, dy_lo, dxs, part, pskip = ops.lerp_bwd(BF16, g32.to(d), h32.to(d), yb.to(d), alpha32.to(d), c_a,
                                                   xs32.to(d) if with_skip else None, skip32.to(d) if with_skip else None,
                                                   dh0, True, False, True, dout_add=addb.to(d))
    assert dh.data_ptr() == dh0.data_ptr() and dy is None
    want_dh = (old.double() + h.grad)
    assert (dh.cpu().double() - want_dh).abs().max().item() < 3e-6 * max(1.0, want_dh.abs().max().item())
    assert (dy_lo.float().cpu().double() - y.grad).abs().max().item() < 5e-3 * y.grad.abs().max().item()   # one bf16 rounding
    da = torch.empty(C, device=d)
    ops.colsum_reduce(part, da, False, kind=1, ref=alpha32.to(d), scale=c_a)
    assert (da.cpu().double() - alpha.grad).abs().max().item() < 1e-4 * max(1.0, alpha.grad.abs().max().item())
    if with_skip:
        assert (dxs.cpu().double() - xs.grad).abs().max().item() < 3e-6 * max(1.0, xs.grad.abs().max().item())
        ds = torch.empty(1, device=d)
        ops.colsum_reduce(pskip, ds, False)
        assert abs(ds.item() - skip.grad.item()) < 1e-4 * max(1.0, abs(skip.grad.item()))


@pytest.mark.parametrize("H,d", [(2, 32), (3, 64), (12, 64)])
def test_qknorm_fwd_bwd(H, d):
    ops = ops_()
    from nvit_amd._lib import F32
    B, T = 2, 37
    C = H * d
    M = B * T
    qkv = rnd(M, 3 * C, seed=1).requires_grad_(True)
    sqk = (rnd(C, seed=2, scale=0.003) + 1 / 32).requires_grad_(True)
    c_q = 32.0
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    s = (sqk * c_q).reshape(1, H, 1, d)
    qh = s * O.nrm(O.heads(q.reshape(B, T, C), H))
    kh = s * O.nrm(O.heads(k.reshape(B, T, C), H))
    vh = O.heads(v.reshape(B, T, C), H)
    gq, gk, gv = rnd(B, H, T, d, seed=3), rnd(B, H, T, d, seed=4), rnd(B, H, T, d, seed=5)
    (qh * gq + kh * gk + vh * gv).sum().backward()
    dv_ = dev()
    dq = qkv.detach().to(dv_)
    gqh, gkh, gvh, rq, rk = ops.qknorm_fwd(F32, dq, 3 * C, dq[:, C:], 3 * C, dq[:, 2 * C:], 3 * C,
                                           sqk.detach().to(dv_), c_q, B, T, H, d)
    assert (gqh.cpu() - qh.detach()).abs().max().item() < 1e-6
    assert (gkh.cpu() - kh.detach()).abs().max().item() < 1e-6
    assert torch.equal(gvh.cpu(), vh.detach().contiguous())
    dqkv = torch.empty(M, 3 * C, device=dv_)
    part = ops.qknorm_bwd(F32, gq.to(dv_), gk.to(dv_), gv.to(dv_), gqh, gkh, rq, rk, sqk.detach().to(dv_), c_q, dqkv,
                          3 * C, dqkv[:, C:], 3 * C, dqkv[:, 2 * C:], 3 * C, B, T, H, d)
    assert (dqkv.cpu() - qkv.grad).abs().max().item() < 5e-6 * max(1.0, qkv.grad.abs().max().item())
    ds = torch.empty(C, device=dv_)
    ops.colsum_reduce(part, ds, False, kind=0, scale=c_q)
    assert (ds.cpu() - sqk.grad).abs().max().item() < 1e-4 * max(1.0, sqk.grad.abs().max().item())


def _interleave(x, F):
    # natural [.., 2F] (u | v) -> interleaved layout of the GEMM shadow (blocks of 16 u, 16 v)
    u, v = x[..., :F], x[..., F:]
    sh = x.shape[:-1]
    return torch.stack([u.reshape(*sh, F // 16, 16), v.reshape(*sh, F // 16, 16)], dim=-2).reshape(*sh, 2 * F)


@pytest.mark.parametrize("F,use_suv", [(64, True), (256, False), (3072, True)])
def test_swiglu_fwd_bwd(F, use_suv):
    ops = ops_()
    from nvit_amd._lib import F32
    M = 77
    uv = rnd(M, 2 * F, seed=1).requires_grad_(True)
    suv = (rnd(2 * F, seed=2, scale=0.1) + 1).requires_grad_(True)
    gscale = 3.0
    z = uv * (suv * gscale) if use_suv else uv
    u, v = z[:, :F], z[:, F:]
    x = u * (v * torch.sigmoid(v))
    g = rnd(M, F, seed=3)
    x.backward(g)
    d = dev()
    uvi = _interleave(uv.detach(), F).contiguous().to(d)
    got = ops.swiglu_fwd(F32, uvi, suv.detach().to(d) if use_suv else None, gscale if use_suv else 1.0, M, F)
    assert (got.cpu() - x.detach()).abs().max().item() < 2e-6 * max(1.0, x.detach().abs().max().item())
    duv, part = ops.swiglu_bwd(F32, g.to(d), uvi, suv.detach().to(d) if use_suv else None,
                               gscale if use_suv else 1.0, M, F)
    want = _interleave(uv.grad, F)
    assert (duv.cpu() - want).abs().max().item() < 5e-6 * max(1.0, want.abs().max().item())
    if use_suv:
        ds = torch.empty(2 * F, device=d)
        ops.colsum_reduce(part, ds, False)
        assert (ds.cpu() - suv.grad).abs().max().item() < 1e-4 * max(1.0, suv.grad.abs().max().item())


def _sdpa_ref(qh, kh, vh, scale):
    s = (qh @ kh.transpose(-1, -2)) * scale
    p = torch.softmax(s, dim=-1)
    return p @ vh, torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("dtype,impl", [(torch.float32, 0), (torch.bfloat16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("B,H,T,d", [(2, 2, 16, 32), (1, 3, 49, 64), (2, 2, 196, 64), (1, 1, 130, 64), (2, 3, 784, 64),
                                     (1, 2, 1, 64), (1, 1, 257, 64)])
def test_attention(dtype, impl, B, H, T, d):
    if impl == 1 and d != 64:
        pytest.skip("MFMA kernel: head dim 64 only")
    ops = ops_()
    from nvit_amd.ops import dt_of
    # |q|=|k|=1.3 per head: logits up to sqrt(d)*1.69 (sharper softmax than the init state)
    q = (1.3 * torch.nn.functional.normalize(rnd(B, H, T, d, seed=1), dim=-1)).to(dtype)
    k = (1.3 * torch.nn.functional.normalize(rnd(B, H, T, d, seed=2), dim=-1)).to(dtype)
    v = rnd(B, H, T, d, seed=3).to(dtype)
    scale = math.sqrt(d)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    o_ref, lse_ref = _sdpa_ref(qf, kf, vf, scale)
    g = rnd(B, H, T, d, seed=4).to(dtype)
    o_ref.backward(g.float())
    dv_ = dev()
    dt = dt_of(q)
    o, lse = ops.attn_fwd(dt, impl, q.to(dv_), k.to(dv_), v.to(dv_), scale)
    o_bhtd = o.float().cpu().reshape(B, T, H, d).permute(0, 2, 1, 3)
    tol = 2e-6 if dtype == torch.float32 else 1e-2
    assert (o_bhtd - o_ref.detach()).abs().max().item() < tol
    assert (lse.cpu() - lse_ref.detach()).abs().max().item() < 1e-4
    g_tok = g.permute(0, 2, 1, 3).reshape(B * T, H * d).contiguous()
    dq, dk, dv = ops.attn_bwd(dt, impl, g_tok.to(dv_), q.to(dv_), k.to(dv_), v.to(dv_), o, lse, scale)
    tolg = 5e-5 if dtype == torch.float32 else 3e-2
    for name, got, ref in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        e = (got.float().cpu() - ref).abs().max().item()
        lim = tolg * max(1.0, ref.abs().max().item())
        assert e < lim, f"{name}: err {e:.3e} >= {lim:.3e}"


@pytest.mark.parametrize("B,H,T", [(2, 3, 784), (1, 2, 130), (1, 1, 16), (2, 2, 64)])
@pytest.mark.parametrize("smul", [1.0, 1.6, 3.0])
def test_attention_bounded_scores(B, H, T, smul):
    """nvit_attn_fwd_bounded: q, k = (sqk*c_q) * unit vectors (the nViT form).  smul 1.0 / 1.6: the bound-relative fast
    path (no running max); smul 3.0: the bound exceeds the safe range and the kernel falls back to the online softmax.
    Against fp32 torch math and against the generic entry point on the same inputs."""
    ops = ops_()
    from nvit_amd._lib import BF16
    d = 64
    C = H * d
    c_q = 32.0
    sqk = (smul / 32.0) * (1.0 + 0.3 * torch.tanh(rnd(C, seed=7)))
    s_eff = (sqk * c_q).reshape(1, H, 1, d)
    q = (s_eff * torch.nn.functional.normalize(rnd(B, H, T, d, seed=1), dim=-1)).bfloat16()
    k = (s_eff * torch.nn.functional.normalize(rnd(B, H, T, d, seed=2), dim=-1)).bfloat16()
    v = rnd(B, H, T, d, seed=3).bfloat16()
    scale = math.sqrt(d)
    o_ref, lse_ref = _sdpa_ref(q.float(), k.float(), v.float(), scale)
    dv_ = dev()
    o, lse = ops.attn_fwd(BF16, 1, q.to(dv_), k.to(dv_), v.to(dv_), scale, sqk.to(dv_), c_q)
    o2, lse2 = ops.attn_fwd(BF16, 1, q.to(dv_), k.to(dv_), v.to(dv_), scale)
    o_bhtd = o.float().cpu().reshape(B, T, H, d).permute(0, 2, 1, 3)
    # (on the fast path the row sum is taken by the MFMA pipe over the bf16-rounded probabilities - the same values the
    #  output is built from, so O stays a properly normalised average - which moves lse by up to 2^-9 (one dominant key) in absolute terms)
    lse_tol = 4e-3 if smul < 3.0 else 1e-4 * max(1.0, lse_ref.abs().max().item())
    o_tol = 1e-2 + 2.0 ** -7 * o_ref.abs().max().item()   # + one bf16 ulp of the stored output at its largest magnitude
    assert (o_bhtd - o_ref).abs().max().item() < o_tol
    assert (lse.cpu() - lse_ref).abs().max().item() < lse_tol
    assert (o.float() - o2.float()).abs().max().item() < o_tol
    assert (lse - lse2).abs().max().item() < lse_tol
    # spike: one key aligned with one query at the largest possible score (the bound itself) must not overflow
    q2, k2 = q.clone(), k.clone()
    k2[0, 0, T // 2] = q2[0, 0, 0]
    o3, lse3 = ops.attn_fwd(BF16, 1, q2.to(dv_), k2.to(dv_), v.to(dv_), scale, sqk.to(dv_), c_q)
    o3_ref, lse3_ref = _sdpa_ref(q2.float(), k2.float(), v.float(), scale)
    assert torch.isfinite(o3.float()).all() and torch.isfinite(lse3).all()
    # (the spiked row copies one V row, |v| up to ~4)
    assert (o3.float().cpu().reshape(B, T, H, d).permute(0, 2, 1, 3) - o3_ref).abs().max().item() < 1e-2 + 2.0 ** -7 * o3_ref.abs().max().item()
    assert (lse3.cpu() - lse3_ref).abs().max().item() < lse_tol


def test_im2col_pool_recon():
    ops = ops_()
    from nvit_amd._lib import F32
    B, ch, S, Pl, Pg = 3, 3, 40, 8, 16
    img = rnd(B, ch, S, S, seed=1)
    A_l, A_g = ops.im2col(F32, img.to(dev()), Pl, Pg)
    T = (S // Pl) ** 2
    assert torch.equal(A_l.cpu().reshape(B, T, -1), O.im2col(img, Pl, Pl, 0))
    assert torch.equal(A_g.cpu().reshape(B, T, -1), O.im2col(img, Pg, Pl, (Pg - Pl) // 2))
    # pool + LN fwd/bwd
    C = 192
    x = rnd(B, T, C, seed=2).requires_grad_(True)
    w = (rnd(C, seed=3, scale=0.1) + 1).requires_grad_(True)
    b = rnd(C, seed=4, scale=0.1).requires_grad_(True)
    ln_ref = O.layer_norm(x.mean(dim=1), w, b)
    g = rnd(B, C, seed=5)
    ln_ref.backward(g)
    d = dev()
    pooled, ln, ln_lo, stats = ops.pool_ln_fwd(F32, x.detach().reshape(B * T, C).to(d), w.detach().to(d),
                                               b.detach().to(d), 1e-5, B, T, C)
    assert (ln.cpu() - ln_ref.detach()).abs().max().item() < 2e-6
    dw = torch.empty(C, device=d)
    db = torch.empty(C, device=d)
    dx = ops.pool_ln_bwd(g.to(d), pooled, w.detach().to(d), stats, dw, db, False, B, T, C)
    assert (dx.cpu().reshape(B, T, C) - x.grad).abs().max().item() < 1e-6
    assert (dw.cpu() - w.grad).abs().max().item() < 1e-5
    assert (db.cpu() - b.grad).abs().max().item() < 1e-5
    # recon loss
    K = ch * Pl * Pl
    raw = rnd(B * T, K, seed=6)
    ref = ((torch.tanh(raw).reshape(B, T, K) - O.im2col(img, Pl, Pl, 0)) ** 2).mean()
    got = ops.recon_loss(raw.to(d), img.to(d), Pl)
    assert abs(got.item() - ref.item()) < 1e-5 * max(1.0, ref.item())


def test_small_reductions_and_cast():
    ops = ops_()
    d = dev()
    a = rnd(300, 52, seed=1)
    b = rnd(300, 52, seed=2)
    out = torch.empty(52, device=d)
    ops.colsum(a.to(d), 300, 52, out, False, b=b.to(d), scale=0.5)
    assert (out.cpu() - 0.5 * (a * b).sum(0)).abs().max().item() < 1e-4
    outp = torch.empty(10, 52, device=d)
    ops.colsum(a.to(d), 300, 52, outp, False, period=10)
    assert (outp.cpu() - a.reshape(30, 10, 52).sum(0)).abs().max().item() < 1e-4
    big = rnd(5000, 64, seed=3)
    ob = torch.empty(64, device=d)
    ops.colsum_big(big.to(d), 5000, 64, ob, False)
    assert (ob.cpu() - big.sum(0)).abs().max().item() < 1e-3
    from nvit_amd._lib import BF16
    c = ops.cast(a.to(d).contiguous(), BF16)
    assert torch.equal(c.cpu(), a.to(torch.bfloat16))
    s = rnd(52, seed=4)
    o2 = ops.scale_cols(a.to(d), s.to(d), 2.0, 300, 52, torch.empty(300, 52, device=d))
    assert (o2.cpu() - a * s * 2.0).abs().max().item() < 1e-5


def test_fused_gemm_swiglu_and_qknorm_match_unfused():
    """Fused-epilogue persistent GEMMs (bf16) against the unfused kernel sequence on the same inputs."""
    ops = ops_()
    from nvit_amd._lib import BF16
    d_ = dev()
    # SwiGLU: M x (2F) with interleaved shadow
    M, F, K = 6000, 512, 256
    A = rnd(M, K, seed=1).bfloat16().to(d_)
    W = (rnd(2 * F, K, seed=2) * 0.05).bfloat16().to(d_)
    gs = (rnd(2 * F, seed=3, scale=0.1) + 1).to(d_)
    assert ops.fusable(BF16, M, 2 * F, K)
    uv, xm = ops.gemm_nt_swiglu(A, W, M, F, K, gs, 3.0)
    uv_ref = ops.gemm_nt(A, W, M, 2 * F, K, out_dtype=torch.bfloat16)
    assert torch.equal(uv, uv_ref)
    # reference gate in fp32 from the fp32 accumulators (uv_ref is rounded, so compare at bf16 tolerance)
    acc = A.float() @ W.float().t()
    z = acc * (gs * 3.0)
    zz = z.reshape(M, F // 16, 2, 16)
    want = (zz[:, :, 0] * (zz[:, :, 1] * torch.sigmoid(zz[:, :, 1]))).reshape(M, F)
    assert (xm.float() - want).abs().max().item() < 2e-2 * max(1.0, want.abs().max().item())
    # q/k normalise: 3 stacked projections, C = 256 (H = 4, d = 64), T = 50
    B, T, H, d = 12, 50, 4, 64
    C = H * d
    M = B * T
    X = rnd(M, C, seed=4).bfloat16().to(d_)
    Wqkv = (rnd(3 * C, C, seed=5) * 0.05).bfloat16().to(d_)
    sqk = (rnd(C, seed=6, scale=0.003) + 1 / 32).to(d_)
    qh, kh, vh, rq, rk = ops.gemm_nt_qknorm(X, Wqkv, M, C, 3, 0, sqk, 32.0, B, T, H, d)
    acc = (X.float() @ Wqkv.float().t()).reshape(B, T, 3, H, d).permute(2, 0, 3, 1, 4)   # [3,B,H,T,d]
    s = (sqk * 32.0).reshape(1, H, 1, d)
    nq = acc[0] / acc[0].norm(dim=-1, keepdim=True) * s
    nk = acc[1] / acc[1].norm(dim=-1, keepdim=True) * s
    assert (qh.float() - nq).abs().max().item() < 1e-2
    assert (kh.float() - nk).abs().max().item() < 1e-2
    assert (vh.float() - acc[2]).abs().max().item() < 2e-2 * acc[2].abs().max().item()
    rq_ref = (1.0 / acc[0].norm(dim=-1)).permute(0, 2, 1).reshape(M, H)
    assert (rq - rq_ref).abs().max().item() < 1e-3 * rq_ref.abs().max().item()


@pytest.mark.parametrize("M,use_suv", [(6000, True), (5123, True), (6000, False)])
def test_fused_gemm_swiglu_bwd_matches_unfused(M, use_suv):
    """Data-gradient GEMM with the SwiGLU backward in its epilogue vs nvit_gemm_nt + nvit_swiglu_bwd (ragged M too)."""
    ops = ops_()
    from nvit_amd._lib import BF16
    d_ = dev()
    F, K = 1024, 256
    dy = rnd(M, K, seed=11).bfloat16().to(d_)
    Wt = (rnd(F, K, seed=12) * 0.05).bfloat16().to(d_)          # transposed shadow of the [K, F] projection
    uv = rnd(M, 2 * F, seed=13).bfloat16().to(d_)                 # saved raw pre-activations (interleaved)
    suv = (rnd(2 * F, seed=14, scale=0.1) + 1).to(d_) if use_suv else None
    gscale = 1.7 if use_suv else 1.0
    assert ops.fusable(BF16, M, F, K)
    duv, part = ops.gemm_nt_swiglu_bwd(dy, Wt, uv, M, F, K, suv, gscale)
    dx = ops.gemm_nt(dy, Wt, M, F, K, out_dtype=torch.bfloat16)
    duv_ref, part_ref = ops.swiglu_bwd(BF16, dx, uv, suv, gscale, M, F)
    err = (duv.float() - duv_ref.float()).abs().max().item()
    assert err <= 2e-2 * max(1.0, duv_ref.float().abs().max().item()), err
    # tighter: mean error (both paths round dx to bf16 first, so they differ only by exp/rcp ulps)
    assert (duv.float() - duv_ref.float()).abs().mean().item() < 1e-3
    if use_suv:
        g = torch.empty(2 * F, device=d_)
        g_ref = torch.empty(2 * F, device=d_)
        ops.colsum_reduce(part, g, False)
        ops.colsum_reduce(part_ref, g_ref, False)
        assert (g - g_ref).abs().max().item() <= 2e-3 * g_ref.abs().max().item()
    else:
        assert part is None


@pytest.mark.parametrize("grad_clip", [0.0, 0.05])
def test_fused_adamw_renorm_matches_torch(grad_clip):
    """FusedAdamW.step_fused (clip + AdamW + row/column renorm in two launches) against the reference sequence
    clip_grad_norm_ -> torch.optim.AdamW.step -> x / ||x|| (train.py:935-946, 461-480) on CPU fp32, three steps."""
    from nvit_amd.optim import FusedAdamW
    d_ = dev()
    shapes = [((40, 768), 1), ((768, 40), 0), ((100, 260), 1), ((1000, 100), 0), ((517,), -1), ((30, 33), -1),
              ((8192 * 2 + 12,), -1), ((3, 5, 8, 8), -1)]
    ref = [torch.nn.Parameter(rnd(*s, seed=20 + i, scale=0.05)) for i, (s, _) in enumerate(shapes)]
    mine = [torch.nn.Parameter(p.detach().clone().to(d_)) for p in ref]
    groups = lambda ps: [{"params": [p for p in ps if p.dim() >= 2], "weight_decay": 0.1},
                         {"params": [p for p in ps if p.dim() < 2], "weight_decay": 0.0}]
    o_ref = torch.optim.AdamW(groups(ref), lr=1e-2, betas=(0.9, 0.95))
    o_my = FusedAdamW(groups(mine), lr=1e-2, betas=(0.9, 0.95))
    dims = {id(p): k for p, (_, k) in zip(mine, shapes) if k >= 0}
    for step in range(3):
        for i, (pr, pm) in enumerate(zip(ref, mine)):
            g = rnd(*pr.shape, seed=100 + 10 * step + i, scale=0.02)
            pr.grad = g.clone()
            pm.grad = g.to(d_)
        if grad_clip > 0:
            gn_ref = torch.nn.utils.clip_grad_norm_(ref, grad_clip)
        o_ref.step()
        with torch.no_grad():
            for pr, (_, k) in zip(ref, shapes):
                if k == 1:
                    pr.copy_(pr / pr.norm(dim=1, keepdim=True))
                elif k == 0:
                    pr.copy_(pr / pr.norm(dim=0, keepdim=True))
        gn = _step_with_dims(o_my, dims, grad_clip)
        if grad_clip > 0:
            assert abs(gn.item() - gn_ref.item()) <= 1e-5 * gn_ref.item()
        for pr, pm in zip(ref, mine):
            err = (pm.detach().cpu() - pr.detach()).abs().max().item()
            assert err <= 2e-6 * max(1.0, pr.detach().abs().max().item()), (step, tuple(pr.shape), err)
    # optimizer state matches torch's layout and values
    sd_ref, sd_my = o_ref.state_dict(), o_my.state_dict()
    assert sd_ref["state"].keys() == sd_my["state"].keys()
    for k in sd_ref["state"]:
        assert float(sd_my["state"][k]["step"]) == float(sd_ref["state"][k]["step"]) == 3.0
        for name in ("exp_avg", "exp_avg_sq"):
            a, b = sd_my["state"][k][name].cpu(), sd_ref["state"][k][name]
            assert (a - b).abs().max().item() <= 1e-6 * max(1e-3, b.abs().max().item())


def _step_with_dims(opt, dims, grad_clip):
    """FusedAdamW.step_fused with a hand-made renorm map (the public method builds it from ViT blocks)."""
    import types
    m = types.SimpleNamespace()
    m.config = types.SimpleNamespace(use_nvit=True)
    rows = [p for g in opt.param_groups for p in g["params"] if dims.get(id(p)) == 1]
    cols = [p for g in opt.param_groups for p in g["params"] if dims.get(id(p)) == 0]
    # pack the matrices into fake blocks: 4 row-normalised + 2 column-normalised slots each, padding with repeats
    blk = types.SimpleNamespace()
    W = lambda p: types.SimpleNamespace(weight=p)
    blk.query, blk.key, blk.value, blk.c_fc = W(rows[0]), W(rows[1]), W(rows[0]), W(rows[1])
    blk.att_c_proj, blk.mlp_c_proj = W(cols[0]), W(cols[1])
    m.transformer = types.SimpleNamespace(h=[blk])
    return opt.step_fused(m, grad_clip)


@pytest.mark.parametrize("B,N", [(8, 10), (128, 1000), (3, 7), (33, 100)])
def test_ce_loss_fwd_bwd(B, N):
    """nvit_ce_loss (loss + gradient in one pass) against F.cross_entropy and its autograd on CPU fp32."""
    from nvit_amd.train import CrossEntropyFn
    logits = rnd(B, N, seed=5, scale=3.0)
    y = torch.randint(0, N, (B,), generator=torch.Generator().manual_seed(6))
    lr = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, y)
    (ref * 1.7).backward()
    lg = logits.to(dev()).requires_grad_(True)
    out = CrossEntropyFn.apply(lg, y.to(dev()))
    (out * 1.7).backward()
    assert abs(out.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    assert (lg.grad.cpu() - lr.grad).abs().max().item() <= 2e-7


def test_normalize_images_input_pipeline():
    """Deterministic part of the reference input pipeline (train.py:1084-1090): ToTensor + Normalize(0.5, 0.5)."""
    ops = ops_()
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (5, 32, 32, 3), generator=g, dtype=torch.uint8)
    want = (u8.permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5
    got = ops.normalize_images(u8.to(dev()))
    assert got.shape == (5, 3, 32, 32) and (got.cpu() - want).abs().max().item() < 1e-6
    f = torch.rand(2, 3, 224, 224, generator=g)
    got = ops.normalize_images(f.to(dev()), 0.5, 0.5)
    assert (got.cpu() - (f - 0.5) / 0.5).abs().max().item() < 1e-6


@pytest.mark.parametrize("periodic", [True, False])
def test_kohonen_map_update_periodic_and_plain(periodic):
    """KohonenMap.update_nodes (reference kohonen.py:121-165) on both topologies (kohonen.py:80-98) against the oracle's
    sequential restatement, plus get_neighborhood_distances and the state_dict keys (`offsets` only when periodic)."""
    from nvit_amd.kohonen import KohonenMap
    d = dev()
    B, T, C, N = 6, 9, 32, 30            # 5 x 6 grid
    torch.manual_seed(3)
    km = KohonenMap(C, N, alpha=0.3, periodic=periodic).to(d).train()
    assert ("offsets" in km.state_dict()) == periodic
    x = rnd(B, T, C, seed=4)
    nodes0 = km.nodes.detach().cpu().clone()
    _, idx = km(x.to(d))
    ref_idx = torch.cdist(x.reshape(-1, C), nodes0).argmin(dim=-1).reshape(B, T)
    assert torch.equal(idx.cpu(), ref_idx)
    km.update_nodes(x.to(d), idx, 0.7)
    want = nodes0.clone()
    O.som_update_(want, x, ref_idx, 0.7, 0.3, periodic)
    assert (km.nodes.detach().cpu() - want).abs().max().item() < 2e-6
    bmu = torch.tensor([4, 5])
    got = km.get_neighborhood_distances(bmu.to(d)).cpu()
    assert torch.equal(got, O.som_neighborhood_d2(bmu, km.m, km.n, periodic))
    assert got[0].item() == (2.0 if periodic else 41.0)     # node (0,0): wrapped (1,1) away vs (4,5) away


@pytest.mark.parametrize("B,T", [(110, 50), (8, 784), (44, 130)])
def test_attention_q_prescale_path_matches_plain(B, T):
    """The training path folds sqrt(d)*log2(e) into q at its producer (GEMM epilogue 4) so that the attention kernels'
    exponent needs no multiply (accumulator seeded with -bound / -lse, straight into v_exp_f32).  Same inputs through
    the plain path (q_prescale = 1) and the pre-scaled one: q_hat differs only by its bf16 rounding, outputs, lse and
    every gradient (d qkv token-major, d sqk) must agree to bf16 rounding; and both against fp32 torch math."""
    ops = ops_()
    from nvit_amd._lib import BF16
    d_ = dev()
    H, d = 4, 64
    C = H * d
    M = B * T
    if not ops.fusable(BF16, M, 3 * C, C):
        pytest.skip("shape below the fused-epilogue threshold")
    X = rnd(M, C, seed=4).bfloat16().to(d_)
    Wqkv = (rnd(3 * C, C, seed=5) * 0.05).bfloat16().to(d_)
    sqk = ((1.0 + 0.2 * torch.tanh(rnd(C, seed=6))) / 32).to(d_)
    c_q, scale = 32.0, math.sqrt(d)
    qpre = ops.attn_q_prescale(d)
    g_tok = rnd(M, C, seed=8).bfloat16().to(d_)
    res = {}
    for tag, qp in (("plain", 1.0), ("pre", qpre)):
        qh, kh, vh, rq, rk = ops.gemm_nt_qknorm(X, Wqkv, M, C, 3, 0, sqk, c_q, B, T, H, d, q_prescale=qp)
        o, lse = ops.attn_fwd(BF16, 1, qh, kh, vh, scale, sqk, c_q, q_prescale=qp)
        dqkv = torch.empty((M, 3 * C), device=d_, dtype=torch.bfloat16)
        pq, pk = ops.attn_bwd_qknorm(g_tok, qh, kh, vh, o, lse, scale, rq, rk, sqk, c_q, dqkv, 3 * C, dqkv[:, C:],
                                     dqkv[:, 2 * C:], 3 * C, q_prescale=qp)
        dsq = torch.empty(C, device=d_)
        ops.colsum_reduce(pq, dsq, False, kind=0, scale=c_q)
        ops.colsum_reduce(pk, dsq, True, kind=0, scale=c_q)
        res[tag] = (qh.float(), kh.float(), o.float(), lse, dqkv.float(), dsq)
    qh0, kh0, o0, lse0, dqkv0, dsq0 = res["plain"]
    qh1, kh1, o1, lse1, dqkv1, dsq1 = res["pre"]
    assert (qh1 / qpre - qh0).abs().max().item() < 2.0 ** -7 * qh0.abs().max().item()      # two bf16 roundings (half an ulp each, different binades) apart
    assert torch.equal(kh1, kh0)
    assert (o1 - o0).abs().max().item() < 1e-2 + 2.0 ** -7 * o0.abs().max().item()
    assert (lse1 - lse0).abs().max().item() < 1e-2      # (scores up to ~10 from two differently rounded q_hat: 2^-9 relative each)
    assert (dqkv1 - dqkv0).abs().max().item() < 2e-2 * max(1e-3, dqkv0.abs().max().item())
    assert (dsq1 - dsq0).abs().max().item() < 2e-2 * max(1e-3, dsq0.abs().max().item())
    # fp32 torch math from the same bf16 operands (autograd through normalise -> scale -> attention)
    Xf = X.float().requires_grad_(True)
    sq = sqk.clone().requires_grad_(True)
    acc = (Xf @ Wqkv.float().t()).reshape(B, T, 3, H, d).permute(2, 0, 3, 1, 4)
    s_eff = (sq * c_q).reshape(1, H, 1, d)
    q = acc[0] / acc[0].norm(dim=-1, keepdim=True) * s_eff
    k = acc[1] / acc[1].norm(dim=-1, keepdim=True) * s_eff
    oref, lref = _sdpa_ref(q, k, acc[2], scale)
    o_tok = oref.permute(0, 2, 1, 3).reshape(M, C)
    o_tok.backward(g_tok.float())
    assert (o1 - o_tok.detach()).abs().max().item() < 2e-2 * max(1.0, o_tok.abs().max().item())
    assert (lse1 - lref.detach()).abs().max().item() < 1e-2
    assert (dsq1 - sq.grad).abs().max().item() < 3e-2 * max(1e-3, sq.grad.abs().max().item())
    # d(qkv) feeds the data-gradient GEMM: compare through it, dX = dqkv @ Wqkv
    dX = dqkv1 @ Wqkv.float()
    assert (dX - Xf.grad).abs().max().item() < 3e-2 * max(1e-3, Xf.grad.abs().max().item())


@pytest.mark.parametrize("C", [64, 192, 768, 1024])
def test_rmsnorm_module_fwd_bwd(C):
    """nvit_amd.model.RMSNorm (reference model.py:170-182) against the formula in fp64 torch math, values and gradients."""
    from nvit_amd.model import RMSNorm
    d = dev()
    x = rnd(5, 37, C, seed=1).requires_grad_(True)
    w = (1.0 + rnd(C, seed=2, scale=0.2))
    g = rnd(5, 37, C, seed=3)
    xd, wd = x.detach().double().requires_grad_(True), w.double().requires_grad_(True)
    ref = xd * torch.rsqrt((xd * xd).mean(dim=-1, keepdim=True) + 1e-6) * wd
    ref.backward(g.double())
    mod = RMSNorm(C).to(d)
    with torch.no_grad():
        mod.weight.copy_(w)
    xg = x.detach().to(d).requires_grad_(True)
    out = mod(xg)
    out.backward(g.to(d))
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 2e-6
    assert (xg.grad.cpu().double() - xd.grad).abs().max().item() < 5e-6
    assert (mod.weight.grad.cpu().double() - wd.grad).abs().max().item() < 2e-5 * max(1.0, wd.grad.abs().max().item())


@pytest.mark.parametrize("B,H,T", [(2, 2, 64), (1, 2, 200), (2, 3, 784), (1, 1, 16), (2, 2, 49), (1, 2, 833)])
def test_attn_bwd_dkv_hand_placed_loop_is_bit_exact(B, H, T):
    """The generated-assembly main loop of the dK/dV kernel (nvit_amd/csrc/gen/gen_attn_dkv32_asm.py) against the
    compiler-built kernel on the fused backward entry point (pre-scaled q, as the training path calls it): dq | dk | dv and
    the sqk partial sums must be IDENTICAL, bit for bit - full and ragged tiles (T = 16, 49, 200, 784, 833), key blocks with
    idle waves, one and several (batch, head) pairs."""
    from nvit_amd import _lib
    from nvit_amd._lib import BF16
    ops = ops_()
    lib = _lib.load()
    d_, dv_ = 64, dev()
    C, M = H * d_, B * T
    g = torch.Generator().manual_seed(B * 1000 + T)
    rn = lambda *s: torch.randn(*s, generator=g)
    sqk = ((1.0 / 32) * (1.0 + 0.05 * torch.tanh(rn(C)))).to(dv_)
    se = (sqk.cpu() * 32.0).reshape(1, H, 1, d_)
    qpre = ops.attn_q_prescale(d_)
    qs = (se * torch.nn.functional.normalize(rn(B, H, T, d_), dim=-1) * qpre).bfloat16().to(dv_)
    k = (se * torch.nn.functional.normalize(rn(B, H, T, d_), dim=-1)).bfloat16().to(dv_)
    v = (rn(B, H, T, d_) * 0.05).bfloat16().to(dv_)
    gt = (rn(M, C) * 1e-3).bfloat16().to(dv_)
    rq, rk = (1.0 + rn(M, H).abs() * 0.1).to(dv_), (1.0 + rn(M, H).abs() * 0.1).to(dv_)
    scale = math.sqrt(d_)
    o, lse = ops.attn_fwd(BF16, 1, qs, k, v, scale, sqk, 32.0, q_prescale=qpre)
    outs = []
    try:
        for mode in (0, 1):
            lib.nvit_set_attn_dkv_asm(mode)
            dqkv = torch.zeros(M, 3 * C, device=dv_, dtype=torch.bfloat16)
            pq, pk = ops.attn_bwd_qknorm(gt, qs, k, v, o, lse, scale, rq, rk, sqk, 32.0, dqkv, 3 * C, dqkv[:, C:], dqkv[:, 2 * C:],
                                         3 * C, q_prescale=qpre)
            torch.cuda.synchronize()
            outs.append((dqkv.clone(), pq.clone(), pk.clone()))
    finally:
        lib.nvit_set_attn_dkv_asm(1)
    (a, pqa, pka) = outs[0]
    assert a[:, C:].float().abs().max().item() > 0
    for (b, pqb, pkb) in outs[1:]:
        assert torch.isfinite(b.float()).all()
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        assert torch.equal(pqa, pqb) and torch.equal(pka, pkb)
