"""GPU parity tests, whole path: the HIP-backed model against the CPU oracle on the same
formula weights and inputs, and against the committed reference golden vectors.

fp32 mode: tolerance 1e-5 on logits (BASELINE.json north_star), tight relative tolerances on
gradients.  bf16 mode: the small configs (micro / mini / tiny) are held to 1e-3 against the fp32 oracle.  At Base /
Large size the rounding of the bf16 GEMM operands themselves moves the logits by 2.7e-3 (measured on the ORACLE:
tools/bf16_budget.py, 98 % of it from rounding the weight operands), so there the HIP path is held to 1e-3 against
the oracle that rounds the same operands to bf16 (`lowp=O.bf16_round`), and its distance to the fp32 oracle must not
exceed that of the emulating oracle by more than 1e-3 (documented deviation)."""
import glob
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from nvit_amd.config import named_config
from nvit_amd.weights import formula_state_dict, synthetic_batch
from oracle import nvit_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def build(cfg, precision, renormed):
    from nvit_amd.model import ViT
    from nvit_amd.train import normalize_matrices
    m = ViT(cfg)
    res = m.load_state_dict(formula_state_dict(cfg), strict=False)   # Kohonen index buffers are not in the formula dict
    assert not res.unexpected_keys and all(k.endswith((".locations", ".offsets")) for k in res.missing_keys)
    m = m.to("cuda:0").set_precision(precision)
    if renormed:
        normalize_matrices(m)
    return m


def oracle_run(cfg, X, y, renormed, lowp=None):
    p = O.make_params(formula_state_dict(cfg))
    if renormed:
        O.renorm_(p, cfg)
    logits, loss, recon = O.loss_and_grads(p, cfg, X, y, lowp)
    return p, logits, loss, recon


CASES = [("micro", 8), ("mini", 4), ("tiny", 32)]


@pytest.mark.parametrize("name,batch", CASES)
@pytest.mark.parametrize("renormed", [False, True])
def test_fp32_forward_backward_vs_oracle(name, batch, renormed):
    torch.set_num_threads(8)
    cfg = named_config(name)
    X, y = synthetic_batch(cfg, batch)
    p, logits_ref, loss_ref, recon_ref = oracle_run(cfg, X, y, renormed)
    m = build(cfg, "fp32", renormed).train()
    logits, aux = m(X.cuda())
    loss = torch.nn.functional.cross_entropy(logits, y.cuda())
    loss.backward()
    err = (logits.detach().cpu() - logits_ref).abs().max().item()
    print(f"[fp32 {name} renorm={renormed}] max|dlogit|={err:.3e} loss {loss.item():.6f} vs {loss_ref.item():.6f}")
    assert err < 1e-5
    assert abs(loss.item() - loss_ref.item()) < 1e-5
    assert abs(aux["reconstruction"].item() - recon_ref.item()) < 1e-5
    have = {n for n, q in m.named_parameters() if q.grad is not None}
    want = {n for n, t in p.items() if t.grad is not None}
    assert have == want
    worst = 0.0
    for n, q in m.named_parameters():
        if q.grad is None:
            continue
        ref = p[n].grad
        e = (q.grad.cpu() - ref).abs().max().item()
        s = ref.abs().max().item()
        worst = max(worst, e / (s + 1e-12))
        assert e <= 2e-4 * s + 1e-8, (n, e, s)
    print(f"   worst relative grad error {worst:.3e}")


@pytest.mark.parametrize("name,batch", CASES)
def test_fp32_matches_reference_golden_and_one_step(name, batch):
    """Directly against numbers recorded from the real reference (tests/golden), incl. one full
    train step (clip + AdamW + renorm) and the step-1 logits."""
    from nvit_amd.train import train_step
    g = np.load(os.path.join(GOLD, f"{name}_b{batch}_init.npz"))
    cfg = named_config(name)
    X, y = synthetic_batch(cfg, batch)
    m = build(cfg, "fp32", False).train()
    opt = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    logits, loss, aux, gnorm = train_step(m, opt, X.cuda(), y.cuda(), 1.0)
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    assert abs(gnorm.item() - float(g["gnorm"])) < 2e-4 * float(g["gnorm"])
    with torch.no_grad():
        logits1, aux1 = m(X.cuda())
    e1 = np.abs(logits1.cpu().numpy() - g["logits1"]).max()
    print(f"[golden {name}] step-1 max|dlogit| = {e1:.3e}")
    assert e1 < 1e-4
    q0 = m.transformer.h[0].query.weight.detach().reshape(-1)[:8].cpu().numpy()
    assert np.abs(q0 - g["q0_head1"]).max() < 2e-6
    for blk in m.transformer.h:
        for lin, dim in ((blk.query, 1), (blk.key, 1), (blk.value, 1), (blk.c_fc, 1), (blk.att_c_proj, 0),
                         (blk.mlp_c_proj, 0)):
            assert (lin.weight.detach().norm(dim=dim) - 1).abs().max().item() < 1e-6


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_training_learns_a_separable_task(precision):
    """End to end as a learner, not only as a parity target: `mini` (T=49, C=128, 16 classes) trained with the fused
    step on images whose class is a fixed random pattern under noise (SNR ~ 1 per pixel) must drive the loss down
    and classify fresh samples of the same distribution."""
    from nvit_amd.train import train_step
    cfg = named_config("mini")
    g = torch.Generator().manual_seed(7)
    S, ncls, B = cfg.image_size, cfg.num_classes, 64
    protos = torch.randn(ncls, cfg.channels, S, S, generator=g) * 0.5

    def batch():
        y = torch.randint(0, ncls, (B,), generator=g)
        X = protos[y] + 0.5 * torch.randn(B, cfg.channels, S, S, generator=g)
        return X.cuda(), y.cuda()

    m = build(cfg, precision, True).train()
    opt = m.configure_optimizers(0.0, 3e-3, (0.9, 0.95), "cuda")
    first = last = None
    for it in range(120):
        X, y = batch()
        _, loss, _, _ = train_step(m, opt, X, y, 1.0)
        if it == 0:
            first = loss.item()
    last = loss.item()
    m.eval()
    X, y = batch()
    with torch.no_grad():
        logits, _ = m(X)
    acc = (logits.argmax(dim=-1) == y).float().mean().item()
    print(f"[learn {precision}] loss {first:.3f} -> {last:.3f}, held-out accuracy {acc:.3f}")
    assert math.isfinite(last) and last < 0.5 * first, (first, last)
    assert acc > 0.9, acc


@pytest.mark.parametrize("name,batch,steps", [("micro", 8, 6), ("mini", 4, 5)])
def test_training_trajectory_vs_oracle(name, batch, steps):
    """Several consecutive train steps (forward, CE, backward, clip, AdamW, renorm: train.py:898-946,989-990) with a new
    batch each step: the HIP path (fused optimizer) in fp32 mode must follow the CPU oracle's own training loop - logits
    and loss of every step, and every parameter at the end; the bf16 mode must follow the same loss curve."""
    from nvit_amd.train import train_step
    torch.set_num_threads(8)
    cfg = named_config(name)
    lr, wd = 3e-3, 0.1      # (a larger step than the default 1e-3: the trajectory has to move for the test to mean anything)
    p = O.make_params(formula_state_dict(cfg))
    o_opt = O.make_optimizer(p, lr=lr, weight_decay=wd)
    m32 = build(cfg, "fp32", False).train()
    mbf = build(cfg, "bf16", False).train()
    opt32 = m32.configure_optimizers(wd, lr, (0.9, 0.95), "cuda")
    optbf = mbf.configure_optimizers(wd, lr, (0.9, 0.95), "cuda")
    worst32 = worstbf = 0.0
    losses = []
    for it in range(steps):
        X, y = synthetic_batch(cfg, batch, seed=100 + it)
        lo, loss_o, _, gn_o = O.train_step(p, cfg, o_opt, X, y, 1.0)
        l32, loss32, _, gn32 = train_step(m32, opt32, X.cuda(), y.cuda(), 1.0)
        lbf, lossbf, _, gnbf = train_step(mbf, optbf, X.cuda(), y.cuda(), 1.0)
        e32 = (l32.detach().cpu() - lo.detach()).abs().max().item()
        ebf = (lbf.detach().float().cpu() - lo.detach()).abs().max().item()
        worst32, worstbf = max(worst32, e32), max(worstbf, ebf)
        losses.append((loss_o.item(), loss32.item(), lossbf.item()))
        assert abs(gn32.item() - gn_o.item()) < 1e-3 * gn_o.item(), (it, gn32.item(), gn_o.item())
    print(f"[trajectory {name}] {steps} steps: max|dlogit| fp32 {worst32:.3e}, bf16 {worstbf:.3e}; losses (oracle, fp32, bf16) "
          + " ".join(f"({a:.4f} {b:.4f} {c:.4f})" for a, b, c in losses))
    assert losses[-1][0] != losses[0][0]
    assert worst32 < 2e-4, worst32                       # fp32: rounding-order differences compounding over the steps
    # bf16: Adam's first steps move every coordinate by ~lr whatever the size of its gradient, so coordinates whose
    # gradient is at the bf16 rounding level take the other sign and the parameters (hence single logits) drift apart at
    # the lr scale; what has to hold is the loss curve (and logits that stay in the same place to a few 1e-2)
    assert worstbf < 5e-2, worstbf
    for a, b, c in losses:
        assert abs(a - b) < 1e-4 and abs(a - c) < 3e-3
    perr = 0.0
    for n, q in m32.named_parameters():
        perr = max(perr, (q.detach().cpu() - p[n].detach()).abs().max().item())
    assert perr < 2e-4, perr


@pytest.mark.parametrize("name,batch", CASES)
def test_bf16_forward_backward_vs_oracle(name, batch):
    cfg = named_config(name)
    X, y = synthetic_batch(cfg, batch)
    p, logits_ref, loss_ref, _ = oracle_run(cfg, X, y, True)
    pe, logits_emu, _, _ = oracle_run(cfg, X, y, True, lowp=O.bf16_round)
    m = build(cfg, "bf16", True).train()
    logits, aux = m(X.cuda())
    loss = torch.nn.functional.cross_entropy(logits, y.cuda())
    loss.backward()
    err = (logits.detach().cpu() - logits_ref).abs().max().item()
    err_emu = (logits.detach().cpu() - logits_emu).abs().max().item()
    lmax = logits_ref.abs().max().item()
    print(f"[bf16 {name}] max|dlogit| vs fp32 oracle {err:.3e} (|logit|max {lmax:.3f}, rel {err / lmax:.3e}); "
          f"vs bf16-operand oracle {err_emu:.3e}")
    # The bf16 bar.  Rounding the GEMM operands to bf16 moves the logits of the ORACLE ITSELF by d_emu (micro 2e-4,
    # mini 4e-4, tiny 1.1e-3, Base 2.7e-3, Large 4.0e-3: tools/bf16_budget.py), so "within 1e-3 of the fp32 oracle"
    # is not a property a bf16-operand computation can have beyond the small configs.  Held instead: within 1e-3 of
    # the oracle that rounds the same operands, and no further from the fp32 oracle than that emulation plus 5e-4.
    d_emu = (logits_emu - logits_ref).abs().max().item()
    assert err_emu < 1e-3, (err_emu, d_emu)
    assert err < d_emu + 5e-4, (err, d_emu)
    # gradients: cosine similarity per parameter against the fp32 oracle
    worst = 1.0
    for n, q in m.named_parameters():
        if q.grad is None:
            continue
        a, b = q.grad.cpu().flatten().double(), p[n].grad.flatten().double()
        if b.norm() < 1e-12:
            continue
        cos = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
        worst = min(worst, cos)
        assert cos > 0.98, (n, cos)
        # scalar parameters (skip_param) have cancellation-dominated gradients of ~1e-4: looser bound
        rtol = 0.1 if a.numel() > 16 else 0.25
        assert abs(a.norm() / b.norm() - 1) < rtol, (n, a.norm().item(), b.norm().item())
    print(f"   worst grad cosine {worst:.5f}")


def test_state_dict_keys_and_no_cpu_fallback():
    from nvit_amd.model import ViT
    cfg = named_config("micro")
    m = ViT(cfg)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32))


def test_bf16_fused_epilogues_match_unfused_and_oracle():
    """A config large enough to take the fused-epilogue persistent GEMMs (q/k normalise, SwiGLU), the
    persistent 256x256 NT/TN kernels and the MFMA attention: against the fp32 oracle and against the
    same model with the fusions switched off."""
    from nvit_amd import ops
    cfg = named_config("mini", n_embd=256, n_head=4, num_classes=16)
    batch = 112
    X, y = synthetic_batch(cfg, batch)
    p, logits_ref, loss_ref, _ = oracle_run(cfg, X, y, True)

    def run(fuse_min):
        old = ops.FUSE_MIN_ELEMS
        ops.FUSE_MIN_ELEMS = fuse_min
        try:
            m = build(cfg, "bf16", True).train()
            logits, aux = m(X.cuda())
            torch.nn.functional.cross_entropy(logits, y.cuda()).backward()
            return logits.detach().cpu(), {n: q.grad.cpu() for n, q in m.named_parameters() if q.grad is not None}
        finally:
            ops.FUSE_MIN_ELEMS = old

    T = (cfg.image_size // cfg.local_patch_size) ** 2
    assert ops.fusable(1, batch * T, 3 * cfg.n_embd, cfg.n_embd), "test config no longer reaches the fused path"
    lf, gf = run(ops.FUSE_MIN_ELEMS)
    lu, gu = run(1 << 62)
    e_or = (lf - logits_ref).abs().max().item()
    e_fu = (lf - lu).abs().max().item()
    print(f"[fused] max|dlogit| vs fp32 oracle {e_or:.3e}, fused vs unfused {e_fu:.3e}")
    assert e_or < 1.5e-3 and e_fu < 1e-3
    for n in gf:
        a, b, r = gf[n].flatten().double(), gu[n].flatten().double(), p[n].grad.flatten().double()
        if r.norm() < 1e-12:
            continue
        cos_u = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
        cos_r = (a @ r / (a.norm() * r.norm() + 1e-30)).item()
        assert cos_u > 0.995 and cos_r > 0.98, (n, cos_u, cos_r)


@pytest.mark.parametrize("name,batch", [("micro_k", 8), ("mini_k", 4)])
def test_kohonen_head_fp32_vs_oracle_and_golden(name, batch):
    """BASELINE config C5 semantics (Kohonen head on) at parity size: aux losses, SOM node update, logits,
    every gradient (incl. the SOM nodes) against the oracle; logits/aux/loss against the reference goldens."""
    from nvit_amd.train import total_loss
    cfg = named_config(name)
    X, y = synthetic_batch(cfg, batch)
    p = O.make_params(formula_state_dict(cfg))
    logits_ref, loss_ref, aux_ref = O.loss_and_grads(p, cfg, X, y, step=1, want_aux=True)
    m = build(cfg, "fp32", False).train()
    logits, aux = m(X.cuda())
    loss = total_loss(cfg, logits, aux, y.cuda())
    loss.backward()
    g = np.load(os.path.join(GOLD, f"{name}_b{batch}_init.npz"))
    assert np.abs(logits.detach().cpu().numpy() - g["logits"]).max() < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 2e-5 * float(g["loss"])
    for i, k in enumerate(("kohonen_consistency", "kohonen_smoothness", "local_quantization", "global_quantization")):
        assert abs(aux[k].item() - float(g["aux"][i])) < 2e-5 * max(1.0, abs(float(g["aux"][i]))), k
        assert abs(aux[k].item() - aux_ref[k].item()) < 2e-5 * max(1.0, abs(aux_ref[k].item())), k
    assert abs(aux["reconstruction"].item() - float(g["recon"])) < 2e-5
    # SOM nodes after the in-forward update
    for mod, key in ((m.local_kohonen, "lnodes_head"), (m.global_kohonen, "gnodes_head")):
        assert np.abs(mod.nodes.detach().reshape(-1)[:8].cpu().numpy() - g[key]).max() < 2e-6
    assert (m.local_kohonen.nodes.detach().cpu() - p["local_kohonen.nodes"].detach()).abs().max().item() < 2e-6
    have = {n for n, q in m.named_parameters() if q.grad is not None}
    want = {n for n, t in p.items() if t.grad is not None}
    assert have == want, have ^ want
    for n, q in m.named_parameters():
        if q.grad is None:
            continue
        ref = p[n].grad
        e, s = (q.grad.cpu() - ref).abs().max().item(), ref.abs().max().item()
        assert e <= 3e-4 * s + 1e-8, (n, e, s)


def test_kohonen_head_bf16_runs_and_tracks_oracle():
    """bf16 operands with the Kohonen head: the SOM node vectors are N(0,1)-sized (|repr| ~ sqrt(C)), so bf16
    operand rounding is amplified through the three cross-attention calls; the meaningful comparison is against
    the oracle that rounds the same GEMM operands to bf16.  Both distances are printed."""
    cfg = named_config("mini_k")
    X, y = synthetic_batch(cfg, 4)
    p = O.make_params(formula_state_dict(cfg))
    logits_ref, loss_ref, aux_ref = O.loss_and_grads(p, cfg, X, y, step=1, want_aux=True)
    pe = O.make_params(formula_state_dict(cfg))
    logits_emu, loss_emu, _ = O.loss_and_grads(pe, cfg, X, y, lowp=O.bf16_round, step=1, want_aux=True)
    from nvit_amd.train import train_step
    m = build(cfg, "bf16", False).train()
    opt = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    logits, loss, aux, gnorm = train_step(m, opt, X.cuda(), y.cuda(), 1.0)
    e32 = (logits.cpu() - logits_ref).abs().max().item()
    eemu = (logits.cpu() - logits_emu).abs().max().item()
    print(f"[bf16 mini_k] max|dlogit| vs fp32 oracle {e32:.3e}, vs bf16-operand oracle {eemu:.3e}, "
          f"oracle fp32 vs bf16-operand {(logits_ref - logits_emu).abs().max().item():.3e}")
    assert eemu < 1e-3 and e32 < 2e-3
    assert abs(loss.item() - loss_ref.item()) < 2e-2 * abs(loss_ref.item())
    assert torch.isfinite(gnorm).item()


def test_block_forward_and_norm_skip_standalone_api():
    """Reference call pattern of model.py:450-452 through the PUBLIC methods: patches_new = block(patches);
    patches = block.norm_skip(patches_new, patches) — must equal the fused path used by ViT.forward."""
    cfg = named_config("mini")
    m = build(cfg, "fp32", True).train()
    B, T, C = 3, m.n_tokens, cfg.n_embd
    x = torch.nn.functional.normalize(torch.randn(B, T, C, generator=torch.Generator().manual_seed(5)), dim=-1)
    p = O.make_params(formula_state_dict(cfg))
    O.renorm_(p, cfg)
    xr = x.clone().requires_grad_(True)
    ref = O.block(p, cfg, 0, xr, None)
    g = torch.randn(B, T, C, generator=torch.Generator().manual_seed(6))
    ref.backward(g)
    blk = m.transformer.h[0]
    xg = x.cuda().requires_grad_(True)
    new = blk(xg)
    out = blk.norm_skip(new, xg)
    out.backward(g.cuda())
    assert (out.detach().cpu() - ref.detach()).abs().max().item() < 1e-6
    assert (xg.grad.cpu() - xr.grad).abs().max().item() < 2e-5 * max(1.0, xr.grad.abs().max().item())
    gs = blk.skip_param.grad.cpu()
    assert (gs - p["transformer.h.0.skip_param"].grad).abs().max().item() < 2e-4 * max(
        1.0, p["transformer.h.0.skip_param"].grad.abs().max().item())


def test_base_config_full_size_properties():
    """BASELINE config C2 (nViT-Base/16 224 px, T=784) at full model size, small batch: size-independent
    properties instead of an oracle run — (1) samples are independent: logits of a batch equal the logits of each
    sample alone; (2) three optimizer steps stay finite and reduce the loss on a fixed batch; (3) after every step
    the six matrices per block have unit rows / columns; (4) fp32 mode and bf16 mode agree to bf16 tolerance."""
    from nvit_amd.model import ViT
    from nvit_amd.train import normalize_matrices, train_step
    from nvit_amd.weights import load_formula_weights
    cfg = named_config("base")
    m = ViT(cfg)
    load_formula_weights(m, cfg, perturb_scalars=False)
    m = m.to("cuda:0").set_precision("bf16").train()
    normalize_matrices(m)
    X, y = synthetic_batch(cfg, 4)
    X, y = X.cuda(), y.cuda()
    with torch.no_grad():
        m.eval()
        full, _ = m(X)
        singles = torch.cat([m(X[i:i + 1])[0] for i in range(4)])
        m.set_precision("fp32")
        full32, _ = m(X[:2])
        m.set_precision("bf16").train()
    assert torch.isfinite(full).all()
    # (a batch of 4 and a batch of 1 take different kernel variants - persistent/fused vs 128-tile/unfused - so the
    #  comparison is at bf16 tolerance, not bitwise)
    bs = (full - singles).abs().max().item()
    print(f"[base full size] batch of 4 vs four single-image calls (different kernel variants): max|dlogit| = {bs:.3e}")
    assert bs < 1.5e-3, "batch rows are not independent"
    d32 = (full[:2] - full32).abs().max().item()
    print(f"[base full size] bf16 vs fp32 mode max|dlogit| = {d32:.3e} (|logit|max {full32.abs().max().item():.3f})")
    assert d32 < 4e-3      # = the operand-rounding cost measured against the oracle (2.7e-3 at Base) + margin
    opt = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    losses = []
    for _ in range(3):
        _, loss, _, gnorm = train_step(m, opt, X, y, 1.0)
        assert torch.isfinite(loss).item() and torch.isfinite(gnorm).item()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    for blk in m.transformer.h:
        for lin, dim in ((blk.query, 1), (blk.key, 1), (blk.value, 1), (blk.c_fc, 1), (blk.att_c_proj, 0),
                         (blk.mlp_c_proj, 0)):
            assert (lin.weight.detach().norm(dim=dim) - 1).abs().max().item() < 1e-5


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graphed_train_step_equals_eager(precision):
    """The hipGraph-captured step replays exactly what the eager step does: same kernels, same order,
    device-side AdamW step counter.  Two models from the same weights, same batches; weights must agree bit for bit."""
    from nvit_amd.train import GraphedTrainStep, train_step
    cfg = named_config("micro")
    X, y = synthetic_batch(cfg, 8)
    X, y = X.cuda(), y.cuda()
    X2, y2 = synthetic_batch(cfg, 8, seed=77)
    X2, y2 = X2.cuda(), y2.cuda()
    me = build(cfg, precision, True)
    mg = build(cfg, precision, True)
    oe = me.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    og = mg.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    warm = 2
    for _ in range(warm):
        train_step(me, oe, X, y)
    g = GraphedTrainStep(mg, og, X, y, warmup=warm)
    for (xb, yb) in ((X, y), (X2, y2), (X, y)):
        le, losse, _, gne = train_step(me, oe, xb, yb)
        lg, lossg, _, gng = g(xb, yb)
        assert torch.equal(le, lg), (le - lg).abs().max().item()
        assert torch.equal(losse, lossg)
        assert torch.equal(gne, gng)
    for (n, pe), (_, pg) in zip(me.named_parameters(), mg.named_parameters()):
        assert torch.equal(pe, pg), n
    assert oe.state_dict()["state"][0]["step"] == og.state_dict()["state"][0]["step"] == warm + 3
    # learning-rate change between replays reaches the captured step through the device table
    for grp in oe.param_groups:
        grp["lr"] = 5e-4
    g.set_lr(5e-4)
    train_step(me, oe, X2, y2)
    g(X2, y2)
    for (n, pe), (_, pg) in zip(me.named_parameters(), mg.named_parameters()):
        assert torch.equal(pe, pg), n


def _poison_free_memory(nbytes=6 << 30):
    """Fill a large block of free HBM with NaNs and release it, so later torch.empty() buffers start as NaN."""
    t = torch.full((nbytes // 4,), float("nan"), device="cuda")
    torch.cuda.synchronize()
    del t


@pytest.mark.parametrize("name,batch", [("micro", 8), ("tiny", 32)])
def test_no_uninitialised_reads_under_nan_poison(name, batch):
    """Every workspace/partial buffer is fully written before it is read: with freed memory poisoned by NaNs the eager
    step and the replayed hipGraph step give the same finite numbers as a clean run (this caught a zero block that a
    captured memset did not refresh)."""
    from nvit_amd.train import GraphedTrainStep, train_step
    cfg = named_config(name)
    X, y = synthetic_batch(cfg, batch)
    X, y = X.cuda(), y.cuda()
    clean = build(cfg, "bf16", True)
    oc = clean.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    ref = [train_step(clean, oc, X, y)[1].item() for _ in range(5)]
    m = build(cfg, "bf16", True)
    o = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    got = []
    for _ in range(2):
        _poison_free_memory()
        got.append(train_step(m, o, X, y)[1].item())
    _poison_free_memory()
    g = GraphedTrainStep(m, o, X, y, warmup=1)
    got.append(float("nan"))           # the warm-up step inside GraphedTrainStep (not returned)
    for _ in range(2):
        _poison_free_memory()
        got.append(g(X, y)[1].item())
    assert got[0] == ref[0] and got[1] == ref[1] and got[3] == ref[3] and got[4] == ref[4], (got, ref)
    for n, p in m.named_parameters():
        assert torch.isfinite(p).all(), n


def test_kohonen_step_under_nan_poison():
    """Same check for the Kohonen-head step (eager only: its SOM schedule is host state)."""
    from nvit_amd.train import train_step
    cfg = named_config("mini_k")
    X, y = synthetic_batch(cfg, 4)
    X, y = X.cuda(), y.cuda()
    clean = build(cfg, "bf16", True)
    oc = clean.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    ref = [train_step(clean, oc, X, y)[1].item() for _ in range(3)]
    m = build(cfg, "bf16", True)
    o = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    got = []
    for _ in range(3):
        _poison_free_memory()
        got.append(train_step(m, o, X, y)[1].item())
    assert got == ref, (got, ref)


def _record_margin(name, batch, vals, key=None):
    """Measured bf16 margins of the full-size tests, appended to gpurun_out/parity_margins.json on the GPU box (a copy is
    committed under profiles/ each round, so a drift of the ABSOLUTE errors stays visible even while the tests pass)."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_margins.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        entry = {k: float(f"{float(v):.4e}") for k, v in vals.items()}
        if key is None:
            data.setdefault(f"{name}_b{batch}", {}).update(entry)
        else:
            data.setdefault(f"{name}_b{batch}", {})[key] = entry
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def _oracle_with_taps(cfg, X, y, lowp):
    p = O.make_params(formula_state_dict(cfg))
    O.renorm_(p, cfg)
    taps = {}
    logits, aux = O.forward(p, cfg, X, lowp, taps=taps, training=True)
    loss = O.cross_entropy(logits, y)
    loss.backward()
    return p, logits.detach(), loss.detach(), {k: v.detach() for k, v in taps.items() if k.startswith("x")}


def _hip_with_taps(m, X, y):
    taps = {}
    object.__setattr__(m, "_taps", taps)
    try:
        logits, aux = m(X.cuda())
        loss = torch.nn.functional.cross_entropy(logits, y.cuda())
        loss.backward()
    finally:
        object.__setattr__(m, "_taps", None)
    B = X.shape[0]
    return logits.detach().cpu(), loss.item(), {k: v.cpu().reshape(B, -1, v.shape[-1]) for k, v in taps.items()
                                                if k.startswith("x")}


def _layer_errors(taps, ref):
    return [(taps[f"x{i}"] - ref[f"x{i}"]).abs().max().item() for i in range(len(ref))]


def _full_size_bf16_parity(name, batch, grads: bool):
    """Shared body of the Base / Large full-size tests.  Returns nothing; asserts and prints."""
    from nvit_amd import _lib
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    cfg = named_config(name)
    X, y = synthetic_batch(cfg, batch)
    p32, l32, loss32, t32 = _oracle_with_taps(cfg, X, y, None)
    # Four CPU evaluations of the same network with every GEMM operand rounded to bf16 (O.KernelRounding): the softmax
    # probabilities rounded relative to the score bound (the HIP kernels' convention) or to the row maximum (the
    # textbook one), each with float32 and with float64 accumulation.  They differ in WHERE the same kind of rounding
    # happens and in summation order only, and they scatter: their mutual max-distances (the `spread`) are 0.4-0.75e-3
    # at Base and 0.7-0.9e-3 at Large, rms 1.1-2.4e-4 (tools/parity_attribution.py prints the whole matrix).  The HIP
    # path is held to 1e-3 of the NEAREST of them and to their own rms scatter: it must be one more member of that
    # family, not an outlier.  (The matched-convention float32 one also supplies the per-layer taps and the gradients.)
    pem, lem, lossem, tem = _oracle_with_taps(cfg, X, y, O.KernelRounding("bound"))
    emus = {"bound/f32": lem}
    with torch.no_grad():
        pf = O.make_params(formula_state_dict(cfg))
        O.renorm_(pf, cfg)
        for tag, lp in (("bound/f64", O.KernelRounding("bound", acc64=True)), ("rowmax/f32", O.KernelRounding("rowmax")),
                        ("rowmax/f64", O.KernelRounding("rowmax", acc64=True))):
            emus[tag] = O.forward(pf, cfg, X, lp, training=True)[0]
        del pf
    rms = lambda t: t.double().pow(2).mean().sqrt().item()
    tags = list(emus)
    spread = max((emus[a] - emus[b]).abs().max().item() for i, a in enumerate(tags) for b in tags[i + 1:])
    spread_rms = max(rms(emus[a] - emus[b]) for i, a in enumerate(tags) for b in tags[i + 1:])
    lmax = l32.abs().max().item()
    d_emu = (lem - l32).abs().max().item()
    floor = (lem - emus["bound/f64"]).abs().max().item()
    print(f"[{name} B={batch}] oracle: |logit|max {lmax:.3f}; bf16-operand oracle vs fp32 oracle {d_emu:.3e} "
          f"(the intrinsic cost of bf16 MFMA operands at this size); the four bf16-operand CPU evaluations scatter by up "
          f"to {spread:.3e} max / {spread_rms:.3e} rms among themselves (summation order alone: {floor:.3e})")
    # ---- fp32 mode: the 1e-5 bar, gradient norms to 1e-3
    m = build(cfg, "fp32", True).train()
    lg, loss, taps = _hip_with_taps(m, X, y)
    e = (lg - l32).abs().max().item()
    print(f"   fp32 mode : max|dlogit| {e:.3e}, loss {loss:.6f} vs {loss32:.6f}, residual stream per layer "
          f"{max(_layer_errors(taps, t32)):.2e}")
    assert e < 1e-5
    worst = 0.0
    for n, q in m.named_parameters():
        if q.grad is None or p32[n].grad is None:
            continue
        a, b = q.grad.cpu().double().norm().item(), p32[n].grad.double().norm().item()
        if b > 1e-9:
            worst = max(worst, abs(a / b - 1))
    print(f"   fp32 mode : worst relative gradient-norm error {worst:.3e}")
    assert worst < 1e-3
    m.zero_grad(set_to_none=True)
    # ---- bf16 mode, default kernel dispatch, then with the persistent kernels forced (the B=128 code path)
    m.set_precision("bf16")
    results = {}
    for tag, nt_impl in (("default dispatch", -1), ("persistent kernels forced", 2)):
        _lib.load().nvit_set_gemm_impl(nt_impl if nt_impl >= 0 else 1, 1)
        try:
            m.zero_grad(set_to_none=True)
            lb, lossb, tb = _hip_with_taps(m, X, y)
        finally:
            _lib.load().nvit_set_gemm_impl(1, 1)
        e32 = (lb - l32).abs().max().item()
        dist = {t: (lb - e).abs().max().item() for t, e in emus.items()}
        eem = min(dist.values())
        print("      HIP vs the four emulations, max|dlogit|: " + ", ".join(f"{t} {v:.3e}" for t, v in dist.items()))
        le32, leem = _layer_errors(tb, t32), _layer_errors(tb, tem)
        lo = _layer_errors(tem, t32)
        print(f"   bf16 mode ({tag}): max|dlogit| vs the nearest bf16-operand oracle {eem:.3e}, vs fp32 oracle {e32:.3e} "
              f"(rel {e32 / lmax:.2e}); rms: HIP-emu {rms(lb - lem):.2e}, HIP-fp32 {rms(lb - l32):.2e}, "
              f"emu-fp32 {rms(lem - l32):.2e}")
        print("      residual stream rms per layer  HIP-vs-emu : " + " ".join(
            f"{rms(tb[f'x{i}'] - tem[f'x{i}']):.1e}" for i in range(len(tem))))
        print("                                     emu-vs-fp32: " + " ".join(
            f"{rms(tem[f'x{i}'] - t32[f'x{i}']):.1e}" for i in range(len(tem))))
        print("      residual stream max|dx| per layer  HIP-vs-emu : " + " ".join(f"{v:.1e}" for v in leem))
        print("                                         HIP-vs-fp32: " + " ".join(f"{v:.1e}" for v in le32))
        print("                                         emu-vs-fp32: " + " ".join(f"{v:.1e}" for v in lo))
        results[tag] = (lb, e32, eem)
        # the bar: within 1e-3 of a CPU evaluation that rounds the same operands, and inside the family's own rms scatter
        assert eem < 1e-3, (tag, dist, spread)
        # ... and, so that a drift of the matched convention cannot hide behind another member of the family: the single
        # emulation that rounds where the kernels round ("bound", float32 accumulation) within 1.2e-3 (measured: Base
        # 6.4e-4, Large 9.1e-4)
        assert dist["bound/f32"] < 1.2e-3, (tag, dist)
        assert max(rms(lb - e) for e in emus.values()) < 1.25 * spread_rms + 1e-5, (tag, spread_rms)
        # ... and no further from the fp32 oracle than that emulation is: the whole logit field in rms (+5 %), and its
        # maximum (one of ~2 000 values of two superposed error fields; it moves by up to 13 % with nothing but the
        # summation order of the kernels, measured over the round's kernel variants) within 15 % + 2e-4
        assert rms(lb - l32) < 1.05 * rms(lem - l32) + 1e-5, (tag, rms(lb - l32), rms(lem - l32))
        assert e32 < 1.15 * d_emu + 2e-4, (tag, e32, d_emu)
        # no kernel term that grows with depth beyond what operand rounding explains: per layer, the HIP stream is
        # at most as far from the emulation as the emulation is from fp32 (x2 slack), elementwise max over [B,T,C]
        for i, (a, b) in enumerate(zip(leem, lo)):
            assert a < 3.0 * b + 1e-4, (tag, i, a, b)
        if grads:
            worst_cos, worst_ratio = 1.0, 0.0
            for n, q in m.named_parameters():
                if q.grad is None or p32[n].grad is None:
                    continue
                a, b = q.grad.cpu().flatten().double(), p32[n].grad.flatten().double()
                if b.norm() < 1e-12:
                    continue
                cos = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
                ratio = abs((a.norm() / b.norm()).item() - 1)
                if a.numel() > 16:          # scalars (skip_param): cancellation-dominated, checked by ratio only
                    worst_cos = min(worst_cos, cos)
                    assert cos > 0.999, (tag, n, cos)
                assert ratio < (0.02 if a.numel() > 16 else 0.25), (tag, n, ratio)
                worst_ratio = max(worst_ratio, ratio if a.numel() > 16 else 0.0)
            print(f"      gradients vs fp32 oracle: worst cosine {worst_cos:.6f}, worst norm ratio error {worst_ratio:.2e}")
    _record_margin(name, batch, dict(hip_vs_emulation=results["default dispatch"][2], hip_vs_fp32=results["default dispatch"][1],
                                     emulation_vs_fp32=d_emu, summation_floor=floor, emulation_spread=spread, logit_max=lmax))
    a, b = results["default dispatch"][0], results["persistent kernels forced"][0]
    print(f"   bf16 mode: default dispatch vs persistent kernels {(a - b).abs().max().item():.3e}")
    assert (a - b).abs().max().item() < 1e-3
    return m


FULL_GOLD = [("base", 6), ("large", 2), ("base_k", 2)]


@pytest.mark.parametrize("name,batch", FULL_GOLD)
def test_fp32_full_size_matches_reference_golden(name, batch):
    """The BASELINE model sizes (C2 Base B=6, C4 Large B=2, C5 Base+Kohonen B=2) DIRECTLY against numbers recorded from
    the imported reference (oracle/make_golden.py full -> tests/golden/<name>_b<B>_renorm.npz; reference
    model.py:403-470, train.py:898-946,989-990): the CPU oracle is not in this chain.  fp32 mode: logits 1e-5 (2e-5
    with the Kohonen head), loss, aux losses, per-parameter gradient norms and leading slices, the clipped global norm,
    and after one full step (clip + AdamW + renorm) the step-1 logits and leading weights."""
    from nvit_amd.train import total_loss
    g = np.load(os.path.join(GOLD, f"{name}_b{batch}_renorm.npz"))
    cfg = named_config(name)
    X, y = synthetic_batch(cfg, batch)
    m = build(cfg, "fp32", True).train()
    opt = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    logits, aux = m(X.cuda())
    loss = total_loss(cfg, logits, aux, y.cuda())
    loss.backward()
    tol = 2e-5 if cfg.use_kohonen else 1e-5
    e = np.abs(logits.detach().cpu().numpy() - g["logits"]).max()
    print(f"[golden {name} B={batch}] fp32 mode vs the reference: max|dlogit| {e:.3e} (|logit|max {np.abs(g['logits']).max():.3f}), "
          f"loss {loss.item():.6f} vs {float(g['loss']):.6f}")
    assert e < tol
    assert abs(loss.item() - float(g["loss"])) < 2e-5 * max(1.0, float(g["loss"]))
    assert abs(aux["reconstruction"].item() - float(g["recon"])) < 2e-5
    if cfg.use_kohonen:
        got = np.array([aux[k].item() for k in ("kohonen_consistency", "kohonen_smoothness", "local_quantization",
                                                 "global_quantization")])
        assert np.abs(got - g["aux"]).max() < 2e-5 * max(1.0, np.abs(g["aux"]).max())
        assert np.abs(m.local_kohonen.nodes.detach().reshape(-1)[:8].cpu().numpy() - g["lnodes_head"]).max() < 5e-6
        assert np.abs(m.global_kohonen.nodes.detach().reshape(-1)[:8].cpu().numpy() - g["gnodes_head"]).max() < 5e-6
    names = [str(n) for n in g["grad_names"]]
    params = dict(m.named_parameters())
    assert {n for n, q in params.items() if q.grad is not None} == set(names)
    worst = 0.0
    for n, gn, gh in zip(names, g["grad_norms"], g["grad_heads"]):
        grad = params[n].grad
        mine = grad.double().norm().item()
        worst = max(worst, abs(mine / gn - 1) if gn > 1e-9 else 0.0)
        assert abs(mine - gn) <= 5e-4 * gn + 1e-7, (n, mine, gn)
        head = grad.reshape(-1)[:8].cpu().numpy() if grad.numel() >= 8 else np.resize(grad.reshape(-1).cpu().numpy(), 8)
        assert np.abs(head - gh).max() <= 1e-3 * max(np.abs(gh).max(), 1e-30) + 1e-6 * gn, n
    print(f"   worst relative gradient-norm error vs the reference {worst:.3e}")
    gnorm = opt.step_fused(m, 1.0)[0].item()
    opt.zero_grad(set_to_none=True)
    assert abs(gnorm - float(g["gnorm"])) < 2e-4 * float(g["gnorm"])
    with torch.no_grad():
        logits1, aux1 = m(X.cuda())
    e1 = np.abs(logits1.cpu().numpy() - g["logits1"]).max()
    print(f"   step-1 max|dlogit| {e1:.3e}")
    assert e1 < 2e-4
    q0 = m.transformer.h[0].query.weight.detach().reshape(-1)[:8].cpu().numpy()
    assert np.abs(q0 - g["q0_head1"]).max() < 2e-6
    pl = m.transformer.h[-1].mlp_c_proj.weight.detach().reshape(-1)[:8].cpu().numpy()
    assert np.abs(pl - g["p_last_head1"]).max() < 2e-6
    _record_margin(name, batch, dict(fp32_mode_vs_reference_fp32=e, fp32_mode_step1_vs_reference=e1), key="reference_fp32")


AUTOCAST_GOLD = [("tiny", 32), ("mini", 4), ("base", 6), ("large", 2), ("base_k", 2)]


@pytest.mark.parametrize("name,batch", AUTOCAST_GOLD)
def test_bf16_deviation_bounded_by_the_references_own_bf16_path(name, batch):
    """The primary bf16 bar, on reference-held data only (tests/golden/<name>_b<B>_autocast.npz, recorded by
    oracle/make_golden.py autocast from the imported reference: its fp32 logits and its logits under
    `torch.autocast("cpu", dtype=torch.bfloat16)`, the context train.py:254 builds and train.py:905 runs the model in):
    the HIP bf16 mode must be no farther from the reference's fp32 logits than the reference's own bf16 path is,
    in max and in rms.  All three numbers are recorded in gpurun_out/parity_margins.json (committed per round)."""
    g = np.load(os.path.join(GOLD, f"{name}_b{batch}_autocast.npz"))
    cfg = named_config(name)
    X, _ = synthetic_batch(cfg, batch)
    ref32, refbf = g["logits_fp32"], g["logits_autocast_bf16"]
    m = build(cfg, "bf16", True).train()
    with torch.no_grad():
        lb, _ = m(X.cuda())
    lb = lb.float().cpu().numpy()
    rms = lambda a: float(np.sqrt(np.mean(np.square(a.astype(np.float64)))))
    hip_dev, ref_dev = np.abs(lb - ref32).max(), np.abs(refbf - ref32).max()
    hip_rms, ref_rms = rms(lb - ref32), rms(refbf - ref32)
    print(f"[autocast {name} B={batch}] |HIP_bf16 - ref_fp32| max {hip_dev:.3e} rms {hip_rms:.3e};  |ref_autocast_bf16 - ref_fp32| "
          f"max {ref_dev:.3e} rms {ref_rms:.3e};  |HIP_bf16 - ref_autocast_bf16| max {np.abs(lb - refbf).max():.3e};  "
          f"|logit|max {np.abs(ref32).max():.3f}")
    _record_margin(name, batch, dict(hip_bf16_vs_reference_fp32=hip_dev, reference_autocast_bf16_vs_reference_fp32=ref_dev,
                                     hip_bf16_vs_reference_autocast_bf16=np.abs(lb - refbf).max(),
                                     hip_bf16_vs_reference_fp32_rms=hip_rms, reference_autocast_vs_fp32_rms=ref_rms,
                                     logit_max=np.abs(ref32).max()), key="reference_autocast")
    assert abs(float(g["max_abs_dev"]) - ref_dev) < 1e-9
    assert hip_dev <= ref_dev, (hip_dev, ref_dev)
    assert hip_rms <= ref_rms, (hip_rms, ref_rms)


def test_base_config_full_size_vs_cpu_oracle():
    """BASELINE config C2 at full model size against the CPU oracle, B=6 (M = 4704: the q/k-norm, SwiGLU and
    SwiGLU-backward GEMM epilogues and the persistent weight-gradient kernel are all on the path; a second pass forces
    the persistent NT kernels that B=128 uses).  fp32 mode: 1e-5 on logits, 1e-3 on gradient norms.  bf16 mode: 1e-3
    against the bf16-operand oracle, gradients cosine >= 0.999 against the fp32 oracle, per-layer residual-stream
    errors printed and bounded."""
    _full_size_bf16_parity("base", 6, grads=True)


def test_checkpoint_resume_is_exact(tmp_path):
    """Reference checkpoint dict (train.py:640-650) through nvit_amd.checkpoint: 2 steps, save, load into a fresh
    model + FusedAdamW, 1 more step == 3 uninterrupted steps, bit for bit (weights, AdamW moments, step counter)."""
    from nvit_amd.checkpoint import load_checkpoint, save_checkpoint
    from nvit_amd.train import train_step
    cfg = named_config("micro")
    X, y = synthetic_batch(cfg, 8)
    X, y = X.cuda(), y.cuda()
    mk_opt = lambda mm: mm.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    a = build(cfg, "bf16", True)
    oa = mk_opt(a)
    for _ in range(3):
        train_step(a, oa, X, y)
    b = build(cfg, "bf16", True)
    ob = mk_opt(b)
    for _ in range(2):
        train_step(b, ob, X, y)
    path = save_checkpoint(tmp_path / "checkpoint_latest.pt", b, ob, 2, {"val/loss": 0.0, "train/loss": 0.0})
    c, oc, ck = load_checkpoint(path, device="cuda", optimizer_factory=mk_opt, trusted=True)
    c.set_precision("bf16").train()
    assert ck["iter_num"] == 2
    train_step(c, oc, X, y)
    for (n, pa), (_, pc) in zip(a.named_parameters(), c.named_parameters()):
        assert torch.equal(pa, pc), n
    sa, sc = oa.state_dict()["state"], oc.state_dict()["state"]
    assert sa.keys() == sc.keys()
    for k in sa:
        assert float(sa[k]["step"]) == float(sc[k]["step"]) == 3.0
        assert torch.equal(sa[k]["exp_avg"], sc[k]["exp_avg"]) and torch.equal(sa[k]["exp_avg_sq"], sc[k]["exp_avg_sq"])


def test_large_config_full_size_vs_cpu_oracle():
    """BASELINE config C4 (nViT-Large/16: C=1024, H=16, L=24) at full model size, B=2, same bars as the Base test
    (forward + per-layer stream + gradients); then one fused optimizer step keeps rows/columns unit."""
    from nvit_amd.train import train_step
    m = _full_size_bf16_parity("large", 2, grads=True)
    cfg = named_config("large")
    X, y = synthetic_batch(cfg, 2)
    m.zero_grad(set_to_none=True)
    opt = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    _, loss, _, gnorm = train_step(m, opt, X.cuda(), y.cuda(), 1.0)
    assert torch.isfinite(loss).item() and torch.isfinite(gnorm).item()
    blk = m.transformer.h[-1]
    assert (blk.c_fc.weight.detach().norm(dim=1) - 1).abs().max().item() < 1e-5
    assert (blk.mlp_c_proj.weight.detach().norm(dim=0) - 1).abs().max().item() < 1e-5


def test_base_kohonen_config_c5_vs_cpu_oracle():
    """BASELINE config C5 (nViT-Base + Kohonen head, 2 maps of 16x16 nodes, C=768, T=784) at full model size, B=2:
    fp32 mode against the oracle - logits, the four aux losses, the reconstruction loss, the SOM nodes after the
    in-forward update, every gradient incl. the node gradients; bf16 mode against both oracles."""
    from nvit_amd.train import total_loss
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    cfg = named_config("base_k")
    X, y = synthetic_batch(cfg, 2)
    p = O.make_params(formula_state_dict(cfg))
    O.renorm_(p, cfg)
    l32, loss32, aux32 = O.loss_and_grads(p, cfg, X, y, step=1, want_aux=True)
    m = build(cfg, "fp32", True).train()
    logits, aux = m(X.cuda())
    loss = total_loss(cfg, logits, aux, y.cuda())
    loss.backward()
    e = (logits.detach().cpu() - l32).abs().max().item()
    print(f"[base_k B=2] fp32 mode: max|dlogit| {e:.3e} (|logit|max {l32.abs().max().item():.3f}), loss {loss.item():.6f} "
          f"vs {loss32.item():.6f}")
    assert e < 2e-5
    assert abs(loss.item() - loss32.item()) < 2e-5 * max(1.0, abs(loss32.item()))
    for k in ("kohonen_consistency", "kohonen_smoothness", "local_quantization", "global_quantization", "reconstruction"):
        assert abs(aux[k].item() - aux32[k].item()) < 2e-5 * max(1.0, abs(aux32[k].item())), k
    for mod, key in ((m.local_kohonen, "local_kohonen.nodes"), (m.global_kohonen, "global_kohonen.nodes")):
        assert (mod.nodes.detach().cpu() - p[key].detach()).abs().max().item() < 5e-6, key
    worst = 0.0
    for n, q in m.named_parameters():
        if q.grad is None or p[n].grad is None:
            continue
        ref = p[n].grad
        err, s = (q.grad.cpu() - ref).abs().max().item(), ref.abs().max().item()
        worst = max(worst, err / (s + 1e-12))
        assert err <= 5e-4 * s + 1e-8, (n, err, s)
    print(f"   fp32 mode: worst relative gradient error {worst:.3e}")
    # bf16 mode (fresh weights: the SOM nodes were updated in place by the forward above)
    pe = O.make_params(formula_state_dict(cfg))
    O.renorm_(pe, cfg)
    lem, _, _ = O.loss_and_grads(pe, cfg, X, y, lowp=O.KernelRounding("bound"), step=1, want_aux=True)
    mb = build(cfg, "bf16", True).train()
    with torch.no_grad():
        lb, auxb = mb(X.cuda())
    e32, eem = (lb.cpu() - l32).abs().max().item(), (lb.cpu() - lem).abs().max().item()
    d_emu = (lem - l32).abs().max().item()
    lmax = l32.abs().max().item()
    print(f"   bf16 mode: max|dlogit| vs bf16-operand oracle {eem:.3e}, vs fp32 oracle {e32:.3e}; oracle bf16-operand vs fp32 "
          f"{d_emu:.3e}")
    # Attribution (tools/parity_attribution.py base_k 2): the SOM indices agree with the oracle on all
    # 2 x 1568 tokens, and at this configuration the four bf16-operand CPU evaluations (see _full_size_bf16_parity) are
    # themselves 1.1-1.7e-3 apart - two of them with IDENTICAL rounding points, differing only in float32 vs float64
    # accumulation, by 1.06e-3.  The residual is the trunk's sensitivity to summation order (it stays 8.8e-4 with the
    # whole cross-attention block computed exactly), not a kernel family, so 1e-3 cannot be met by ANY two evaluations
    # here; the HIP path must be no farther from the nearest emulation than 1e-3 or the emulations' own scatter.
    pf = O.make_params(formula_state_dict(cfg))
    O.renorm_(pf, cfg)
    emus = {"bound/f32": lem}
    with torch.no_grad():
        for tag, lp in (("bound/f64", O.KernelRounding("bound", acc64=True)), ("rowmax/f32", O.KernelRounding("rowmax")),
                        ("rowmax/f64", O.KernelRounding("rowmax", acc64=True))):
            pf = O.make_params(formula_state_dict(cfg))      # (the forward updates the SOM nodes in place)
            O.renorm_(pf, cfg)
            emus[tag] = O.forward(pf, cfg, X, lp, training=True, step=1)[0]
    tags = list(emus)
    spread = max((emus[a] - emus[b]).abs().max().item() for i, a in enumerate(tags) for b in tags[i + 1:])
    floor = (lem - emus["bound/f64"]).abs().max().item()
    dist = {t: (lb.cpu() - e).abs().max().item() for t, e in emus.items()}
    eem = min(dist.values())
    print(f"   HIP vs the four emulations: " + ", ".join(f"{t} {v:.3e}" for t, v in dist.items()) +
          f"; their own scatter {spread:.3e} (summation order alone {floor:.3e})")
    _record_margin("base_k", 2, dict(hip_vs_emulation=eem, hip_vs_fp32=e32, emulation_vs_fp32=d_emu, summation_floor=floor,
                                     emulation_spread=spread, logit_max=lmax))
    assert eem < max(1e-3, spread), (dist, spread)
    assert e32 < d_emu + 1e-3, (e32, d_emu)
