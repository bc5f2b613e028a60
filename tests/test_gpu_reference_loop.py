"""The reference trainer's OWN step sequence, run on the HIP model through the `nvit` import shim.

`nvit/train.py` itself cannot be imported here (kornia / wandb / dynaconf / torchvision are absent),
so this file re-enacts what its hot loop does with the model, in the loop's order and with the torch pieces the loop
uses (citations into /root/reference/nvit/train.py):

    from nvit.model import ViT, ViTConfig                                   :37   (resolved by the repo's nvit/ shim)
    ViT(ViTConfig(**model_args)); model.to(device)                          :422, :431
    optimizer = model.configure_optimizers(wd, lr, (b1, b2), device)        :124-129
    scaler = GradScaler()              (enabled for bfloat16 too)           :134-136
    ctx = autocast(device_type, dtype=bfloat16)                             :254
    with context, ctx: logits, aux_losses = model(X); F.cross_entropy       :904-906
        total_loss += weight * aux (in place, Kohonen only); / grad_accum   :909-928
    scaler.scale(total_loss).backward()                                     :930-931
    scaler.unscale_(optimizer); clip_grad_norm_(model.parameters(), clip)   :935-938
    scaler.step(optimizer); scaler.update()                                 :940-942
    optimizer.zero_grad(set_to_none=True)                                   :946
    normalize_matrices(): six `.weight.data.copy_(x.float()/x.float().norm(dim=k))` per block   :461-480, :989-990

Nothing from nvit_amd.train (the fused step) is used here: the optimizer step arrives through
`scaler.step -> FusedAdamW.step()` (the unfused entry), the clip is torch's, the re-normalisation is the trainer's
torch-op form.  fp32 mode is held to the golden vectors recorded from the real reference (tests/golden); bf16 mode
(the mode the trainer's autocast setting means) to the CPU oracle's loss trajectory."""
import os
from contextlib import nullcontext
from dataclasses import asdict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from nvit_amd.config import named_config
from nvit_amd.weights import formula_state_dict, synthetic_batch
from oracle import nvit_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _trainer_normalize_matrices(model) -> None:
    """What Trainer.normalize_matrices does (train.py:461-480): rows of q/k/v/c_fc, columns of the two c_proj, fp32,
    written back through `.weight.data.copy_` - with torch operators, as the trainer has it."""
    for block in model.transformer.h:
        for lin, dim in ((block.query, 1), (block.key, 1), (block.value, 1), (block.att_c_proj, 0), (block.c_fc, 1),
                         (block.mlp_c_proj, 0)):
            w = lin.weight.data
            w32 = w.float()
            lin.weight.data.copy_((w32 / w32.norm(p=2, dim=dim, keepdim=True)).to(dtype=w.dtype))


def _trainer_iteration(model, optimizer, scaler, ctx, X, y, grad_clip=1.0, grad_accum=1, consistency_weight=0.1,
                       smoothness_weight=0.1):
    """One pass of the batch loop body, train.py:898-946 + :989-990."""
    total_loss = torch.tensor(torch.inf, device=X.device)
    for micro_step in range(grad_accum):
        context = nullcontext()       # (DDP no_sync() branch: single process here)
        with context, ctx:
            logits, aux_losses = model(X)
            class_loss = F.cross_entropy(logits, y)
            total_loss = class_loss
            if model.config.use_kohonen:
                total_loss += consistency_weight * aux_losses["kohonen_consistency"]
                total_loss += smoothness_weight * aux_losses["kohonen_smoothness"]
                total_loss += model.config.local_quantization_weight * aux_losses["local_quantization"]
                total_loss += model.config.global_quantization_weight * aux_losses["global_quantization"]
                total_loss += model.config.reconstruction_weight * aux_losses["reconstruction"]
            total_loss = total_loss / grad_accum
        if scaler is not None:
            scaler.scale(total_loss).backward()
        else:
            total_loss.backward()
    if grad_clip != 0.0:
        if scaler is not None:
            scaler.unscale_(optimizer)
        torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
    if scaler is not None:
        scaler.step(optimizer)
        scaler.update()
    else:
        optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    if model.config.use_nvit:
        _trainer_normalize_matrices(model)
    return logits.detach(), total_loss.detach() * grad_accum


def _build_through_shim(cfg, precision):
    from nvit.model import ViT, ViTConfig          # train.py:37, served by the repo's import shim
    model_args = asdict(cfg)
    model = ViT(ViTConfig(**model_args))           # train.py:422
    res = model.load_state_dict(formula_state_dict(cfg), strict=False)
    assert not res.unexpected_keys and all(k.endswith((".locations", ".offsets")) for k in res.missing_keys)
    model.to("cuda:0")                             # train.py:431
    model.precision = precision                    # the one line the trainer does not have (default: NVIT_PRECISION or bf16)
    return model.train()


@pytest.mark.parametrize("name,batch", [("micro", 8), ("tiny", 32)])
def test_reference_loop_fp32_mode_matches_reference_golden(name, batch):
    g = np.load(os.path.join(GOLD, f"{name}_b{batch}_init.npz"))
    cfg = named_config(name)
    X, y = synthetic_batch(cfg, batch)
    X, y = X.cuda(), y.cuda()
    model = _build_through_shim(cfg, "fp32")
    optimizer = model.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    scaler = torch.amp.GradScaler("cuda")
    ctx = torch.autocast(device_type="cuda", dtype=torch.bfloat16)
    logits, loss = _trainer_iteration(model, optimizer, scaler, ctx, X, y)
    e0 = np.abs(logits.cpu().numpy() - g["logits"]).max()
    assert e0 < 2e-5, e0
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    model.eval()
    with torch.no_grad(), ctx:
        logits1, _ = model(X)
    e1 = np.abs(logits1.cpu().numpy() - g["logits1"]).max()
    print(f"[reference loop, fp32 mode, {name}] step-0 max|dlogit| {e0:.2e}, step-1 {e1:.2e} (golden from the real reference)")
    assert e1 < 1e-4, e1
    q0 = model.transformer.h[0].query.weight.detach().reshape(-1)[:8].cpu().numpy()
    assert np.abs(q0 - g["q0_head1"]).max() < 2e-6
    for blk in model.transformer.h:
        assert (blk.c_fc.weight.detach().norm(dim=1) - 1).abs().max().item() < 1e-6
        assert (blk.mlp_c_proj.weight.detach().norm(dim=0) - 1).abs().max().item() < 1e-6
    st = optimizer.state_dict()["state"]
    assert all(float(v["step"]) == 1.0 for v in st.values())


@pytest.mark.parametrize("name,batch", [("micro", 8), ("tiny", 32), ("mini_k", 4)])
def test_reference_loop_bf16_mode_tracks_oracle_losses(name, batch):
    """The trainer's actual setting (autocast bf16 + GradScaler): three iterations, loss per iteration against the CPU
    oracle's fp32 train step (train.py semantics on CPU: no autocast, :254), final weights unit-norm."""
    cfg = named_config(name)
    X, y = synthetic_batch(cfg, batch)
    p = O.make_params(formula_state_dict(cfg))
    oopt = O.make_optimizer(p)
    want = []
    for it in range(3):
        if cfg.use_kohonen:
            for t in p.values():
                t.grad = None
            lg, aux = O.forward(p, cfg, X, None, training=True, step=it + 1)
            ls = O.total_loss(cfg, lg, aux, y)
            ls.backward()
            torch.nn.utils.clip_grad_norm_([t for t in p.values() if t.grad is not None], 1.0)
            oopt.step()
            oopt.zero_grad(set_to_none=True)
            O.renorm_(p, cfg)
            want.append(ls.item())
        else:
            want.append(O.train_step(p, cfg, oopt, X, y)[1].item())
    model = _build_through_shim(cfg, "bf16")
    optimizer = model.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    scaler = torch.amp.GradScaler("cuda")
    ctx = torch.autocast(device_type="cuda", dtype=torch.bfloat16)
    got = [_trainer_iteration(model, optimizer, scaler, ctx, X.cuda(), y.cuda())[1].item() for _ in range(3)]
    print(f"[reference loop, bf16 mode, {name}] losses {got} vs oracle {want}")
    for a, b in zip(got, want):
        assert abs(a - b) < 3e-3 * max(1.0, abs(b)), (got, want)
    assert scaler.get_scale() == 65536.0      # no overflow was ever reported: the scaled backward stayed finite
    for prm in model.parameters():
        assert torch.isfinite(prm).all()


def test_public_api_numerics_vs_oracle():
    """Numeric checks of the public entry points callers of the reference can reach: module-level `justnorm`
    (model.py:43-44), `Block.justnorm` (:89-90), `ViT.combine_representations` (:477-480) and
    `CrossAttentionBlock.forward(local, global_)` (:219-275), forward values and input gradients."""
    from nvit.model import justnorm
    from nvit_amd.train import normalize_matrices
    cfg = named_config("mini")
    m = _build_through_shim(cfg, "fp32")
    normalize_matrices(m)
    g = torch.Generator().manual_seed(11)
    B, T, C = 3, m.n_tokens, cfg.n_embd
    x = torch.randn(B, T, C, generator=g)
    w = torch.randn(B, T, C, generator=g)
    # justnorm / Block.justnorm: value and gradient
    for fn in (justnorm, m.transformer.h[0].justnorm):
        xr = x.clone().requires_grad_(True)
        (O.nrm(xr) * w).sum().backward()
        xg = x.cuda().requires_grad_(True)
        out = fn(xg)
        (out * w.cuda()).sum().backward()
        assert (out.detach().cpu() - O.nrm(x)).abs().max().item() < 1e-6
        assert (xg.grad.cpu() - xr.grad).abs().max().item() < 2e-6 * max(1.0, xr.grad.abs().max().item())
    # a bf16 input keeps its dtype (the reference's justnorm is dtype-preserving outside autocast)
    assert justnorm(x.cuda().bfloat16()).dtype == torch.bfloat16
    # combine_representations = nrm(a * b)
    a, b = torch.randn(B, T, C, generator=g), torch.randn(B, T, C, generator=g)
    got = m.combine_representations(a.cuda(), b.cuda()).cpu()
    assert (got - O.nrm(a * b)).abs().max().item() < 1e-6
    # CrossAttentionBlock.forward as a public call, un-normalised inputs like the patch embeddings
    p = O.make_params(formula_state_dict(cfg))
    O.renorm_(p, cfg)
    loc, glo = torch.randn(B, T, C, generator=g), torch.randn(B, T, C, generator=g)
    lr_, gr_ = loc.clone().requires_grad_(True), glo.clone().requires_grad_(True)
    ref = O.cross_block(p, cfg, lr_, gr_, None)
    (ref * w).sum().backward()
    lg_, gg_ = loc.cuda().requires_grad_(True), glo.cuda().requires_grad_(True)
    out = m.cross_attention(lg_, gg_)
    (out * w.cuda()).sum().backward()
    assert out.shape == (B, T, C)
    assert (out.detach().cpu() - ref.detach()).abs().max().item() < 2e-6
    for got_g, ref_g in ((lg_.grad, lr_.grad), (gg_.grad, gr_.grad)):
        assert (got_g.cpu() - ref_g).abs().max().item() < 3e-5 * max(1.0, ref_g.abs().max().item())
    pw = m.cross_attention.q_local.weight.grad.cpu()
    rw = p["cross_attention.q_local.weight"].grad
    assert (pw - rw).abs().max().item() < 2e-4 * rw.abs().max().item() + 1e-8


def test_rmsnorm_module_dtypes_vs_reference_formula():
    """`RMSNorm` (reference model.py:170-182; dead on the nViT path, a working public module): computed in fp32, the
    normalised value cast back to the INPUT dtype, then multiplied by the fp32 weight - for fp32 and bf16 inputs."""
    from nvit.model import RMSNorm
    g = torch.Generator().manual_seed(3)
    mod = RMSNorm(256).cuda()
    with torch.no_grad():
        mod.weight.copy_(torch.rand(256, generator=g) + 0.5)
    x = torch.randn(5, 7, 256, generator=g)
    w = mod.weight.detach().cpu()

    def ref(t):
        t32 = t.float()
        n = t32 * torch.rsqrt(t32.pow(2).mean(-1, keepdim=True) + mod.eps)
        return w * n.to(t.dtype)

    out = mod(x.cuda())
    assert out.dtype == torch.float32 and (out.cpu() - ref(x)).abs().max().item() < 2e-6
    xb = x.bfloat16()
    outb = mod(xb.cuda())
    assert outb.dtype == ref(xb).dtype
    assert (outb.float().cpu() - ref(xb).float()).abs().max().item() < 2e-2   # one bf16 ulp of values up to ~4
    xg = x.cuda().requires_grad_(True)
    mod(xg).sum().backward()
    xr = x.clone().requires_grad_(True)
    ref(xr).sum().backward()
    assert (xg.grad.cpu() - xr.grad).abs().max().item() < 2e-5
