"""Golden-vector generator — runs ONLY in the build container (needs /root/reference).

Imports the real reference model (`/root/reference/nvit/model.py`) on CPU, loads the
closed-form formula weights (nvit_amd/weights.py) through `load_state_dict`, and
records small input/output vectors under tests/golden/.  The reference never
travels; only these numbers do.  `flash_attn` is not installed (and is only called
when config.flash_attn=True, model.py:121-122), so an empty stub module is
registered before the import.

`nvit/train.py` cannot be imported (kornia/wandb/dynaconf/torchvision absent, CUDA
required at train.py:110-111), so the step driver below re-states its order
(train.py:898-946,989-990) around the *reference's own* model, optimizer factory
(`ViT.configure_optimizers`) and torch ops; normalize_matrices (train.py:461-480)
is applied to the reference module's weights with torch ops.

Full-size pins (round 4): the BASELINE configurations themselves - Base (C2) B=6, Large (C4) B=2,
Base + Kohonen (C5) B=2, post-renorm weight state, the shapes tests/test_gpu_model.py runs - are
recorded from the imported reference in fp32 (logits, loss, aux losses, per-parameter gradient norms
and slices, one optimizer step), and the reference's OWN bf16 path (`torch.autocast("cpu",
dtype=torch.bfloat16)` around `model(X)`, train.py:254,905; logits only) is recorded for
tiny / mini / base / large / base_k, so that the bf16 deviation of the HIP path is bounded by
reference-held data: |HIP_bf16 - ref_fp32| <= |ref_autocast_bf16 - ref_fp32|.

Usage:  python oracle/make_golden.py            (writes the small-config tests/golden/*.npz)
        python oracle/make_golden.py full       (the full-size fp32 pins; minutes of CPU)
        python oracle/make_golden.py autocast   (the reference's bf16-autocast logits)
"""
from __future__ import annotations

import os
import sys
import types

sys.dont_write_bytecode = True
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _REPO)

import numpy as np
import torch
import torch.nn.functional as F

_stub = types.ModuleType("flash_attn")


def _no_flash(*a, **k):
    raise RuntimeError("flash_attn is not available in this container")


_stub.flash_attn_func = _no_flash
sys.modules.setdefault("flash_attn", _stub)
sys.path.insert(0, "/root/reference")
from nvit.model import ViT as RefViT, ViTConfig as RefConfig  # noqa: E402

from dataclasses import asdict  # noqa: E402

from nvit_amd.config import named_config  # noqa: E402
from nvit_amd.weights import formula_state_dict, synthetic_batch  # noqa: E402

OUT = os.path.join(_REPO, "tests", "golden")
CASES = [  # (config name, batch)
    ("micro", 8),
    ("mini", 4),
    ("tiny", 32),
    ("micro_k", 8),   # Kohonen head (BASELINE config C5 semantics at parity size)
    ("mini_k", 4),
]
FULL_CASES = [("base", 6), ("large", 2), ("base_k", 2)]      # renormed state only (the state the GPU tests run)
AUTOCAST_CASES = [("tiny", 32), ("mini", 4), ("base", 6), ("large", 2), ("base_k", 2)]
AUX_KEYS = ("kohonen_consistency", "kohonen_smoothness", "local_quantization", "global_quantization")


def ref_total_loss(cfg, logits, aux, y):
    """train.py:906-926 with settings.yaml consistency_weight = smoothness_weight = 0.1."""
    loss = F.cross_entropy(logits, y)
    if cfg.use_kohonen:
        loss = (loss + 0.1 * aux["kohonen_consistency"] + 0.1 * aux["kohonen_smoothness"]
                + cfg.local_quantization_weight * aux["local_quantization"]
                + cfg.global_quantization_weight * aux["global_quantization"]
                + cfg.reconstruction_weight * aux["reconstruction"])
    return loss


@torch.no_grad()
def ref_normalize_matrices(model) -> None:
    for blk in model.transformer.h:
        for lin, dim in ((blk.query, 1), (blk.key, 1), (blk.value, 1), (blk.att_c_proj, 0),
                         (blk.c_fc, 1), (blk.mlp_c_proj, 0)):
            w = lin.weight.data
            lin.weight.data.copy_((w.float() / w.float().norm(p=2, dim=dim, keepdim=True)).to(w.dtype))


def known_answer_check() -> None:
    """validates this harness (torch RNG init, not committed)."""
    torch.manual_seed(0)
    cfg = RefConfig(use_nvit=True, flash_attn=False, image_size=32, n_layer=12, n_head=3, n_embd=192,
                    num_classes=10, use_kohonen=False, bias=False, dropout=0.0)
    m = RefViT(cfg).train()
    x = torch.randn(32, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    y = torch.randint(0, 10, (32,), generator=torch.Generator().manual_seed(2))
    logits, aux = m(x)
    ce = F.cross_entropy(logits, y)
    want = torch.tensor([-0.108308, -0.110730, 0.636149, -0.235373, 0.476821])
    assert torch.allclose(logits[0, :5], want, atol=2e-6), logits[0, :5]
    assert abs(ce.item() - 2.243258) < 2e-6 and abs(aux["reconstruction"].item() - 1.002458) < 2e-6
    print("known-answer record (SURVEY §9.3) reproduced")


def one_case(name: str, batch: int, renormed: bool) -> dict:
    cfg = named_config(name)
    ref = RefViT(RefConfig(**asdict(cfg)))
    sd = formula_state_dict(cfg, perturb_scalars=True)
    res = ref.load_state_dict(sd, strict=False)   # only the Kohonen index buffers may be absent from the formula dict
    assert not res.unexpected_keys and all(k.endswith((".locations", ".offsets")) for k in res.missing_keys), res
    if renormed:
        ref_normalize_matrices(ref)
    ref.train()
    X, y = synthetic_batch(cfg, batch)
    logits, aux = ref(X)
    loss = ref_total_loss(cfg, logits, aux, y)
    loss.backward()
    rec = {"logits": logits.detach().numpy(), "loss": np.float64(loss.item()),
           "recon": np.float64(aux["reconstruction"].item())}
    if cfg.use_kohonen:
        rec["aux"] = np.array([aux[k].item() for k in AUX_KEYS])
        rec["lnodes_head"] = ref.local_kohonen.nodes.detach().reshape(-1)[:8].numpy().copy()   # after the SOM update
        rec["gnodes_head"] = ref.global_kohonen.nodes.detach().reshape(-1)[:8].numpy().copy()
    names, gn, heads_ = [], [], []
    for n, p in ref.named_parameters():
        if p.grad is None:
            continue
        names.append(n)
        gn.append(p.grad.double().norm().item())
        heads_.append(p.grad.reshape(-1)[:8].numpy().copy() if p.grad.numel() >= 8
                      else np.resize(p.grad.reshape(-1).numpy(), 8))
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(gn)
    rec["grad_heads"] = np.stack(heads_)
    # one optimizer step in the reference's order
    opt = ref.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cpu")
    gnorm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
    opt.step()
    opt.zero_grad(set_to_none=True)
    ref_normalize_matrices(ref)
    rec["gnorm"] = np.float64(gnorm.item())
    with torch.no_grad():
        logits1, aux1 = ref(X)
        rec["logits1"] = logits1.numpy()
        rec["loss1"] = np.float64(ref_total_loss(cfg, logits1, aux1, y).item())
        rec["recon1"] = np.float64(aux1["reconstruction"].item())
        w = ref.transformer.h[0].query.weight
        rec["q0_head1"] = w.reshape(-1)[:8].numpy().copy()
        wp = ref.transformer.h[-1].mlp_c_proj.weight
        rec["p_last_head1"] = wp.reshape(-1)[:8].numpy().copy()
    return rec


def build_ref(name: str, renormed: bool):
    cfg = named_config(name)
    ref = RefViT(RefConfig(**asdict(cfg)))
    res = ref.load_state_dict(formula_state_dict(cfg, perturb_scalars=True), strict=False)
    assert not res.unexpected_keys and all(k.endswith((".locations", ".offsets")) for k in res.missing_keys), res
    if renormed:
        ref_normalize_matrices(ref)
    return cfg, ref.train()


def autocast_case(name: str, batch: int) -> dict:
    """The reference's own bf16 path on the CPU (train.py:254 builds `torch.autocast(device_type, dtype)`; the
    forward runs inside it at train.py:905), next to its fp32 path, same weights / inputs, train mode, renormed."""
    cfg, ref = build_ref(name, True)
    X, _ = synthetic_batch(cfg, batch)
    with torch.no_grad():
        l32, _ = ref(X)
    cfg, ref = build_ref(name, True)            # fresh module: the Kohonen forward mutates the SOM nodes
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        lbf, _ = ref(X)
    lbf = lbf.float()
    d = (lbf - l32).abs()
    return {"logits_fp32": l32.numpy(), "logits_autocast_bf16": lbf.numpy(),
            "max_abs_dev": np.float64(d.max().item()), "rms_dev": np.float64(d.double().pow(2).mean().sqrt().item()),
            "logit_max": np.float64(l32.abs().max().item())}


def main() -> None:
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "4")))
    known_answer_check()
    os.makedirs(OUT, exist_ok=True)
    mode = sys.argv[1] if len(sys.argv) > 1 else "small"
    if mode == "full":
        for name, batch in FULL_CASES:
            rec = one_case(name, batch, True)
            path = os.path.join(OUT, f"{name}_b{batch}_renorm.npz")
            np.savez_compressed(path, **rec)
            print(path, "loss", rec["loss"], "loss1", rec["loss1"], "gnorm", rec["gnorm"], flush=True)
        return
    if mode == "autocast":
        for name, batch in AUTOCAST_CASES:
            rec = autocast_case(name, batch)
            path = os.path.join(OUT, f"{name}_b{batch}_autocast.npz")
            np.savez_compressed(path, **rec)
            print(path, "ref autocast-bf16 vs ref fp32: max", rec["max_abs_dev"], "rms", rec["rms_dev"],
                  "|logit|max", rec["logit_max"], flush=True)
        return
    for name, batch in CASES:
        for renormed in (False, True):
            rec = one_case(name, batch, renormed)
            path = os.path.join(OUT, f"{name}_b{batch}_{'renorm' if renormed else 'init'}.npz")
            np.savez_compressed(path, **rec)
            print(path, "loss", rec["loss"], "loss1", rec["loss1"], "gnorm", rec["gnorm"])


if __name__ == "__main__":
    main()
