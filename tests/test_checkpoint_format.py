"""Checkpoint / stat formats of the reference trainer: CPU-only (module construction and
state_dict handling need no kernel)."""
import numpy as np
import torch

from nvit_amd.config import named_config
from nvit_amd.weights import formula_state_dict

REF_KEYS = {"model", "optimizer", "model_args", "iter_num", "metrics", "config", "rng_state_pytorch",
            "rng_state_numpy", "timestamp"}   # /root/reference/nvit/train.py:640-650


def test_checkpoint_roundtrip_and_reference_keys(tmp_path):
    from nvit_amd.checkpoint import load_checkpoint, save_checkpoint, stat_row
    from nvit_amd.model import ViT
    cfg = named_config("micro")
    m = ViT(cfg)
    m.load_state_dict(formula_state_dict(cfg))
    opt = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cpu")   # container only on CPU; nothing is stepped here
    path = save_checkpoint(tmp_path / "out" / "checkpoint_latest.pt", m, opt, 17, {"val/loss": 1.5, "train/loss": 1.25},
                           {"training": {"batch_size": 32}})
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(raw) == REF_KEYS
    assert raw["iter_num"] == 17 and raw["model_args"]["n_embd"] == cfg.n_embd
    assert isinstance(raw["rng_state_numpy"], tuple) and raw["rng_state_pytorch"].dtype == torch.uint8
    # every reference state_dict key is there, and nothing private (operand shadows) leaks out
    assert set(raw["model"]) == set(formula_state_dict(cfg))
    m2, opt2, ck = load_checkpoint(path, device="cpu", trusted=True,
                                   optimizer_factory=lambda mm: mm.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cpu"))
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert ck["metrics"]["val/loss"] == 1.5 and opt2 is not None
    row = stat_row(17, 1e-3, {"train/loss": 1.25, "val/loss": 1.5}, m)
    fields = row.split()
    assert fields[0] == "1.700000e+01" and fields[1] == "1.0000e-03" and fields[4:13] == ["0.0:.4e"] * 9
    assert len(fields) == 13 + 1 + 4 * cfg.n_layer
