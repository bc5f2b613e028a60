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
    # default load path: weights_only (nothing from the file is executed), including the numpy RNG tuple
    np.random.seed(123)
    m2, opt2, ck = load_checkpoint(path, device="cpu",
                                   optimizer_factory=lambda mm: mm.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cpu"))
    assert np.array_equal(np.random.get_state()[1], raw["rng_state_numpy"][1])   # numpy RNG restored without trusted=True
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert ck["metrics"]["val/loss"] == 1.5 and opt2 is not None
    row = stat_row(17, 1e-3, {"train/loss": 1.25, "val/loss": 1.5}, m)
    fields = row.split()
    assert fields[0] == "1.700000e+01" and fields[1] == "1.0000e-03" and fields[4:13] == ["0.0:.4e"] * 9
    assert len(fields) == 13 + 1 + 4 * cfg.n_layer


def test_kohonen_checkpoint_has_the_reference_state_dict_keys(tmp_path):
    """A use_kohonen checkpoint carries the maps' persistent buffers `locations` / `offsets` (reference kohonen.py:62,78;
    SURVEY.md §9.5), so the reference's strict load_state_dict accepts it; and it loads back here weights-only."""
    from nvit_amd.checkpoint import load_checkpoint, save_checkpoint
    from nvit_amd.model import ViT
    cfg = named_config("micro_k")
    m = ViT(cfg)
    m.load_state_dict(formula_state_dict(cfg), strict=False)
    opt = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cpu")
    path = save_checkpoint(tmp_path / "ck.pt", m, opt, 3, {"val/loss": 0.0, "train/loss": 0.0})
    raw = torch.load(path, map_location="cpu", weights_only=False)
    want = set(formula_state_dict(cfg)) | {f"{k}_kohonen.{b}" for k in ("local", "global") for b in ("locations", "offsets")}
    assert set(raw["model"]) == want
    assert raw["model"]["local_kohonen.locations"].dtype == torch.int64
    assert tuple(raw["model"]["local_kohonen.offsets"].shape) == (8, 2)
    m2, _, _ = load_checkpoint(path, device="cpu")
    assert torch.equal(m2.local_kohonen.locations, m.local_kohonen.locations)
    assert torch.equal(m2.global_kohonen.nodes, m.global_kohonen.nodes)


class _Weird:   # a global that is not on the weights-only allow list
    pass


def test_untrusted_pickle_is_refused(tmp_path):
    """A file that needs arbitrary unpickling is not loaded unless the caller says it wrote it."""
    import pytest
    from nvit_amd.checkpoint import load_checkpoint

    path = tmp_path / "bad.pt"
    torch.save({"model": {}, "model_args": {}, "x": _Weird()}, path)
    with pytest.raises(RuntimeError):
        load_checkpoint(path, device="cpu")


def test_public_helpers_present():
    """API the reference exposes and callers may reach (model.py:43-44,89-90,477-480; kohonen.py:80-98)."""
    from nvit_amd import model as M
    from nvit_amd.kohonen import KohonenMap
    assert callable(M.justnorm) and callable(M.Block.justnorm) and callable(M.ViT.combine_representations)
    km = KohonenMap(8, 16)
    d = km.get_neighborhood_distances(torch.tensor([0, 3]))
    # periodic 4x4 grid: node (0,3) itself 0; (0,0) is one step away through the wrap; (2,1) is 2 rows, 2 cols away
    assert d.shape == (16,) and d[3].item() == 0.0 and d[0].item() == 1.0 and d[2 * 4 + 1].item() == 8.0
    # against the oracle's restatement of kohonen.py:80-98
    from oracle import nvit_oracle as O
    for loc in ([0, 0], [1, 2], [3, 3]):
        assert torch.equal(km.get_neighborhood_distances(torch.tensor(loc)),
                           O.som_neighborhood_d2(torch.tensor(loc), km.m, km.n))


def test_reference_import_lines_resolve_to_this_implementation():
    """`from nvit.model import ViT, ViTConfig` (reference train.py:37) and `from nvit.kohonen import KohonenMap`
    (reference model.py:10) work unchanged and give the nvit_amd classes (nvit/ is an import shim)."""
    import importlib
    m = importlib.import_module("nvit.model")
    k = importlib.import_module("nvit.kohonen")
    from nvit_amd import model as M
    from nvit_amd.config import ViTConfig
    from nvit_amd.kohonen import KohonenMap
    assert m.ViT is M.ViT and m.ViTConfig is ViTConfig and m.Block is M.Block
    assert m.CrossAttentionBlock is M.CrossAttentionBlock and m.RMSNorm is M.RMSNorm and m.justnorm is M.justnorm
    assert k.KohonenMap is KohonenMap
    import nvit
    assert len(list(nvit.__path__)) >= 1
