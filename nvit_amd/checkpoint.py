"""On-disk formats of the reference trainer, so artefacts move both ways.

* checkpoint dict: same keys and value kinds as `Trainer.save_checkpoint` (/root/reference/nvit/train.py:628-650) —
  `model` (state_dict with the reference's names/shapes; the private bf16 operand shadows and the Kohonen index
  buffers are not part of it), `optimizer` (FusedAdamW keeps torch.optim.AdamW's state_dict layout), `model_args`
  (`asdict(ViTConfig)`), `iter_num`, `metrics`, `config`, `rng_state_pytorch`, `rng_state_numpy`, `timestamp`.
  A checkpoint written by the reference loads here and vice versa (`Trainer.load_checkpoint`, train.py:375-395).
* `out/stat` row: `Trainer.write_statistics` / `get_hparams_str` (train.py:1037-1072), same fields and formatting.

Files are read with `weights_only=True` (nothing from the file is executed): tensors, numbers, strings, containers,
plus the three numpy reconstruction globals that the `rng_state_numpy` tuple (`np.random.get_state()`) pickles to,
allow-listed for this load only.  A file that still does not load that way is refused unless `trusted=True`, which
falls back to a full unpickle - only for files you wrote yourself.
"""
from __future__ import annotations

import time
from dataclasses import asdict
from pathlib import Path
from typing import Any, Dict, Optional

import numpy as np
import torch

from .config import ViTConfig


def _model_state(model) -> Dict[str, torch.Tensor]:
    """The reference's `model.state_dict()` key set, including the Kohonen maps' persistent index buffers
    `locations` / `offsets` (kohonen.py:62,78): `Trainer.load_checkpoint` loads it with strict=True."""
    m = model.module if hasattr(model, "module") else model
    return dict(m.state_dict())


def _numpy_safe_globals():
    """Globals a pickled `np.random.get_state()` tuple refers to (array reconstruction only, no code)."""
    try:
        from numpy._core import multiarray as ma      # numpy >= 1.26 / 2.x
    except ImportError:                                # older numpy: same function, old module path
        from numpy.core import multiarray as ma
    allow = [ma._reconstruct, np.ndarray, np.dtype, type(np.dtype(np.uint32))]
    # files written under numpy 1.x name the same function through the old module path
    allow.append((ma._reconstruct, "numpy.core.multiarray._reconstruct"))
    return allow


def build_checkpoint(model, optimizer, iter_num: int, metrics: Dict[str, float], config: Optional[Dict[str, Any]] = None,
                     rng_state_pytorch: Optional[torch.Tensor] = None) -> Dict[str, Any]:
    m = model.module if hasattr(model, "module") else model
    return {
        "model": _model_state(m),
        "optimizer": optimizer.state_dict(),
        "model_args": asdict(m.config),
        "iter_num": iter_num,
        "metrics": dict(metrics),
        "config": dict(config or {}),
        "rng_state_pytorch": torch.get_rng_state() if rng_state_pytorch is None else rng_state_pytorch,
        "rng_state_numpy": np.random.get_state(),
        "timestamp": time.strftime("%d_%m_%Y-%Hh%Mm"),
    }


def save_checkpoint(path, model, optimizer, iter_num: int, metrics: Dict[str, float],
                    config: Optional[Dict[str, Any]] = None) -> Path:
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(build_checkpoint(model, optimizer, iter_num, metrics, config), path)
    return path


def load_checkpoint(path, device="cuda", optimizer_factory=None, restore_rng: bool = True, trusted: bool = False):
    """-> (model, optimizer | None, checkpoint dict).  `optimizer_factory(model)` builds the optimizer whose state is
    then restored (e.g. `lambda m: m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")`)."""
    from .model import ViT
    import pickle
    allow = _numpy_safe_globals()      # built outside the try: an import problem here is not a property of the file
    try:
        with torch.serialization.safe_globals(allow):
            ck = torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:   # (missing / unreadable files raise their own OSError untouched)
        if not trusted:
            raise RuntimeError(f"{path}: not loadable with weights_only=True ({type(e).__name__}: {e}); pass "
                               "trusted=True only for a file you wrote yourself") from e
        ck = torch.load(path, map_location="cpu", weights_only=False)
    model = ViT(ViTConfig(**ck["model_args"]))
    res = model.load_state_dict(ck["model"], strict=False)
    bad = [k for k in res.missing_keys if not k.endswith((".locations", ".offsets"))]
    if bad or res.unexpected_keys:
        raise RuntimeError(f"checkpoint/model mismatch: missing {bad}, unexpected {list(res.unexpected_keys)}")
    model = model.to(device)
    opt = None
    if optimizer_factory is not None:
        opt = optimizer_factory(model)
        opt.load_state_dict(ck["optimizer"])
    if restore_rng:
        if "rng_state_pytorch" in ck and ck["rng_state_pytorch"] is not None:
            torch.set_rng_state(ck["rng_state_pytorch"].cpu())
        if ck.get("rng_state_numpy") is not None:
            np.random.set_state(ck["rng_state_numpy"])
    return model, opt, ck


def hparams_str(model) -> str:
    """`Trainer.get_hparams_str` (train.py:1037-1060): mean effective sz, then per block sqk / attn_alpha / mlp_alpha / suv."""
    m = model.module if hasattr(model, "module") else model
    if not m.config.use_nvit:
        return ""
    cfg = m.config
    s = f"{torch.mean(m.sz * (cfg.sz_init_value / cfg.sz_init_scaling)):.5f} "
    for blk in m.transformer.h:
        s += f"{torch.mean(blk.sqk * (blk.sqk_init_value / blk.sqk_init_scaling)):.5f} "
        s += f"{torch.mean(blk.attn_alpha * (blk.attn_alpha_init_value / blk.attn_alpha_init_scaling)):.5f} "
        s += f"{torch.mean(blk.mlp_alpha * (blk.mlp_alpha_init_value / blk.mlp_alpha_init_scaling)):.5f} "
        s += f"{torch.mean(blk.suv * (blk.suv_init_value / blk.suv_init_scaling)):.5f} "
    return s


def stat_row(iter_num: int, lr: float, losses: Dict[str, float], model) -> str:
    """One line of `out/stat` exactly as `Trainer.write_statistics` formats it (train.py:1063-1072), including its nine
    literal placeholder fields."""
    row = f"{iter_num:.6e} {lr:.4e} {losses['train/loss']:.4e} {losses['val/loss']:.4e} "
    row += "0.0:.4e " * 9
    return row + hparams_str(model) + "\n"
