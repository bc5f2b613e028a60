"""AdamW whose step is fused with gradient clipping and the nGPT weight re-normalisation.

`FusedAdamW` IS a `torch.optim.AdamW` (same constructor, param groups, `state_dict()` layout: per-parameter
`step`, `exp_avg`, `exp_avg_sq`), so everything the reference trainer does with its optimizer keeps working
(/root/reference/nvit/train.py:456-459 build, :942-946 step / zero_grad, :640-650 checkpointing).  What changes
is how a step runs on the MI355X: `step_fused(model, grad_clip)` performs

    clip_grad_norm_(params, grad_clip)  ->  AdamW.step()  ->  Trainer.normalize_matrices()
    (train.py:935-941)                     (train.py:942-944)  (train.py:461-480, 989-990)

in two HIP launches (`nvit_grad_sqnorm`, `nvit_adamw_renorm`) over a device-side parameter table, reading each of
p, g, m, v once and writing p, m, v once.  Plain `step()` (no clip, no renorm) uses the same kernel, so code that
calls `optimizer.step()` followed by `normalize_matrices(model)` gives identical weights.  There is no torch
fallback: the parameters must live on the HIP device.
"""
from __future__ import annotations

import math
import struct
from typing import Dict, List, Optional

import torch

from . import _lib
from .ops import _p, _s, check

_CHUNK = 8192
_ROWS_PER_ITEM = 16
_SLAB_COLS = 32
_NPART = 1024


def _f32_bits(x: float) -> int:
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


class FusedAdamW(torch.optim.AdamW):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        # the torch base class is only the container (param groups, state, state_dict); its kernels never run
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, fused=False)
        self._cache = None

    # ------------------------------------------------------------------ state
    def _ensure_state(self, p: torch.Tensor) -> Dict:
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _table(self, renorm_dims: Dict[int, int]):
        """Device table for the current (param, grad, state) pointers; rebuilt only when one of them moves."""
        rows: List[List[int]] = []
        key = []
        first_item = first_chunk = 0
        max_slab_rows = 0
        beta_eps = None
        for gi, group in enumerate(self.param_groups):
            be = (group["betas"][0], group["betas"][1], group["eps"])
            if group.get("amsgrad") or group.get("maximize"):
                raise RuntimeError("FusedAdamW: amsgrad / maximize are not supported")
            for p in group["params"]:
                if p.grad is None:
                    continue
                if beta_eps is None:
                    beta_eps = be
                elif be != beta_eps:
                    raise RuntimeError("FusedAdamW: all parameter groups must share betas and eps")
                if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdamW: parameters must be contiguous fp32 tensors on the HIP device")
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous() or g.is_sparse:
                    raise RuntimeError("FusedAdamW: gradients must be dense contiguous fp32")
                st = self._ensure_state(p)
                kind = renorm_dims.get(id(p), -1)
                if kind >= 0 and p.dim() == 2:
                    r, c = p.shape
                else:
                    kind, r, c = -1, 1, p.numel()
                if kind == 1:
                    if c % 4 or c > 1536:
                        raise RuntimeError(f"FusedAdamW: row-normalised matrix needs cols % 4 == 0 and <= 1536 (got {c})")
                    items = math.ceil(r / _ROWS_PER_ITEM)
                elif kind == 0:
                    if r > 1152:
                        raise RuntimeError(f"FusedAdamW: column-normalised matrix with {r} rows exceeds the LDS slab")
                    items = math.ceil(c / _SLAB_COLS)
                    max_slab_rows = max(max_slab_rows, r)
                else:
                    items = math.ceil(p.numel() / _CHUNK)
                hyper = _f32_bits(group["lr"]) | (_f32_bits(group["weight_decay"]) << 32)
                if hyper >= 1 << 63:
                    hyper -= 1 << 64
                rows.append([p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                             r, c, kind, first_item, first_chunk, hyper])
                key.append((p.data_ptr(), g.data_ptr(), kind, group["lr"], group["weight_decay"]))
                first_item += items
                first_chunk += math.ceil(p.numel() / _CHUNK)
        key = tuple(key)
        if self._cache is None or self._cache["key"] != key:
            if not rows:
                return None
            dev = None
            for group in self.param_groups:
                for p in group["params"]:
                    dev = p.device
                    break
                if dev is not None:
                    break
            self._cache = {
                "key": key,
                "table": torch.tensor(rows, dtype=torch.int64).to(dev),
                "n": len(rows), "items": first_item, "chunks": first_chunk, "slab": max_slab_rows,
                "partial": torch.empty(_NPART, device=dev, dtype=torch.float32),
                "gnorm": torch.empty(1, device=dev, dtype=torch.float32),
                "betas_eps": beta_eps,
            }
        return self._cache

    # ------------------------------------------------------------------ steps
    @torch.no_grad()
    def step_fused(self, model=None, grad_clip: float = 0.0) -> Optional[torch.Tensor]:
        """clip (if grad_clip > 0) + AdamW + (if `model` is given) normalize_matrices; returns the pre-clip grad norm
        as a 1-element device tensor when clipping is on."""
        dims: Dict[int, int] = {}
        if model is not None:
            m = model.module if hasattr(model, "module") else model
            if m.config.use_nvit:
                for blk in m.transformer.h:
                    for nme in ("query", "key", "value", "c_fc"):
                        dims[id(getattr(blk, nme).weight)] = 1
                    dims[id(blk.att_c_proj.weight)] = 0
                    dims[id(blk.mlp_c_proj.weight)] = 0
        c = self._table(dims)
        if c is None:
            return None
        # per-parameter step counters (host side, as torch's non-capturable AdamW keeps them)
        t = None
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                st["step"] += 1
                tp = float(st["step"])
                if t is None:
                    t = tp
                elif tp != t:
                    raise RuntimeError("FusedAdamW: parameters with different step counts are not supported")
        b1, b2, eps = c["betas_eps"]
        lib = _lib.load()
        clip = grad_clip is not None and grad_clip > 0.0
        if clip:
            check(lib.nvit_grad_sqnorm(_p(c["table"]), c["n"], c["chunks"], _p(c["partial"]), _NPART, _s()),
                  "nvit_grad_sqnorm")
        check(lib.nvit_adamw_renorm(_p(c["table"]), c["n"], c["items"], c["slab"], b1, b2, eps, 1.0 - b1 ** t,
                                    1.0 - b2 ** t, _p(c["partial"]) if clip else None, _NPART if clip else 0,
                                    float(grad_clip) if clip else 0.0, _p(c["gnorm"]) if clip else None, _s()),
              "nvit_adamw_renorm")
        return c["gnorm"] if clip else None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.step_fused(None, 0.0)
        return loss
