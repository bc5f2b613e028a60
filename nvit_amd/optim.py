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

One step counter serves all parameters (it lives on the device so that a captured step replays correctly).  torch
keeps one per parameter; the two agree as long as every parameter that is ever updated receives a gradient from the
first step on - true for this model, whose never-updated parameters (rmsnorm_*, the recon head without the Kohonen
loss) never get a gradient at all.  A parameter whose first gradient arrives late would be bias-corrected with the
global count; `_loaded_step` refuses checkpoints whose per-parameter counts disagree.
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
        self._t = 0          # optimizer steps taken (host mirror of the device counter hyper[0])
        self._hyper = None   # device float[3]: step, 1/(1-b1^t), 1/sqrt(1-b2^t) - maintained by nvit_adamw_tick
        self._staging = None  # pinned host image of the device table
        self._hyper_pin = None   # pinned image of the lr/weight_decay column (eager lr schedules)
        self._hyper_ev = None

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
        hypers: List[int] = []   # per row: lr | weight_decay bits (column 9 of the table)
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
                key.append((p.data_ptr(), g.data_ptr(), kind))
                hypers.append(hyper)
                first_item += items
                first_chunk += math.ceil(p.numel() / _CHUNK)
        key = tuple(key)
        if self._cache is not None and self._cache["key"] == key and self._cache["hypers"] != hypers:
            # only lr / weight_decay moved (the reference trainer rewrites lr every iteration): update the hyper column in
            # place instead of rebuilding table, workspaces and a pinned image
            self._write_hyper_column(hypers)
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
            n = len(rows)
            host = torch.tensor(rows, dtype=torch.int64)
            table = torch.empty((n, 10), dtype=torch.int64, device=dev)
            if torch.cuda.is_current_stream_capturing():
                # Inside a hipGraph capture (the gradients autograd allocates there have new addresses, so the table is
                # rebuilt once): pinned memory cannot be allocated now, so the image goes through the persistent
                # staging buffer that GraphedTrainStep reserved; the captured memcpy re-sends that buffer on every
                # replay, which is also how a later learning-rate change reaches the captured step.
                if self._staging is None or self._staging.shape[0] < n:
                    raise RuntimeError("FusedAdamW: call reserve_staging() before capturing a step")
                self._staging[:n].copy_(host)
                table.copy_(self._staging[:n], non_blocking=True)
                staged = True
            else:
                # eager: a fresh pinned image per rebuild (the host allocator keeps it alive until the async copy has
                # run; re-using one buffer would race with copies still queued behind a GPU that runs steps behind)
                table.copy_(host.pin_memory(), non_blocking=True)
                staged = False
            self._cache = {
                "key": key,
                "hypers": hypers,
                "table": table,
                "n": len(rows), "items": first_item, "chunks": first_chunk, "slab": max_slab_rows,
                "partial": torch.empty(_NPART, device=dev, dtype=torch.float32),
                "gnorm": torch.empty(1, device=dev, dtype=torch.float32),
                "betas_eps": beta_eps,
                "staged": staged,
                "n_elems": sum(r[4] * r[5] for r in rows),
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
        b1, b2, eps = c["betas_eps"]
        lib = _lib.load()
        dev = c["table"].device
        if self._hyper is None:
            self._t = self._loaded_step()
            self._hyper = torch.tensor([float(self._t), 0.0, 0.0], dtype=torch.float32).pin_memory().to(
                dev, non_blocking=True)
        # the step counter lives on the device (bias corrections are computed there), so a captured step replays
        # correctly; the host mirror only feeds state_dict()
        check(lib.nvit_adamw_tick(_p(self._hyper), b1, b2, _s()), "nvit_adamw_tick")
        self._t += 1
        clip = grad_clip is not None and grad_clip > 0.0
        if clip:
            check(lib.nvit_grad_sqnorm(_p(c["table"]), c["n"], c["chunks"], _p(c["partial"]), _NPART, _s()),
                  "nvit_grad_sqnorm")
        check(lib.nvit_adamw_renorm(_p(c["table"]), c["n"], c["items"], c["slab"], b1, b2, eps, 0.0, 0.0,
                                    _p(c["partial"]) if clip else None, _NPART if clip else 0,
                                    float(grad_clip) if clip else 0.0, _p(c["gnorm"]) if clip else None,
                                    _p(self._hyper), _s()),
              "nvit_adamw_renorm")
        return c["gnorm"] if clip else None

    # ------------------------------------------------------------------ step counter <-> torch's per-parameter state
    def _loaded_step(self) -> int:
        """Step count found in the (possibly loaded) per-parameter state; all parameters must agree."""
        t = None
        for st in self.state.values():
            if "step" in st:
                v = int(float(st["step"]))
                if t is not None and v != t:
                    raise RuntimeError("FusedAdamW: parameters with different step counts are not supported")
                t = v
        return t or 0

    def _write_hyper_column(self, hypers: List[int]) -> None:
        c = self._cache
        hcol = torch.tensor(hypers, dtype=torch.int64)
        if c["staged"]:
            torch.cuda.current_stream().synchronize()   # no replay may be reading the staging buffer while it changes
            self._staging[:c["n"], 9].copy_(hcol)
            c["table"].copy_(self._staging[:c["n"]], non_blocking=True)
        else:
            # one persistent pinned column, guarded by an event: the previous async copy must have left it
            if self._hyper_pin is None or self._hyper_pin.numel() < c["n"]:
                self._hyper_pin = torch.empty(max(c["n"], 64), dtype=torch.int64).pin_memory()
                self._hyper_ev = torch.cuda.Event()
            else:
                self._hyper_ev.synchronize()
            self._hyper_pin[:c["n"]].copy_(hcol)
            c["table"][:, 9].copy_(self._hyper_pin[:c["n"]], non_blocking=True)
            self._hyper_ev.record()
        c["hypers"] = list(hypers)

    def rewrite_hyper(self) -> None:
        """Push the param groups' current lr / weight_decay into the device table in place (same addresses, so a
        captured step picks them up on its next replay)."""
        c = self._cache
        if c is None:
            return
        col = []
        for group in self.param_groups:
            hyper = _f32_bits(group["lr"]) | (_f32_bits(group["weight_decay"]) << 32)
            if hyper >= 1 << 63:
                hyper -= 1 << 64
            for p in group["params"]:
                if p.grad is not None:
                    col.append(hyper)
        if len(col) != c["n"]:
            raise RuntimeError("FusedAdamW.rewrite_hyper: parameter set changed since the table was built")
        self._write_hyper_column(col)

    def reserve_staging(self) -> None:
        """Pinned host image for the parameter table, allocated ahead of a hipGraph capture."""
        n = sum(len(g["params"]) for g in self.param_groups)
        if self._staging is None or self._staging.shape[0] < n:
            self._staging = torch.empty((max(n, 64), 10), dtype=torch.int64).pin_memory()

    def note_replay(self, n: int = 1) -> None:
        """A captured step was replayed n times (the device counter advanced; keep the host mirror in sync)."""
        self._t += n

    def _sync_state_steps(self) -> None:
        for st in self.state.values():
            if "exp_avg" in st:
                st["step"] = torch.tensor(float(self._t), dtype=torch.float32)

    def state_dict(self):
        self._sync_state_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._cache = None
        self._hyper = None   # re-created from the loaded step count on the next step

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.step_fused(None, 0.0)
        return loss
