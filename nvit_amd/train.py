"""Post-step weight re-normalisation and the reference's train-step order.

`normalize_matrices(model)` replaces Trainer.normalize_matrices
(/root/reference/nvit/train.py:461-480): same six matrices per block, same axes, fp32 in
place — but ONE persistent HIP launch instead of ~30 torch kernels per block.

`train_step` reproduces the call sequence of the reference hot loop
(train.py:898-946,989-990): forward -> cross_entropy -> backward -> clip_grad_norm_(1.0) ->
AdamW.step -> zero_grad(set_to_none) -> normalize_matrices.  With the FusedAdamW that
`configure_optimizers` returns, clip + AdamW + renorm run as two HIP launches (optim.py);
with a plain torch optimizer the three steps run separately, same result.
"""
from __future__ import annotations

import torch

from . import ops
from .optim import FusedAdamW

_RENORM_ROWS = ("query", "key", "value", "c_fc")     # dim=1
_RENORM_COLS = ("att_c_proj", "mlp_c_proj")          # dim=0


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def normalize_matrices(model) -> None:
    m = _unwrap(model)
    if not m.config.use_nvit:
        return
    mats = []
    for blk in m.transformer.h:
        for n in ("query", "key", "value"):
            mats.append((getattr(blk, n).weight.data, 1))
        mats.append((blk.att_c_proj.weight.data, 0))
        mats.append((blk.c_fc.weight.data, 1))
        mats.append((blk.mlp_c_proj.weight.data, 0))
    dev = mats[0][0].device
    if dev.type != "cuda":
        raise RuntimeError("normalize_matrices: parameters must live on the HIP device (no CPU fallback)")
    key = tuple(w.data_ptr() for w, _ in mats)
    cache = getattr(m, "_renorm_cache", None)
    if cache is None or cache[0] != key:
        table, items = ops.renorm_table(mats, dev)
        cache = (key, table, items)
        object.__setattr__(m, "_renorm_cache", cache)
    ops.renorm_weights(cache[1], cache[2])


class CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(logits, y) (reference train.py:906) as one HIP pass that also produces the gradient."""

    @staticmethod
    def forward(ctx, logits, y):
        if logits.device.type != "cuda" or logits.dtype != torch.float32 or y.dtype != torch.int64:
            raise RuntimeError("CrossEntropyFn: fp32 logits and int64 labels on the HIP device (no CPU path)")
        logits = logits.contiguous()
        B, N = logits.shape
        rowloss = torch.empty(B, device=logits.device, dtype=torch.float32)
        loss = torch.empty(1, device=logits.device, dtype=torch.float32)
        dlogits = torch.empty_like(logits)
        ops.check(ops._lib.load().nvit_ce_loss(ops._p(logits), ops._p(y.contiguous()), ops._p(rowloss), ops._p(loss),
                                               ops._p(dlogits), B, N, ops._s()), "nvit_ce_loss")
        ctx.save_for_backward(dlogits)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None


def total_loss(config, logits: torch.Tensor, aux, y: torch.Tensor, consistency_weight: float = 0.1,
               smoothness_weight: float = 0.1) -> torch.Tensor:
    """Loss of the reference loop (train.py:906-926): CE, plus the weighted aux losses iff the Kohonen head is on
    (consistency/smoothness weights are settings.yaml `training.*`, the others come from the model config)."""
    loss = CrossEntropyFn.apply(logits, y)
    if config.use_kohonen:
        loss = (loss + consistency_weight * aux["kohonen_consistency"] + smoothness_weight * aux["kohonen_smoothness"]
                + config.local_quantization_weight * aux["local_quantization"]
                + config.global_quantization_weight * aux["global_quantization"]
                + config.reconstruction_weight * aux["reconstruction"])
    return loss


def train_step(model, optimizer, X: torch.Tensor, y: torch.Tensor, grad_clip: float = 1.0,
               sync_grads=None):
    """One optimizer step in the reference's order; returns (logits, loss, aux, grad_norm)."""
    logits, aux = model(X)
    loss = total_loss(_unwrap(model).config, logits, aux, y)
    loss.backward()
    if sync_grads is not None:
        sync_grads()
    if isinstance(optimizer, FusedAdamW):
        # clip + AdamW + normalize_matrices in two launches (nvit_grad_sqnorm, nvit_adamw_renorm)
        gnorm = optimizer.step_fused(model, grad_clip)
        if gnorm is not None:
            gnorm = gnorm[0].clone()
        optimizer.zero_grad(set_to_none=True)
    else:
        params = [p for p in _unwrap(model).parameters() if p.grad is not None]
        gnorm = torch.nn.utils.clip_grad_norm_(params, grad_clip) if grad_clip != 0.0 else None
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        normalize_matrices(model)
    # detached: a live reference to the step's autograd graph would also pin its AccumulateGrad nodes (and the stream
    # they were created on), which breaks a later hipGraph capture of the step
    return logits.detach(), loss.detach(), {k: v.detach() for k, v in aux.items()}, gnorm


class GraphedTrainStep:
    """The whole train step (forward, loss, backward, clip + AdamW + renorm) captured once as a hipGraph and replayed.

    SURVEY.md §8f F2: for C1-sized models the eager step is bound by ~600 kernel launches, not by the GPU.  Needs the
    FusedAdamW optimizer (its step counter and bias corrections live on the device, so replays stay correct), a
    model without the Kohonen head (its SOM schedule is host state), a fixed batch shape and a single process
    (the data-parallel wrapper launches RCCL work from autograd hooks and stays eager).  The learning rate is the one
    in the optimizer's param groups at capture time; `set_lr` rewrites it on the device between replays.
    """

    def __init__(self, model, optimizer, X: torch.Tensor, y: torch.Tensor, grad_clip: float = 1.0, warmup: int = 3):
        m = _unwrap(model)
        if not isinstance(optimizer, FusedAdamW):
            raise RuntimeError("GraphedTrainStep needs the FusedAdamW returned by ViT.configure_optimizers")
        if m.config.use_kohonen:
            raise RuntimeError("GraphedTrainStep: the Kohonen head keeps host-side step state; run it eagerly")
        if hasattr(model, "module"):
            raise RuntimeError("GraphedTrainStep: wrap the bare model (data-parallel steps run eagerly)")
        if X.device.type != "cuda":
            raise RuntimeError("GraphedTrainStep: inputs must live on the HIP device")
        self.model, self.optimizer, self.grad_clip = model, optimizer, grad_clip
        self.X, self.y = X.clone(), y.clone()
        side = torch.cuda.Stream()   # warm-up and capture share one stream: autograd's gradient accumulators are
        side.wait_stream(torch.cuda.current_stream())   # bound to the stream they are first used on
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):   # builds every cache (shadow/renorm tables, LDS attributes, workspaces)
                train_step(model, optimizer, self.X, self.y, grad_clip)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        optimizer.zero_grad(set_to_none=True)
        optimizer.reserve_staging()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            logits, aux = model(self.X)
            loss = total_loss(m.config, logits, aux, self.y)
            loss.backward()
            gnorm = optimizer.step_fused(model, grad_clip)
        self.logits, self.loss = logits.detach(), loss.detach()
        self.aux = {k: v.detach() for k, v in aux.items()}
        self.gnorm = gnorm
        optimizer.note_replay(-1)   # the capture pass records the step but does not execute it

    def set_lr(self, lr: float) -> None:
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        self.optimizer.rewrite_hyper()

    def __call__(self, X: torch.Tensor, y: torch.Tensor):
        """One optimizer step on (X, y); returns (logits, loss, aux, grad_norm) as static device tensors that the
        next call overwrites."""
        self.X.copy_(X, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        self.graph.replay()
        self.optimizer.note_replay()
        return self.logits, self.loss, self.aux, (self.gnorm[0] if self.gnorm is not None else None)
