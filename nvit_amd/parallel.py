"""Data parallelism for the nViT train step: one process per GPU, RCCL all-reduce over xGMI.

Replaces the reference's DistributedDataParallel wrapping (/root/reference/nvit/train.py:438-446,
`no_sync()` at :899-902).  The reference's own loop never actually all-reduces (it calls the
un-wrapped module); this wrapper implements the intended semantics: every
rank ends backward with the MEAN of the per-rank gradients.

Design (MI355X: 8 GPUs, fully connected xGMI mesh, 7 links/GPU):
  * gradients are packed into flat fp32 buckets in reverse registration order (= the order in which
    backward produces them: head, blocks L-1..0, cross-attention, patch embedding); one nGPT block
    (9.4 M params, 37.7 MB) fills one bucket, so each all-reduce is large enough to run at link
    bandwidth and there are only ~14 collectives per step for Base;
  * a bucket's all-reduce is launched (async, on RCCL's own stream) as soon as its last gradient has
    been accumulated, so communication overlaps the remaining backward kernels;
  * parameters that never receive a gradient (rmsnorm_* weights, reconstruction head without the
    Kohonen loss) are detected on the first backward and left out of the buckets;
  * `no_sync()` suppresses communication for gradient-accumulation micro-steps;
  * after the collective, `p.grad` is re-pointed at its slice of the reduced flat bucket (no copy back).
Works with any backend of torch.distributed ("nccl" = RCCL on ROCm; "gloo" for the CPU tests).
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn


class _Bucket:
    def __init__(self, params: List[nn.Parameter]):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += p.numel()
        self.flat: Optional[torch.Tensor] = None
        self.pending = 0
        self.handle = None


class DataParallel(nn.Module):
    def __init__(self, module: nn.Module, process_group=None, bucket_cap_mb: float = 40.0,
                 broadcast_parameters: bool = True) -> None:
        super().__init__()
        if not dist.is_initialized():
            raise RuntimeError("DataParallel needs an initialised torch.distributed process group")
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_cap = int(bucket_cap_mb * 1024 * 1024)
        self._sync = True
        self._buckets: Optional[List[_Bucket]] = None   # built after the first backward
        self._bucket_of = {}
        self._callback_queued = False
        self._first_done = False
        self._seen = set()
        if broadcast_parameters:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=0, group=process_group)
        # Kohonen head (config C5): the SOM nodes are mutated inside forward from rank-local data and the reference
        # never re-synchronises them; policy here: average the freshly updated nodes across ranks.
        if hasattr(module, "_node_sync"):
            def _sync_nodes(*tensors):
                for t in tensors:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
                    t.div_(self.world)
            object.__setattr__(module, "_node_sync", _sync_nodes)
        self._params = [p for p in module.parameters() if p.requires_grad]
        for p in self._params:
            p.register_post_accumulate_grad_hook(self._hook)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    @contextmanager
    def no_sync(self):
        old = self._sync
        self._sync = False
        try:
            yield
        finally:
            self._sync = old

    # ------------------------------------------------------------------ internals
    def _queue_callback(self) -> None:
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def _hook(self, p: nn.Parameter) -> None:
        if not self._sync:
            return
        self._queue_callback()
        if not self._first_done:
            self._seen.add(p)
            return
        b = self._bucket_of.get(p)
        if b is None:  # a parameter that had no gradient on the first step: reduce it on its own
            self._late.append(p)
            return
        i = b.index[p]
        sl = b.flat[b.offsets[i]: b.offsets[i] + p.numel()].view_as(p)
        sl.copy_(p.grad)
        p.grad = sl
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket) -> None:
        b.flat.div_(self.world)
        b.handle = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _build_buckets(self) -> None:
        used = [p for p in reversed(self._params) if p in self._seen]
        buckets: List[_Bucket] = []
        cur: List[nn.Parameter] = []
        size = 0
        for p in used:
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > self.bucket_cap:
                buckets.append(_Bucket(cur))
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            buckets.append(_Bucket(cur))
        for b in buckets:
            ref = b.params[0]
            b.flat = torch.zeros(b.numel, device=ref.device, dtype=ref.dtype)
            b.index = {p: i for i, p in enumerate(b.params)}
            for p in b.params:
                self._bucket_of[p] = b
        self._buckets = buckets

    def _end_of_backward(self) -> None:
        self._callback_queued = False
        if not self._first_done:
            # first synchronised backward: learn which parameters take part, reduce without overlap
            self._build_buckets()
            self._first_done = True
            self._late = []
            for b in self._buckets:
                for i, p in enumerate(b.params):
                    sl = b.flat[b.offsets[i]: b.offsets[i] + p.numel()].view_as(p)
                    sl.copy_(p.grad)
                    p.grad = sl
                self._launch(b)
        for b in self._buckets:
            if b.handle is None and b.pending != len(b.params) and b.pending != 0:
                # some gradients of this bucket were not produced this step: treat them as zeros
                for i, p in enumerate(b.params):
                    if p.grad is None or p.grad.data_ptr() != b.flat[b.offsets[i]:].data_ptr():
                        b.flat[b.offsets[i]: b.offsets[i] + p.numel()].zero_()
                self._launch(b)
        for p in self._late:
            p.grad.div_(self.world)
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
        self._late = []
        for b in self._buckets:
            if b.handle is not None:
                b.handle.wait()
                b.handle = None
            b.pending = len(b.params)

    def finish(self) -> None:
        """Kept for explicit callers (bench.py): the end-of-backward callback already waited."""
        return None

    @property
    def num_buckets(self) -> int:
        return 0 if self._buckets is None else len(self._buckets)
