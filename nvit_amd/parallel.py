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
  * `collective="xgmi"`: the same buckets reduced by the hand-written direct collective, per bucket on a side stream,
    phases separated by device-side flags (no host barrier); call `close()` before tearing the process group down;
  * after the collective, `p.grad` is re-pointed at its slice of the reduced flat bucket (no copy back);
  * gradients are PRODUCED in the buckets where possible ("gradient as bucket view"): the model's backward asks
    `module._grad_sink` for the destination of a weight gradient and the weight-gradient GEMM writes straight into
    the bucket slice, so the post-accumulate hook finds `p.grad` already in place and copies nothing.  Slices
    start on 16-byte boundaries (the fused optimizer and the norm kernel read gradients as float4);
  * the q/k/v (and cross-attention k/v) weights are laid out adjacently in GEMM row order inside their bucket, so the
    one stacked [3C, C] weight-gradient GEMM output IS the three bucket slices.
Works with any backend of torch.distributed ("nccl" = RCCL on ROCm; "gloo" for the CPU tests).
"""
from __future__ import annotations

import os
from contextlib import contextmanager
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn


_ALIGN = 4   # elements: every slice of a flat fp32 bucket starts on a 16-byte boundary


class _Done:
    """Stand-in for a collective's work handle when the bucket is reduced later as part of a bigger buffer."""

    def wait(self) -> None:
        return None


class _Bucket:
    def __init__(self, params: List[nn.Parameter]):
        self.params = params
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN   # padding elements stay zero (allocated with zeros)
        self.numel = off
        self.flat: Optional[torch.Tensor] = None
        self.pending = 0
        self.handle = None

    def slice_of(self, i: int) -> torch.Tensor:
        p = self.params[i]
        return self.flat[self.offsets[i]: self.offsets[i] + p.numel()].view_as(p)


class DataParallel(nn.Module):
    def __init__(self, module: nn.Module, process_group=None, bucket_cap_mb: float = 40.0,
                 broadcast_parameters: bool = True, collective: str = "rccl",
                 xgmi_timeout_s: Optional[float] = None) -> None:
        """collective: "rccl" (default) = bucketed torch.distributed all-reduce overlapped with backward;
        "xgmi" (EXPERIMENTAL: verified with ranks sharing one device only, never timed on links) = the hand-written direct
        reduce-scatter / all-gather over IPC-mapped buffers with device-side phase flags (nvit_amd/xgmi.py):
        all buckets live in one symmetric buffer, each bucket is reduced on a side stream as soon as it is complete."""
        super().__init__()
        if not dist.is_initialized():
            raise RuntimeError("DataParallel needs an initialised torch.distributed process group")
        if collective not in ("rccl", "xgmi"):
            raise ValueError("collective must be 'rccl' or 'xgmi' (xgmi: experimental direct collective)")
        self.module = module
        self.collective = collective
        self.xgmi_timeout_s = xgmi_timeout_s   # None: NVIT_XGMI_TIMEOUT_S or 1800 s (the reference's process-group timeout)
        self._xg = None
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_cap = int(bucket_cap_mb * 1024 * 1024)
        self._avg = self._decide_avg(next(module.parameters()).device)
        self._sync = True
        self._buckets: Optional[List[_Bucket]] = None   # built after the first backward
        self._bucket_of = {}
        self._callback_queued = False
        self._first_done = False
        self._seen = set()
        self._issued = set()
        self.copies = 0   # gradients copied into a bucket by the hook (0 per step once the gradient sink is active)
        self._exposed = None   # [(event before, event after)] around the end-of-backward waits (profile_exposed)
        if broadcast_parameters:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=0, group=process_group)
        # Kohonen head (config C5): the SOM nodes are mutated inside forward from rank-local data and the reference
        # never re-synchronises them; policy here: average the freshly updated nodes across ranks.
        if hasattr(module, "_node_sync"):
            def _sync_nodes(*tensors):
                for t in tensors:
                    self._mean_all_reduce(t, async_op=False)
            object.__setattr__(module, "_node_sync", _sync_nodes)
        self._params = [p for p in module.parameters() if p.requires_grad]
        for p in self._params:
            p.register_post_accumulate_grad_hook(self._hook)
        # The persistent GEMMs take one workgroup per CU; a collective's kernels (RCCL's, or the direct xGMI ones) that run
        # under backward hold some CUs for a while, and a statically dealt GEMM then waits for its slowest workgroup (+13 %
        # per GEMM with 32-64 CUs pinned; tools/sched_interference.py).  With dynamic tile hand-out the late
        # workgroups simply take fewer tiles (+2-4 %) at a cost of 0.8 % of an undisturbed step: the default under data
        # parallelism (NVIT_GEMM_SCHED=s keeps the static deal; the single-process default stays static).
        self.gemm_sched = os.environ.get("NVIT_GEMM_SCHED", "d" if self.world > 1 else "s")
        if next(module.parameters()).is_cuda:
            from . import _lib
            _lib.load().nvit_set_gemm_sched(1 if self.gemm_sched[:1] in ("d", "1") else 0)

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
        self._issued.discard(p)   # its gradient has been accumulated: p.grad is set from here on
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
        sl = b.slice_of(i)
        if p.grad.data_ptr() != sl.data_ptr():   # not produced in place by the gradient sink: copy in
            sl.copy_(p.grad)
            p.grad = sl
            self.copies += 1
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket) -> None:
        if self._xg is not None:
            # direct xGMI collective, per bucket, on the communication stream: it starts when the kernels that produced
            # this bucket's gradients (already enqueued on the current stream) are done, waits for the peers ON THE DEVICE,
            # and runs under the rest of backward (nvit_amd/xgmi.py; the 1/N is folded into the reduce)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._comm.wait_event(ev)
            e = self._xg.begin(b.slot)
            st = self._comm.cuda_stream
            self._xg.reduce_scatter_(b.slot, e, 1.0 / self.world, b.off, b.numel, stream=st)
            self._xg.all_gather_(b.slot, e, b.off, b.numel, stream=st)
            b.handle = _Done()
            return
        b.handle = self._mean_all_reduce(b.flat, async_op=True)

    def _decide_avg(self, device) -> bool:
        """Whether the 1/N is folded into the collective (ReduceOp.AVG) - decided ONCE, at construction, and
        COLLECTIVELY: every rank probes its communicator with a one-element AVG all-reduce and the outcomes are combined
        with a MIN all-reduce (an op every backend has), so all ranks issue the same collectives for the rest of the run
        even if one communicator rejects AVG (a per-call `except` fallback would let ranks disagree and hang)."""
        ok = 0
        if dist.get_backend(self.group) == "nccl":
            try:
                t = torch.full((1,), float(dist.get_rank(self.group) + 1), device=device)
                dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
                ok = int(abs(t.item() - (self.world + 1) / 2.0) < 1e-5)
            except (RuntimeError, ValueError):
                ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=device if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item())

    def _mean_all_reduce(self, t: torch.Tensor, async_op: bool):
        """Mean over ranks.  On RCCL the 1/N is part of the collective (ReduceOp.AVG: no extra pass over the bucket -
        a separate `div_` was 479 MB of read + write per Base step on the critical stream); gloo (CPU tests and the
        one-GPU DP tests) has no AVG, there the scale runs before the sum.  Which of the two was settled in __init__."""
        if self._avg:
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        t.div_(self.world)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _ordered_used(self) -> List[nn.Parameter]:
        """Reverse registration order (= the order backward produces gradients in), except that weights whose gradients
        come out of ONE stacked GEMM are placed adjacently in that GEMM's row order (`module._stacked_grads()`)."""
        used = [p for p in reversed(self._params) if p in self._seen]
        groups = getattr(self.module, "_stacked_grads", lambda: [])()
        pos = {p: i for i, p in enumerate(used)}
        for grp in groups:
            ids = {id(q) for q in grp}
            if not all(q in pos for q in grp):
                continue
            first = min(pos[q] for q in grp)
            rest = [q for q in used if id(q) not in ids]
            n_before = sum(1 for q in used[:first] if id(q) not in ids)
            used = rest[:n_before] + list(grp) + rest[n_before:]
            pos = {p: i for i, p in enumerate(used)}
        return used

    def _sink(self, params, shape) -> Optional[torch.Tensor]:
        """Destination for the gradient of `params` (one parameter, or several whose gradients one GEMM produces
        stacked along dim 0): a view of the flat bucket, or None when the gradient must go to a fresh tensor (buckets
        not built yet, a gradient is already being accumulated, parameters not adjacent)."""
        if self._buckets is None:
            return None
        b = self._bucket_of.get(params[0])
        # a slice is handed out once per backward pass (a weight used several times per forward - the cross-attention
        # block under the Kohonen head - produces several partial gradients; only one of them may live in the bucket)
        if b is None or any(q.grad is not None or q in self._issued for q in params):
            return None
        i0 = b.index[params[0]]
        off = b.offsets[i0]
        n = 0
        for k, q in enumerate(params):
            if self._bucket_of.get(q) is not b or b.index[q] != i0 + k or b.offsets[i0 + k] != off + n:
                return None
            n += q.numel()
        self._issued.update(params)
        return b.flat[off: off + n].view(shape)

    def _build_buckets(self) -> None:
        used = self._ordered_used()
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
        if self.collective == "xgmi" and buckets:
            from .xgmi import MAX_SLOTS, XgmiAllReduce
            if len(buckets) > MAX_SLOTS:
                raise RuntimeError(f"collective='xgmi': {len(buckets)} buckets, at most {MAX_SLOTS} (raise bucket_cap_mb)")
            self._xg = XgmiAllReduce(sum(b.numel for b in buckets), buckets[0].params[0].device, self.group,
                                     slots=len(buckets), timeout_s=self.xgmi_timeout_s)
            self._comm = torch.cuda.Stream()
        off = 0
        for k, b in enumerate(buckets):
            ref = b.params[0]
            if self._xg is not None:   # bucket = slice of the symmetric buffer every rank maps (numel is a multiple of 4)
                b.flat = self._xg.buffer[off: off + b.numel]
                b.slot, b.off = k, off
                off += b.numel
            else:
                b.flat = torch.zeros(b.numel, device=ref.device, dtype=ref.dtype)
            b.index = {p: i for i, p in enumerate(b.params)}
            for p in b.params:
                self._bucket_of[p] = b
        self._buckets = buckets
        if hasattr(self.module, "_grad_sink"):
            object.__setattr__(self.module, "_grad_sink", self._sink)

    def _end_of_backward(self) -> None:
        self._callback_queued = False
        if not self._first_done:
            # first synchronised backward: learn which parameters take part, reduce without overlap
            self._build_buckets()
            self._first_done = True
            self._late = []
            for b in self._buckets:
                for i, p in enumerate(b.params):
                    sl = b.slice_of(i)
                    sl.copy_(p.grad)
                    p.grad = sl
                self._launch(b)
        for b in self._buckets:
            if b.handle is None and b.pending != len(b.params) and b.pending != 0:
                # some gradients of this bucket were not produced this step: treat them as zeros
                for i, p in enumerate(b.params):
                    # (a gradient produced in place this step already sits in its slice: only the missing ones are
                    #  zeroed; p.grad is None for those after zero_grad(set_to_none=True))
                    if p.grad is None:
                        b.slice_of(i).zero_()
                self._launch(b)
        timed = self._exposed is not None and self._first_done and self._buckets and self._buckets[0].flat.is_cuda
        if timed:   # what the compute stream still has to wait for once backward's own kernels are done = exposed comm
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
        if self._xg is not None:
            # the optimizer (current stream) runs after the last all-gather, and the next backward may rewrite the
            # buckets only when every peer has finished reading them: both are device-side waits, no host round trip
            done = torch.cuda.Event()
            done.record(self._comm)
            torch.cuda.current_stream().wait_event(done)
            self._xg.wait_gathered([b.slot for b in self._buckets])
        for p in self._late:
            self._mean_all_reduce(p.grad, async_op=False)
        self._late = []
        for b in self._buckets:
            if b.handle is not None:
                b.handle.wait()
                b.handle = None
            b.pending = len(b.params)
        if self._xg is not None:
            self._xg.poll_error()      # one pinned host word, no synchronisation: a timed-out device-side wait raises here
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream())
            self._exposed.append((e0, e1))

    def profile_exposed(self, on: bool = True) -> None:
        """Record, per synchronised backward, the time the compute stream spends between the end of backward's own
        kernels and the completion of the last collective (two events around the end-of-backward waits): the part of the
        gradient exchange that was NOT hidden behind backward."""
        self._exposed = [] if on else None

    def exposed_ms(self) -> List[float]:
        """Per-step exposed communication time in ms (call after a device synchronise)."""
        return [a.elapsed_time(b) for a, b in (self._exposed or [])]

    def describe(self) -> dict:
        """What the communicator and the bucket layout look like from this rank (bench.py's `dist` object)."""
        bk = self._buckets or []
        return {"backend": dist.get_backend(self.group), "world_size": self.world, "rank": dist.get_rank(self.group),
                "collective": ("xgmi direct reduce-scatter + all-gather" if self.collective == "xgmi" else
                               "all_reduce(AVG)" if self._avg else "div_ + all_reduce(SUM)"),
                "gemm_tile_schedule": "dynamic" if self.gemm_sched[:1] in ("d", "1") else "static",
                "buckets": len(bk), "bucket_bytes": [int(b.numel) * 4 for b in bk],
                "grad_bytes_per_step": int(sum(b.numel for b in bk)) * 4,
                "copies_total": self.copies}

    def close(self) -> None:
        """Release the direct collective's peer mappings (every rank; no-op for RCCL)."""
        if self._xg is not None:
            self._xg.check_error()
            self._xg.close()
            self._xg = None

    def finish(self) -> None:
        """Kept for explicit callers (bench.py): the end-of-backward callback already waited."""
        return None

    @property
    def num_buckets(self) -> int:
        return 0 if self._buckets is None else len(self._buckets)
