"""Gradient all-reduce by direct peer reads over xGMI: reduce-scatter + all-gather kernels
(`nvit_amd/csrc/xgmi.hip`) over symmetric flat buffers that every rank maps through IPC.

Reference intent: the DDP gradient averaging of /root/reference/nvit/train.py:438-446.  An MI355X node is a fully
connected xGMI mesh (7 links per GPU); the direct form moves S/N bytes over each link per phase with all links busy at
once, instead of walking a ring.  Sums are taken in rank order by exactly one owner per element, so replicas stay
bit-identical.

What is verified: the kernels, the IPC exchange and the phase protocol, with 2-4 ranks SHARING one MI355X
(`tests/test_gpu_xgmi.py`); on a multi-GPU node the same code reads over the links (peer access must be enabled between
the devices; not measurable on the one-GPU build box, which is why `DataParallel` keeps RCCL as its default).
Phases are separated by `stream.synchronize()` + a host barrier of the process group - correct everywhere, with no
overlap with backward; a device-side flag protocol is the follow-up once it can be validated on real links.

PyTorch is plumbing here: device memory, the IPC handle exchange (`torch.multiprocessing.reductions`, the mechanism
behind CUDA tensors in torch.multiprocessing queues) and the host barrier (`torch.distributed`).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.distributed as dist

from . import _lib
from .ops import _s, check


class XgmiAllReduce:
    """All-reduce (sum * scale) of one flat fp32 buffer of `numel` elements per rank, in place.

    `self.buffer` is this rank's symmetric buffer (numel rounded up to a multiple of 4; the padding is zero and is
    reduced like everything else).  Fill it, call `all_reduce_()`, read it back."""

    def __init__(self, numel: int, device: torch.device, group=None) -> None:
        if not dist.is_initialized():
            raise RuntimeError("XgmiAllReduce needs an initialised torch.distributed process group")
        if device.type != "cuda":
            raise RuntimeError("XgmiAllReduce: HIP device buffers only (no CPU path)")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if self.world > 8:
            raise RuntimeError("XgmiAllReduce: at most 8 ranks (one node)")
        self.numel = (numel + 3) // 4 * 4
        self.buffer = torch.zeros(self.numel, device=device, dtype=torch.float32)
        # export this buffer, import everyone else's: IPC handles travel as picklable (rebuild_fn, args) pairs
        from torch.multiprocessing.reductions import reduce_tensor
        mine = reduce_tensor(self.buffer)
        handles: List[Optional[tuple]] = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
        self._peers: List[torch.Tensor] = []
        for r in range(self.world):
            if r == self.rank:
                self._peers.append(self.buffer)
            else:
                fn, args = handles[r]
                t = fn(*args)                      # opens the IPC handle: a tensor aliasing rank r's buffer
                if t.numel() != self.numel or t.dtype != torch.float32:
                    raise RuntimeError("XgmiAllReduce: ranks disagree on the buffer size")
                self._peers.append(t)
        self._ptrs = (C.c_int64 * self.world)(*[t.data_ptr() for t in self._peers])
        self.chunk = int(_lib.load().nvit_xgmi_chunk(self.numel, self.world))
        self._barrier()   # nobody may start before every rank has opened every handle

    def _barrier(self) -> None:
        torch.cuda.current_stream().synchronize()
        dist.barrier(group=self.group)

    def all_reduce_(self, scale: float = 1.0, numel: Optional[int] = None) -> torch.Tensor:
        """buffer[:numel] <- scale * sum over ranks of their buffer[:numel] (bit-identical on every rank).
        numel (default: the whole buffer) is rounded up to a multiple of 4 and must be the same on every rank."""
        lib = _lib.load()
        n = self.numel if numel is None else (int(numel) + 3) // 4 * 4
        if n <= 0 or n > self.numel:
            raise ValueError("XgmiAllReduce.all_reduce_: numel out of range")
        self._barrier()   # every rank has finished WRITING its buffer
        check(lib.nvit_xgmi_reduce_scatter(self._ptrs, self.world, self.rank, n, float(scale), _s()),
              "nvit_xgmi_reduce_scatter")
        self._barrier()   # every owner has reduced its chunk
        check(lib.nvit_xgmi_all_gather(self._ptrs, self.world, self.rank, n, _s()), "nvit_xgmi_all_gather")
        self._barrier()   # nobody reads a peer any more: the buffers may be overwritten
        return self.buffer

    def close(self) -> None:
        """Drop the peer mappings (call on every rank before the owners free their buffers)."""
        self._barrier()
        self._peers = [self.buffer]
