"""Gradient all-reduce by direct peer reads over xGMI: reduce-scatter + all-gather kernels
(`nvit_amd/csrc/xgmi.hip`) over symmetric flat buffers that every rank maps through IPC.

Reference intent: the DDP gradient averaging of /root/reference/nvit/train.py:438-446.  An MI355X node is a fully
connected xGMI mesh (7 links per GPU); the direct form moves S/N bytes over each link per phase with all links busy at
once, instead of walking a ring.  Sums are taken in rank order by exactly one owner per element, so replicas stay
bit-identical.

Phases are separated ON THE DEVICE: every rank owns a flag block in uncached device memory (exported over IPC next to
the data buffer); a rank announces "my region is written" / "my chunk is reduced" / "I have gathered" by storing the
call's epoch into every rank's flag block and waits by polling its own (protocol: header of xgmi.hip).  A collective is
therefore two kernel launches on a stream - no `stream.synchronize()`, no host barrier - and several regions ("slots",
one per gradient bucket) can be in flight at once, which is what lets `DataParallel(collective="xgmi")` issue the
collective of a bucket on a side stream while backward is still producing the next one.  The only host barriers left
are at construction (nobody may start before every rank has opened every handle) and in `close()`.

EXPERIMENTAL on real links.  What is verified: kernels, IPC exchange and the flag protocol with 2-4 ranks SHARING one
MI355X (`tests/test_gpu_xgmi.py`: skewed ranks, several slots in flight, bit-identical sums); on a multi-GPU node the
same code reads and signals over the links, which the one-GPU build box cannot execute or time.  Every wait is bounded
(`timeout_s`, default 1800 s = the reference's process-group timeout, train.py:224).  A timeout is FATAL and fails closed:
the rank that gives up poisons every rank's error word, no kernel announces a phase after that (nobody gathers an
unreduced chunk), and `poll_error()` - a read of one pinned host word that the end-of-backward kernel refreshes, no
synchronisation - raises on every rank within the step (`DataParallel` calls it at the end of every backward).

PyTorch is plumbing here: device memory, the IPC handle exchange for the data buffer (`torch.multiprocessing.reductions`,
the mechanism behind CUDA tensors in torch.multiprocessing queues) and `torch.distributed` for the handle exchange.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib
from .ops import _s, check

MAX_SLOTS = 64


class XgmiAllReduce:
    """All-reduce (sum * scale) of regions of one flat fp32 buffer of `numel` elements per rank, in place.

    `self.buffer` is this rank's symmetric buffer (numel rounded up to a multiple of 4; the padding is zero and is
    reduced like everything else).  `slots`: how many regions may be in flight at once (one flag set each)."""

    def __init__(self, numel: int, device: torch.device, group=None, slots: int = 1,
                 timeout_s: Optional[float] = None) -> None:
        if not dist.is_initialized():
            raise RuntimeError("XgmiAllReduce needs an initialised torch.distributed process group")
        if device.type != "cuda":
            raise RuntimeError("XgmiAllReduce: HIP device buffers only (no CPU path)")
        if not 1 <= slots <= MAX_SLOTS:
            raise ValueError(f"XgmiAllReduce: 1..{MAX_SLOTS} slots")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if self.world > 8:
            raise RuntimeError("XgmiAllReduce: at most 8 ranks (one node)")
        self.device = device
        self.slots = slots
        self.numel = (numel + 3) // 4 * 4
        self.buffer = torch.zeros(self.numel, device=device, dtype=torch.float32)
        lib = _lib.load()
        if timeout_s is None:
            timeout_s = float(os.environ.get("NVIT_XGMI_TIMEOUT_S", "1800"))
        self.timeout_s = float(timeout_s)
        check(lib.nvit_xgmi_set_timeout(self.timeout_s), "nvit_xgmi_set_timeout")
        hp, dp = C.c_void_p(), C.c_void_p()
        check(lib.nvit_xgmi_errword_alloc(C.byref(hp), C.byref(dp)), "nvit_xgmi_errword_alloc")
        self._err_host, self._err_dev = hp.value, dp.value
        self._err_view = C.cast(self._err_host, C.POINTER(C.c_uint))
        # ---- which device is every rank on?  Ranks on different devices need peer access (xGMI / PCIe P2P).
        idx = device.index if device.index is not None else torch.cuda.current_device()
        props = torch.cuda.get_device_properties(idx)
        ident = (str(getattr(props, "uuid", "")), getattr(props, "pci_bus_id", None), getattr(props, "pci_device_id", None))
        idents: List[Optional[tuple]] = [None] * self.world
        dist.all_gather_object(idents, ident, group=group)
        self.shared_device = all(i == idents[0] for i in idents)
        if not self.shared_device and torch.cuda.device_count() > 1:
            # (ranks that each see ONE device through HIP_VISIBLE_DEVICES cannot query their peers here; the IPC open
            #  below then succeeds or fails on its own and says so)
            for d in range(torch.cuda.device_count()):
                if d != idx and not torch.cuda.can_device_access_peer(idx, d):
                    raise RuntimeError(f"XgmiAllReduce (experimental): device {idx} cannot access peer device {d}; the "
                                       "direct collective needs peer access between all ranks' devices - use "
                                       "collective='rccl'")
        # ---- data buffer: IPC handles travel as picklable (rebuild_fn, args) pairs
        from torch.multiprocessing.reductions import reduce_tensor
        mine = reduce_tensor(self.buffer)
        # ---- flag block: uncached device memory from the library, raw 64-byte IPC handle
        own = C.c_void_p()
        hbuf = (C.c_char * 64)()
        check(lib.nvit_xgmi_flags_alloc(slots, C.byref(own), hbuf), "nvit_xgmi_flags_alloc")
        self._own_flags = own.value
        handles: List[Optional[tuple]] = [None] * self.world
        dist.all_gather_object(handles, (mine, bytes(hbuf.raw)), group=group)
        self._peers: List[torch.Tensor] = []
        self._flag_ptrs: List[int] = []
        for r in range(self.world):
            if r == self.rank:
                self._peers.append(self.buffer)
                self._flag_ptrs.append(self._own_flags)
                continue
            (fn, args), fh = handles[r]
            t = fn(*args)                      # opens the IPC handle: a tensor aliasing rank r's buffer
            if t.numel() != self.numel or t.dtype != torch.float32:
                raise RuntimeError("XgmiAllReduce: ranks disagree on the buffer size")
            self._peers.append(t)
            p = C.c_void_p()
            check(lib.nvit_xgmi_flags_open(C.create_string_buffer(fh, 64), C.byref(p)), "nvit_xgmi_flags_open")
            self._flag_ptrs.append(p.value)
        self._ptrs = (C.c_int64 * self.world)(*[t.data_ptr() for t in self._peers])
        self._fptrs = (C.c_int64 * self.world)(*self._flag_ptrs)
        self.chunk = int(lib.nvit_xgmi_chunk(self.numel, self.world))
        self._epoch = [0] * slots
        self._closed = False
        self._host_barrier()   # nobody may start before every rank has opened every handle

    # ------------------------------------------------------------------ host-side helpers
    def _host_barrier(self) -> None:
        torch.cuda.current_stream().synchronize()
        dist.barrier(group=self.group)

    def _region(self, off: int, numel: Optional[int]):
        off = int(off)
        n = self.numel - off if numel is None else (int(numel) + 3) // 4 * 4
        if off < 0 or off % 4 or n <= 0 or off + n > self.numel:
            raise ValueError("XgmiAllReduce: region out of range (offset and length are multiples of 4 elements)")
        return off, n

    # ------------------------------------------------------------------ the collective, phase by phase
    def begin(self, slot: int) -> int:
        """Start call number epoch+1 on `slot` (every rank makes the same calls in the same order per slot)."""
        self._epoch[slot] += 1
        return self._epoch[slot]

    def reduce_scatter_(self, slot: int, epoch: int, scale: float = 1.0, off: int = 0, numel: Optional[int] = None,
                        stream=None) -> None:
        off, n = self._region(off, numel)
        check(_lib.load().nvit_xgmi_reduce_scatter_sync(self._ptrs, self._fptrs, self.world, self.rank, self.slots, slot,
                                                        epoch, off, n, float(scale), _s() if stream is None else stream),
              "nvit_xgmi_reduce_scatter_sync")

    def all_gather_(self, slot: int, epoch: int, off: int = 0, numel: Optional[int] = None, stream=None) -> None:
        off, n = self._region(off, numel)
        check(_lib.load().nvit_xgmi_all_gather_sync(self._ptrs, self._fptrs, self.world, self.rank, self.slots, slot, epoch,
                                                    off, n, _s() if stream is None else stream),
              "nvit_xgmi_all_gather_sync")

    def wait_gathered(self, slots: Sequence[int], stream=None) -> None:
        """Enqueue (on the current stream, or `stream`) the wait for every peer to have finished reading this rank's
        regions of `slots` in their latest call: after it the regions may be rewritten."""
        slots = [s for s in slots if self._epoch[s] > 0]
        if not slots:
            return
        # slots and epochs travel BY VALUE in the kernel arguments: no device array, no pageable host-to-device copy (which
        # torch follows with a stream synchronise), nothing to keep alive, capturable in a hipGraph
        h_slots = (C.c_int * len(slots))(*slots)
        h_epochs = (C.c_uint * len(slots))(*[self._epoch[s] & 0xFFFFFFFF for s in slots])
        check(_lib.load().nvit_xgmi_wait_gathered(self._fptrs, self.world, self.rank, self.slots, h_slots, h_epochs,
                                                  len(slots), self._err_dev, _s() if stream is None else stream),
              "nvit_xgmi_wait_gathered")

    def all_reduce_(self, scale: float = 1.0, numel: Optional[int] = None, slot: int = 0, off: int = 0) -> torch.Tensor:
        """buffer[off : off + numel] <- scale * sum over ranks (bit-identical on every rank), stream-ordered on the current
        stream: safe to read and to rewrite the region in later work of that stream.  Three kernel launches, no host
        synchronisation and no host-to-device copy."""
        e = self.begin(slot)
        self.reduce_scatter_(slot, e, scale, off, numel)
        self.all_gather_(slot, e, off, numel)
        self.wait_gathered([slot])
        return self.buffer

    def _raise(self, w: int) -> None:
        what = {1: "reduce-scatter waiting for the peers' gradients", 2: "all-gather waiting for the peers' reduced chunks",
                3: "waiting for the peers to finish reading"}.get(w >> 8, "?")
        raise RuntimeError(f"XgmiAllReduce (rank {self.rank}): a device-side wait timed out after {self.timeout_s:g} s on some "
                           f"rank ({what}, slot {(w & 0xff) - 1}); the collective failed closed - gradients of this and "
                           "later steps were NOT reduced, the replicas must not continue")

    def poll_error(self) -> None:
        """Non-blocking: raise if the pinned host copy of the error word (refreshed by every wait_gathered kernel) is set.
        Lags the device by at most the kernels still in flight, i.e. a failure surfaces within a step."""
        w = int(self._err_view[0])
        if w:
            self._raise(w)

    def check_error(self) -> None:
        """Synchronise the current stream and raise if any device-side wait timed out on any rank."""
        w = C.c_uint(0)
        check(_lib.load().nvit_xgmi_flags_error(self._own_flags, self.slots, C.byref(w), _s()), "nvit_xgmi_flags_error")
        if w.value:
            self._raise(w.value)

    def close(self) -> None:
        """Drop the peer mappings (every rank, before the owners free their buffers)."""
        if self._closed:
            return
        self._closed = True
        self._host_barrier()     # every rank has finished its device work on the mapped buffers
        lib = _lib.load()
        for r, p in enumerate(self._flag_ptrs):
            if r != self.rank:
                check(lib.nvit_xgmi_flags_close(p), "nvit_xgmi_flags_close")
        self._peers = [self.buffer]
        self._host_barrier()     # nobody still maps this rank's flag block
        check(lib.nvit_xgmi_flags_free(self._own_flags), "nvit_xgmi_flags_free")
        self._own_flags = None
        check(lib.nvit_xgmi_errword_free(self._err_host), "nvit_xgmi_errword_free")
        self._err_host = self._err_dev = self._err_view = None
