"""Worker of tests/test_gpu_xgmi.py: one rank of an N-rank run of nvit_amd.xgmi.XgmiAllReduce, all ranks on cuda:0.
usage: python tests/xgmi_worker.py <out.json>   (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the env)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from nvit_amd.xgmi import XgmiAllReduce


def fill(rank, n, step, dev):
    i = torch.arange(n, device=dev, dtype=torch.float32)
    return torch.sin(i * (0.001 * (rank + 1))) * (1.0 + rank) + step * 0.125


def late_rank(out_path, rank, world, dev):
    """A rank that arrives later than the timeout (the last rank sleeps 3x the 1 s bound before its collective): the
    waiting ranks must give up, EVERY rank must see the failure (also the late one, whose kernels find the poisoned
    error word), no rank may end up with a partially reduced or gathered buffer, and nothing may hang."""
    n = 1 << 20
    ar = XgmiAllReduce(n, dev, slots=2, timeout_s=1.0)
    ar.buffer[:] = fill(rank, n, 0, dev)
    before = ar.buffer.clone()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.time()
    if rank == world - 1:
        time.sleep(3.0)
    res = {"rank": rank, "raised": False}
    try:
        ar.all_reduce_(1.0 / world, numel=n)
        torch.cuda.synchronize()
        ar.poll_error()           # the pinned host word, as DataParallel reads it at the end of backward
    except RuntimeError as e:
        res["raised"], res["message"] = True, str(e)
    res["seconds"] = time.time() - t0
    res["buffer_untouched"] = bool(torch.equal(ar.buffer, before))
    try:                          # a second call on the poisoned communicator returns at once and raises again
        t1 = time.time()
        ar.all_reduce_(1.0 / world, numel=n, slot=1)
        torch.cuda.synchronize()
        ar.poll_error()
        res["second_raised"] = False
    except RuntimeError:
        res["second_raised"] = True
    res["second_seconds"] = time.time() - t1
    try:
        ar.check_error()
        res["check_error_raised"] = False
    except RuntimeError:
        res["check_error_raised"] = True
    ar.close()
    json.dump(res, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method="env://")
    if len(sys.argv) > 2 and sys.argv[2] == "late":
        return late_rank(out_path, rank, world, dev)
    res = {"rank": rank, "world": world, "cases": []}
    sizes = (4, 1000, 1 << 20, 9437185)     # incl. sizes that are not multiples of 4 * world, and one nGPT block + 1
    ar = XgmiAllReduce(max(sizes), dev, slots=4)   # ONE symmetric buffer (one IPC export per process); prefixes of it are reduced
    res["shared_device"] = bool(ar.shared_device)
    for n in sizes:
        n4 = (n + 3) // 4 * 4
        worst, equal = 0.0, True
        for step in range(3):
            ar.buffer.zero_()
            ar.buffer[:n] = fill(rank, n, step, dev)
            ar.buffer[n4:] = 7.0                                  # outside the reduced prefix: must stay untouched
            if step == 1:
                time.sleep(0.05 * rank)                           # ranks arrive at different times: the flags must absorb it
            got = ar.all_reduce_(1.0 / world, numel=n)[:n].clone()   # (no host barrier inside: device-side flags only)
            want = sum(fill(r, n, step, dev) for r in range(world)) / world   # same order 0..world-1 as the kernel
            worst = max(worst, (got - want).abs().max().item())
            both = [torch.empty(n) for _ in range(world)]
            dist.all_gather(both, got.cpu())
            equal = equal and all(torch.equal(both[0], b) for b in both[1:])
            pad_ok = bool((ar.buffer[n:n4] == 0).all().item()) and bool((ar.buffer[n4:] == 7.0).all().item())
        res["cases"].append({"n": n, "chunk": int(ar.chunk), "max_err": worst, "bit_identical_across_ranks": equal,
                             "padding_zero": pad_ok})
    # several regions in flight: four "buckets" (slots) of different sizes; ALL reduce-scatters are enqueued before ANY
    # all-gather, on two streams, with skewed ranks and no host synchronisation in between - a bucket's reduce-scatter
    # starts (and finishes) while earlier buckets' all-gathers have not even been enqueued
    regions = [(0, 262144), (262144, 1000), (263144, 4), (263148, 3000000)]   # (offset, numel), offsets multiples of 4
    s2 = torch.cuda.Stream()
    ok_multi, worst = True, 0.0
    for step in range(3):
        ar.buffer.zero_()
        for k, (off, n) in enumerate(regions):
            ar.buffer[off:off + n] = fill(rank, n, 10 * step + k, dev)
        torch.cuda.synchronize()
        time.sleep(0.03 * ((rank + step) % world))
        eps = [ar.begin(k) for k in range(len(regions))]
        cur = torch.cuda.current_stream()
        for k, (off, n) in enumerate(regions):
            st = (s2 if k % 2 else cur).cuda_stream
            ar.reduce_scatter_(k, eps[k], 1.0 / world, off, n, stream=st)
        for k, (off, n) in reversed(list(enumerate(regions))):
            st = (s2 if k % 2 else cur).cuda_stream
            ar.all_gather_(k, eps[k], off, n, stream=st)
        cur.wait_stream(s2)
        ar.wait_gathered(range(len(regions)))
        for k, (off, n) in enumerate(regions):
            got = ar.buffer[off:off + n].clone()
            want = sum(fill(r, n, 10 * step + k, dev) for r in range(world)) / world
            worst = max(worst, (got - want).abs().max().item())
            both = [torch.empty(n) for _ in range(world)]
            dist.all_gather(both, got.cpu())
            ok_multi = ok_multi and all(torch.equal(both[0], b) for b in both[1:])
    res["multi_slot"] = {"max_err": worst, "bit_identical_across_ranks": ok_multi}
    ar.check_error()
    ar.close()
    json.dump(res, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
