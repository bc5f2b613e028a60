"""Worker of tests/test_gpu_xgmi.py: one rank of an N-rank run of nvit_amd.xgmi.XgmiAllReduce, all ranks on cuda:0.
usage: python tests/xgmi_worker.py <out.json>   (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the env)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from nvit_amd.xgmi import XgmiAllReduce


def fill(rank, n, step, dev):
    i = torch.arange(n, device=dev, dtype=torch.float32)
    return torch.sin(i * (0.001 * (rank + 1))) * (1.0 + rank) + step * 0.125


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method="env://")
    res = {"rank": rank, "world": world, "cases": []}
    sizes = (4, 1000, 1 << 20, 9437185)     # incl. sizes that are not multiples of 4 * world, and one nGPT block + 1
    ar = XgmiAllReduce(max(sizes), dev)     # ONE symmetric buffer (one IPC export per process); prefixes of it are reduced
    for n in sizes:
        n4 = (n + 3) // 4 * 4
        worst, equal = 0.0, True
        for step in range(3):
            ar.buffer.zero_()
            ar.buffer[:n] = fill(rank, n, step, dev)
            ar.buffer[n4:] = 7.0                                  # outside the reduced prefix: must stay untouched
            got = ar.all_reduce_(1.0 / world, numel=n)[:n].clone()
            want = sum(fill(r, n, step, dev) for r in range(world)) / world   # same order 0..world-1 as the kernel
            worst = max(worst, (got - want).abs().max().item())
            both = [torch.empty(n) for _ in range(world)]
            dist.all_gather(both, got.cpu())
            equal = equal and all(torch.equal(both[0], b) for b in both[1:])
            pad_ok = bool((ar.buffer[n:n4] == 0).all().item()) and bool((ar.buffer[n4:] == 7.0).all().item())
        res["cases"].append({"n": n, "chunk": int(ar.chunk), "max_err": worst, "bit_identical_across_ranks": equal,
                             "padding_zero": pad_ok})
    ar.close()
    json.dump(res, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
