"""N>1 path on CPU: nvit_amd.parallel.DataParallel with the gloo backend, world_size 2.

Checks the parity target of SURVEY.md §8e / §9.1-Q4: after backward every rank holds the gradient
of the single-process run on the concatenated batch (mean of per-rank gradients), parameters that
never receive gradients are skipped, and no_sync() defers communication across micro-steps."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


class _StackedLinear(torch.autograd.Function):
    """y = x @ cat(w1, w2)^T with the two weight gradients produced by ONE stacked product, written into the
    destination the data-parallel wrapper offers (the pattern of the q/k/v weight-gradient GEMM in nvit_amd/model.py)."""

    @staticmethod
    def forward(ctx, x, w1, w2, owner):
        ctx.save_for_backward(x, w1, w2)
        ctx.par, ctx.owner = (w1, w2), owner
        return x @ torch.cat((w1, w2), dim=0).t()

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2 = ctx.saved_tensors
        n1 = w1.shape[0]
        sink = ctx.owner._grad_sink
        shape = (w1.shape[0] + w2.shape[0], w1.shape[1])
        g = sink(ctx.par, shape) if sink is not None else None
        if g is None:
            g = torch.empty(shape)
        else:
            ctx.owner.sunk += 1
        torch.mm(dy.t(), x, out=g)
        return dy @ torch.cat((w1, w2), dim=0), g[:n1], g[n1:], None


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Linear(16, 64)
        self.unused = nn.Parameter(torch.ones(7))          # never receives a gradient (like rmsnorm_*)
        self.odd = nn.Parameter(torch.ones(1))             # 1-element parameter (like skip_param): slices must stay aligned
        self.w2 = nn.Parameter(torch.randn(20, 64) * 0.1)  # registered BEFORE w1: the wrapper must re-order the pair
        self.w1 = nn.Parameter(torch.randn(12, 64) * 0.1)
        self.b = nn.Linear(32, 300)
        self.c = nn.Linear(300, 5)
        self._grad_sink = None
        self.sunk = 0

    def _stacked_grads(self):
        return [(self.w1, self.w2)]

    def forward(self, x):
        h = torch.tanh(self.a(x)) * self.odd
        h = _StackedLinear.apply(h, self.w1, self.w2, self)
        return self.c(torch.tanh(self.b(torch.tanh(h))))


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(8, 16, generator=g), torch.randint(0, 5, (8,), generator=g)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nvit_amd.parallel import DataParallel
        torch.set_num_threads(1)
        net = Net()
        if rank == 1:  # replicas start different: construction must broadcast rank 0's parameters
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(1.0)
        dp = DataParallel(net, bucket_cap_mb=0.02)  # tiny cap -> several buckets
        X, y = _data()
        xs, ys = X.chunk(world)[rank], y.chunk(world)[rank]
        res = {}
        for step in range(3):  # step 0 = discovery path, steps 1.. = overlapped bucket path
            for p in net.parameters():
                p.grad = None
            nn.functional.cross_entropy(dp(xs), ys).backward()
            res[f"step{step}"] = {n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()}
        res["buckets"] = dp.num_buckets
        res["sunk"] = net.sunk
        res["copies"] = dp.copies
        res["aligned"] = all(p.grad.data_ptr() % 16 == 0 for p in net.parameters() if p.grad is not None)
        res["w_adjacent"] = net.w2.grad.data_ptr() == net.w1.grad.data_ptr() + net.w1.numel() * 4
        # gradient accumulation: 2 micro-steps, only the last one communicates
        for p in net.parameters():
            p.grad = None
        halves = xs.chunk(2), ys.chunk(2)
        with dp.no_sync():
            (nn.functional.cross_entropy(dp(halves[0][0]), halves[1][0]) / 2).backward()
        (nn.functional.cross_entropy(dp(halves[0][1]), halves[1][1]) / 2).backward()
        res["accum"] = {n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()}
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_dataparallel_gloo_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    ref = Net()
    X, y = _data()
    nn.functional.cross_entropy(ref(X), y).backward()
    want = {n: p.grad for n, p in ref.named_parameters()}
    for rank in range(world):
        res = out[rank]
        assert res["buckets"] >= 3
        assert res["aligned"] and res["w_adjacent"]
        assert res["sunk"] == 2      # steps 1 and 2: the stacked gradient was produced inside the bucket ...
        # ... so per overlapped step only the 7 parameters that torch's own backward produces are copied
        assert res["copies"] == 2 * 7, res["copies"]
        for key in ("step0", "step1", "step2", "accum"):
            for n, g in res[key].items():
                if n == "unused":
                    assert g is None
                    continue
                assert torch.allclose(g, want[n], atol=1e-6, rtol=1e-5), (rank, key, n)
    # both ranks bit-identical
    for n in want:
        if n != "unused":
            assert torch.equal(out[0]["step2"][n], out[1]["step2"][n])
