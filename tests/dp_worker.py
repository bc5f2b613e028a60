"""Worker of tests/test_gpu_dp.py: one rank of a 2-rank data-parallel run of the REAL model, both ranks on cuda:0,
gloo backend (the 1-GPU box has no second device for RCCL).  Started as a fresh process per rank.
usage: python tests/dp_worker.py <config> <out.json> [rccl|xgmi]   (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the env)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from nvit_amd.config import named_config
from nvit_amd.model import ViT
from nvit_amd.parallel import DataParallel
from nvit_amd.train import total_loss
from nvit_amd.weights import formula_state_dict, synthetic_batch


PRECISION = os.environ.get("NVIT_DP_TEST_PRECISION", "fp32")
BUCKET_CAP_MB = float(os.environ.get("NVIT_DP_TEST_CAP_MB", "0.25"))


def make(cfg):
    m = ViT(cfg)
    m.load_state_dict(formula_state_dict(cfg), strict=False)
    return m.to("cuda:0").set_precision(PRECISION).train()


def grads_of(m):
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def main():
    name, out_path = sys.argv[1], sys.argv[2]
    collective = sys.argv[3] if len(sys.argv) > 3 else "rccl"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="env://")
    cfg = named_config(name)
    koh = cfg.use_kohonen
    B = 8
    X, y = synthetic_batch(cfg, B)
    xs, ys = X.chunk(world)[rank].cuda(), y.chunk(world)[rank].cuda()
    m = make(cfg)
    if rank == 1:   # replicas start different: the wrapper must broadcast rank 0's parameters
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.5)
    dp = DataParallel(m, bucket_cap_mb=BUCKET_CAP_MB, collective=collective)
    opt = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    res = {"rank": rank, "config": name}

    def backward(model, xb, yb, scale=1.0):
        logits, aux = model(xb)
        (total_loss(cfg, logits, aux, yb) * scale).backward()

    # reference: single process, concatenated batch (rank 0 only; meaningful without the Kohonen head, whose SOM
    # update is rank-local by construction)
    ref = None
    if rank == 0 and not koh:
        ref = make(cfg)
        ropt = ref.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")

    worst_by_step = []
    for step in range(3):   # step 0 = discovery path, steps 1-2 = overlapped buckets with in-place gradients
        worst = 0.0
        c0 = dp.copies
        backward(dp, xs, ys)
        torch.cuda.synchronize()
        res[f"copies_step{step}"] = dp.copies - c0
        g = grads_of(m)
        res["n_grads"] = len(g)
        if ref is not None:
            backward(ref, X.cuda(), y.cuda())
            per = {n: rel(g[n], t) for n, t in grads_of(ref).items()}
            big = {n: e for n, e in per.items() if g[n].numel() > 16}      # scalars (skip_param) are reported apart: in
            worst = max(big.values())                                      # bf16 mode they are cancellation-dominated
            if step == 0:
                res["worst_params_step0"] = sorted(((e, n) for n, e in per.items()), reverse=True)[:6]
                res["scalar_err_step0"] = max([e for n, e in per.items() if g[n].numel() <= 16] or [0.0])
        worst_by_step.append(worst)
        # cross-rank equality of the reduced gradients
        flat = torch.cat([t.flatten() for t in g.values()]).cpu()
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        res[f"ranks_equal_step{step}"] = all(torch.equal(both[0], b) for b in both[1:])
        res[f"aligned_step{step}"] = all(p.grad.data_ptr() % 16 == 0 for p in m.parameters() if p.grad is not None)
        # fused clip + AdamW + renorm on the bucket-view gradients
        gn = opt.step_fused(dp, 1.0)
        opt.zero_grad(set_to_none=True)
        if ref is not None:
            ropt.step_fused(ref, 1.0)
            ropt.zero_grad(set_to_none=True)
    res["grad_err_vs_single_process"] = worst_by_step
    res["buckets"] = dp.num_buckets
    res["describe"] = dp.describe()
    if koh:
        nodes = torch.cat([m.local_kohonen.nodes.detach().flatten(), m.global_kohonen.nodes.detach().flatten()]).cpu()
        both = [torch.empty_like(nodes) for _ in range(world)]
        dist.all_gather(both, nodes)
        res["nodes_equal"] = all(torch.equal(both[0], b) for b in both[1:])
    # parameters after 3 optimizer steps: identical on both ranks, and equal to the single-process run
    flat = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    res["params_equal_across_ranks"] = all(torch.equal(both[0], b) for b in both[1:])
    if ref is not None:
        res["param_err_vs_single_process"] = max(
            (a.detach() - b.detach()).abs().max().item() for a, b in zip(m.parameters(), ref.parameters()))
    # gradient accumulation: two micro-steps, only the second communicates
    q = xs.shape[0] // 2
    with dp.no_sync():
        backward(dp, xs[:q], ys[:q], 0.5)
    backward(dp, xs[q:], ys[q:], 0.5)
    torch.cuda.synchronize()
    g = grads_of(m)
    if ref is not None:
        worst = 0.0
        # mean over ranks of (0.5*g(first half) + 0.5*g(second half)) = gradient of the mean loss over the four quarters
        for r in range(world):
            xr, yr = X.chunk(world)[r].cuda(), y.chunk(world)[r].cuda()
            backward(ref, xr[:q], yr[:q], 0.5 / world)
            backward(ref, xr[q:], yr[q:], 0.5 / world)
        for n, t in grads_of(ref).items():
            worst = max(worst, rel(g[n], t))
        res["accum_err_vs_single_process"] = worst
    dp.close()     # (direct collective: checks the device-side error word and unmaps the peers; no-op for the default)
    json.dump(res, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
