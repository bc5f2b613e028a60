#!/usr/bin/env python3
"""bench.py — images/sec of the full nViT train step on MI355X (BASELINE.json metric).

A "step" = forward -> cross-entropy -> backward -> (DP gradient all-reduce) -> clip(1.0) ->
AdamW -> zero_grad -> weight re-normalisation, on one synthetic batch resident in HBM
(reference order: /root/reference/nvit/train.py:898-946,989-990).

  python bench.py --gpus 1 --steps K --warmup W            (single GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see README/DESIGN for the fields).  The `roofline` object
is for the dominant kernel family (the NT MFMA GEMM): algorithmic FLOPs of its launches over
their HIP-event durations inside the timed region.  `cpu_baseline` times the CPU oracle
(a port of the reference semantics, fp32) on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

PEAK_BF16_TFLOPS = 2516.6   # 256 CU x 4096 FLOP/clk/CU x 2.4 GHz, dense (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def cpu_baseline(cfg_name: str, sample_batch: int, threads: int):
    """Oracle (port of the reference's CPU fp32 semantics) timed on the host cores."""
    from nvit_amd.config import named_config
    from nvit_amd.weights import formula_state_dict, load_formula_weights, synthetic_batch
    from oracle import nvit_oracle as O
    torch.set_num_threads(threads)
    cfg = named_config(cfg_name)
    p = O.make_params(formula_state_dict(cfg, perturb_scalars=False))
    O.renorm_(p, cfg)
    opt = O.make_optimizer(p)
    X, y = synthetic_batch(cfg, sample_batch)
    t0 = time.time()
    O.train_step(p, cfg, opt, X, y)
    dt = time.time() - t0
    return {"value": round(sample_batch / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"{cfg_name} (same workload), batch {sample_batch}, 1 full train step, fp32 torch-CPU oracle, "
                      f"{dt:.1f} s"}


def pmc_traffic():
    """HBM-side bytes per gemm_nt launch from the committed rocprofv3 PMC summary of this same command
    (profiles/*_pmc.json, made by tools/summarize_profile.py; PMC counters cannot be read inside the run)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))["traffic_bytes_per_launch"])
    except Exception:
        return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="base")
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo for rehearsals")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: all ranks use cuda:0 (needs --backend gloo); not a valid benchmark")
    ap.add_argument("--cpu-sample-batch", type=int, default=8)
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one hipGraph (1 GPU, no Kohonen head); the per-kernel roofline is then "
                         "taken from a short eager pass after the timed region")
    args = ap.parse_args()

    import torch.distributed as dist
    from nvit_amd import ops
    from nvit_amd.config import named_config, train_flops_per_image
    from nvit_amd.model import ViT
    from nvit_amd.train import GraphedTrainStep, normalize_matrices, train_step
    from nvit_amd.weights import formula_state_dict, load_formula_weights, synthetic_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus}"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend, init_method="env://")

    cfg = named_config(args.config)
    model = ViT(cfg)
    load_formula_weights(model, cfg, perturb_scalars=False)
    model = model.to(dev).set_precision(args.precision).train()
    normalize_matrices(model)           # steady (unit-norm) state, BASELINE.md §2
    opt = model.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
    sync = None
    if world > 1:
        from nvit_amd.parallel import DataParallel
        dp = DataParallel(model)
        sync = dp.finish
        step_model = dp
    else:
        step_model = model
    X, y = synthetic_batch(cfg, args.batch, seed=1234 + rank)
    X, y = X.to(dev), y.to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        train_step(step_model, opt, X, y, 1.0, sync_grads=sync)
    barrier()
    prof_steps = args.steps
    if args.graph:
        if world > 1:
            raise SystemExit("--graph is single-process (the data-parallel step runs eagerly)")
        graphed = GraphedTrainStep(model, opt, X, y, 1.0, warmup=1)
        l0=graphed(X, y)
        barrier()
        print("DBG first replay", l0[1].item(), file=sys.stderr)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            logits, loss, aux, gnorm = graphed(X, y)
            if os.environ.get('DBGSYNC'):
                torch.cuda.synchronize()
                bad = [(n, tuple(p.shape)) for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
                tot = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)).item()
                part = opt._cache['partial']
                print("DBG replay", loss.item(), gnorm.item(), "torch-norm", tot, "bad", bad[:4], "partial finite", torch.isfinite(part).all().item(), "nparams", sum(1 for p in model.parameters() if p.grad is not None), opt._cache['n'], file=sys.stderr)
        barrier()
        dt = time.perf_counter() - t0
        print("DBG after timed", loss.item(), file=sys.stderr)
        prof_steps = min(args.steps, 3)
        ops.prof_enable(True)
        ops.prof_collect()
        for _ in range(prof_steps):
            train_step(step_model, opt, X, y, 1.0)
        barrier()
    else:
        ops.prof_enable(True)
        ops.prof_collect()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            logits, loss, aux, gnorm = train_step(step_model, opt, X, y, 1.0, sync_grads=sync)
        barrier()
        dt = time.perf_counter() - t0
    ops.prof_enable(False)
    prof = ops.prof_collect()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    if not torch.isfinite(loss).item():
        raise SystemExit("non-finite loss in the timed region")

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        gf = train_flops_per_image(cfg) / 1e9
        g = prof["gemm_nt"]            # plain-epilogue bf16 NT GEMMs: the dominant kernel of the step
        ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        gfu = prof.get("gemm_fused", {"flops": 0.0, "ms": 0.0})   # same kernel with SwiGLU / q-k-norm / SwiGLU-bwd epilogues
        fam_ms = g["ms"] + gfu["ms"]
        fam = (g["flops"] + gfu["flops"]) / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else 0.0
        T = (cfg.image_size // cfg.local_patch_size) ** 2
        out = {
            "metric": "images/sec (train step) nViT-B/16 224px", "value": round(value, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic" if not args.share_gpu else "synthetic (REHEARSAL: ranks share one GPU, not a benchmark)",
            "config": {"workload": f"nViT-{args.config} {cfg.image_size}px patches {cfg.local_patch_size}/"
                                   f"{cfg.global_patch_size} T={T} C={cfg.n_embd} L={cfg.n_layer} H={cfg.n_head}, "
                                   f"full train step (fwd+bwd+clip+AdamW+renorm), synthetic images, formula weights",
                       "per_gpu_batch": args.batch, "global_batch": world * args.batch,
                       "parallelism": f"dp{world}", "train_gflop_per_image": round(gf, 3)},
            "step_mfma_frac": round(value * gf / 1e3 / (world * PEAK_BF16_TFLOPS), 4),
            "roofline": {"bound": "mfma", "kernel": "gemm_nt (persistent 256x256 tile, bf16 v_mfma_f32_16x16x32, LDS-DMA ring)",
                         "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic(),
                         "launches": g["launches"], "avg_launch_ms": round(g["ms"] / max(1, g["launches"]), 4),
                         "family_achieved_incl_fused_epilogues": round(fam, 1)},
            "kernel_ms_per_step": {k: round(v["ms"] / prof_steps, 3) for k, v in prof.items() if v["launches"]},
        }
        # HBM-bound kernels of the path: algorithmic bytes (as declared at each launch) / HIP-event time, vs 8 TB/s
        n_upd = opt._cache["n_elems"] if getattr(opt, "_cache", None) else sum(p.numel() for p in model.parameters())
        hbm = {}
        for fam, label in (("rowops", "row kernels (LERP/norm_skip fwd+bwd, reductions)"), ("patchify", "im2col"),
                           ("shadow", "bf16 operand copies of the weights")):
            f = prof.get(fam)
            if f and f["ms"] > 0 and f["bytes"] > 0:
                hbm[fam] = {"what": label, "GB/s": round(f["bytes"] / (f["ms"] * 1e-3) / 1e9, 1),
                            "frac_of_8TB/s": round(f["bytes"] / (f["ms"] * 1e-3) / 8e12, 3)}
        f = prof.get("renorm")
        if f and f["ms"] > 0:
            by = 32.0 * n_upd * prof_steps   # p,g,m,v read + p,m,v written + g read again for the global norm
            hbm["optimizer+renorm"] = {"what": "clip + AdamW + normalize_matrices, 32 B/parameter",
                                       "GB/s": round(by / (f["ms"] * 1e-3) / 1e9, 1),
                                       "frac_of_8TB/s": round(by / (f["ms"] * 1e-3) / 8e12, 3)}
        out["hbm_kernels"] = hbm
        if args.graph:
            out["graph"] = True
            out["roofline"]["source"] = f"{prof_steps} eager steps after the timed hipGraph replays"
        if world == 1 and not args.no_cpu_baseline:
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            threads = max(1, min(16, avail))   # one-GPU box share is 16 cores; never oversubscribe
            print(f"[bench] GPU part done ({value:.1f} img/s); timing the CPU oracle on {threads} threads ...",
                  file=sys.stderr, flush=True)
            out["cpu_baseline"] = cpu_baseline(args.config, args.cpu_sample_batch, threads)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
