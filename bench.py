#!/usr/bin/env python3
"""bench.py — images/sec of the full nViT train step on MI355X (BASELINE.json metric).

A "step" = forward -> cross-entropy -> backward -> (DP gradient all-reduce) -> clip(1.0) ->
AdamW -> zero_grad -> weight re-normalisation, on one synthetic batch resident in HBM
(reference order: /root/reference/nvit/train.py:898-946,989-990).

  python bench.py --gpus 1 --steps K --warmup W            (single GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see README/DESIGN for the fields).  The `roofline` object
is for the dominant kernel family (the plain-epilogue NT MFMA GEMM): algorithmic FLOPs of its
launches over their HIP-event durations inside the timed region; `roofline.families` lists
EVERY MFMA family of the step (plain NT, each fused epilogue, weight-gradient TN, attention
forward / backward) with its own achieved TFLOP/s and fraction, so the weakest one is visible.
The peak is derived from the device (CU count x 4096 bf16 FLOP/clk/CU x max clock), both printed.
`cpu_baseline` times the CPU oracle (a port of the reference semantics, fp32) on this box's host
cores as BASELINE.md §3 prescribes: C2 (Base) at B=8, 3 steps after one warm-up, and C1 (Tiny) at
B=32, 10 steps.  `--check` runs the full-shape numeric check (tools/fullshape_check.py) and an
end-to-end comparison after the timed region.
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

NOMINAL_PEAK_BF16_TFLOPS = 2516.6   # 256 CU x 4096 FLOP/clk/CU x 2.4 GHz, dense (MI355X_MICROARCH.md)
BF16_FLOP_PER_CLK_PER_CU = 4096     # 4 SIMDs x (16x16x32 MFMA = 16384 FLOP per 16 cycles)
PEAK_HBM_GBS = 8000.0


def device_peak(dev):
    """(CU count, max clock in MHz, dense bf16 MFMA peak in TFLOP/s) from the device properties."""
    prop = torch.cuda.get_device_properties(dev)
    cus = int(prop.multi_processor_count)
    mhz = float(getattr(prop, "clock_rate", 0)) / 1e3   # kHz -> MHz
    if mhz <= 0:
        mhz = 2400.0
    return cus, mhz, cus * BF16_FLOP_PER_CLK_PER_CU * mhz * 1e6 / 1e12


def _time_oracle_steps(cfg_name: str, batch: int, steps: int, warm: int, threads: int):
    from nvit_amd.config import named_config
    from nvit_amd.weights import formula_state_dict, synthetic_batch
    from oracle import nvit_oracle as O
    torch.set_num_threads(threads)
    cfg = named_config(cfg_name)
    p = O.make_params(formula_state_dict(cfg, perturb_scalars=False))
    O.renorm_(p, cfg)
    opt = O.make_optimizer(p)
    X, y = synthetic_batch(cfg, batch)
    for _ in range(warm):
        O.train_step(p, cfg, opt, X, y)
    t0 = time.time()
    for i in range(steps):
        O.train_step(p, cfg, opt, X, y)
        print(f"[bench] cpu oracle {cfg_name} B={batch}: step {i + 1}/{steps} ({time.time() - t0:.1f} s)",
              file=sys.stderr, flush=True)
    dt = time.time() - t0
    return batch * steps / dt, dt


def cpu_baseline(cfg_name: str, sample_batch: int, threads: int):
    """Oracle (port of the reference's CPU fp32 semantics) timed on the host cores, BASELINE.md §3: the benchmarked
    config at B=8 for 3 steps after one warm-up step, and C1 (Tiny, the reference's own CPU-runnable case) at B=32
    for 10 steps after one warm-up."""
    ips, dt = _time_oracle_steps(cfg_name, sample_batch, 3, 1, threads)
    out = {"value": round(ips, 4), "unit": "images/sec", "cores": threads, "kind": "port",
           "sample": f"{cfg_name} (same workload), batch {sample_batch}, 3 full train steps after 1 warm-up, fp32 "
                     f"torch-CPU oracle, {dt:.1f} s"}
    ips1, dt1 = _time_oracle_steps("tiny", 32, 10, 1, threads)
    out["c1_tiny"] = {"value": round(ips1, 2), "unit": "images/sec",
                      "sample": f"tiny (C1), batch 32, 10 full train steps after 1 warm-up, {dt1:.1f} s"}
    return out


def pmc_traffic():
    """(HBM-side bytes per plain gemm_nt launch, per-family bytes per launch, the profile file they come from) from the
    committed rocprofv3 PMC summary of this same command (profiles/*_pmc.json, made by tools/summarize_profile.py; PMC
    counters cannot be read inside the run, so the source file is named in the JSON line and goes stale only visibly)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))
    if not files:
        return None, {}, None
    try:
        js = json.load(open(files[-1]))
        return float(js["traffic_bytes_per_launch"]), js.get("families", {}), os.path.basename(files[-1])
    except Exception:
        return None, {}, None


def run_parity(model, cfg, args, X):
    """Parity facts of the benchmarked mode that can be stated without the CPU oracle (which bench.py touches only in
    its cpu_baseline leg): logits of the first 4 images in the benchmarked precision against the exact-f32 mode of the
    same HIP model (that mode is held to 1e-5 of the CPU oracle at this size by tests/test_gpu_model.py), outside the
    timed region."""
    model.eval()
    with torch.no_grad():
        lo, _ = model(X[:4])
        model.set_precision("fp32")
        ref, _ = model(X[:4])
        model.set_precision(args.precision)
    model.train()
    d = (lo - ref).abs()
    lmax = ref.abs().max().item()
    out = {"what": f"{args.precision} logits vs the exact-f32 mode of the same model, first 4 images of the batch",
           "max_abs": float(f"{d.max().item():.3e}"), "rms": float(f"{d.double().pow(2).mean().sqrt().item():.3e}"),
           "logit_max_abs": round(lmax, 3), "max_rel_to_logit_range": float(f"{d.max().item() / lmax:.3e}"),
           "fp32_mode_vs_reference": "<= 1e-5 against golden vectors of the imported reference at this model size "
                                     "(tests/test_gpu_model.py, Base B=6: 1.7e-6; one full step later 5.2e-6)"}
    if args.precision == "bf16":
        out["north_star_tolerance"] = 1e-3
        out["meets_1e-3_vs_fp32"] = bool(d.max().item() < 1e-3)
        out["documented_deviation"] = (
            "beyond the small configs the bf16 rounding of the GEMM operands alone moves the ORACLE's logits by more than "
            "1e-3 (Base 2.7e-3, Large 4.0e-3; activations alone 1.13e-3 at Large); the primary bar is reference-held: "
            "|HIP_bf16 - ref_fp32| <= |ref_autocast_bf16 - ref_fp32| on fixtures recorded from the imported reference "
            "(tests/golden/*_autocast.npz: Base B=6 2.87e-3 vs 7.7e-3, Large B=2 4.27e-3 vs 8.9e-3, Base+Kohonen B=2 "
            "2.39e-3 vs 1.5e-2; profiles/r04_parity_margins.json); secondary: within 1e-3 of the CPU evaluation that rounds "
            "the same operands at the same points (DESIGN.md 2)")
    return out


def run_check(model, cfg, args, X, dev):
    """Correctness of the benchmarked shape itself, outside the timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fullshape_check
    T = (cfg.image_size // cfg.local_patch_size) ** 2
    rep = fullshape_check.check_all(B=args.batch, T=T, C=cfg.n_embd, H=cfg.n_head, verbose=False, dev=dev)
    bad = [k for k, v in rep.rows.items() if not v["ok"]]
    # end to end: logits of the whole batch in the benchmarked mode vs the exact-f32 mode on the first 4 images
    model.eval()
    with torch.no_grad():
        full, _ = model(X)
        model.set_precision("fp32")
        ref, _ = model(X[:4])
        model.set_precision(args.precision)
    model.train()
    e2e = (full[:4] - ref).abs().max().item()
    return {"kernels_checked": len(rep.rows), "failed": bad, "ok": not bad and e2e < 1e-2,
            "worst_rel_to_tol": round(max(v["err"] / v["tol"] for v in rep.rows.values()), 3),
            "e2e_logits_B%d_%s_vs_fp32mode_B4_max_abs" % (args.batch, args.precision): float(f"{e2e:.3e}"),
            "e2e_logit_max_abs": round(ref.abs().max().item(), 3)}


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
    ap.add_argument("--collective", default="rccl", choices=["rccl", "xgmi"],
                    help="gradient exchange: torch.distributed all-reduce (RCCL; default) or the direct xGMI collective")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: all ranks use cuda:0 (needs --backend gloo); not a valid benchmark")
    ap.add_argument("--cpu-sample-batch", type=int, default=8)
    ap.add_argument("--check", action="store_true",
                    help="after the timed region: numeric check of every GEMM shape / epilogue and of attention at this "
                         "batch size (tools/fullshape_check.py) and bf16 B=batch logits vs the exact-f32 mode")
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
        dp = DataParallel(model, collective=args.collective)
        dp.profile_exposed(True)
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
    timed_gemm_nt = None      # plain gemm_nt launches timed INSIDE the timed region (eager path)
    if args.graph:
        if world > 1:
            raise SystemExit("--graph is single-process (the data-parallel step runs eagerly)")
        graphed = GraphedTrainStep(model, opt, X, y, 1.0, warmup=1)
        graphed(X, y)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            logits, loss, aux, gnorm = graphed(X, y)
        barrier()
        dt = time.perf_counter() - t0
        prof_steps = min(args.steps, 3)
        ops.prof_enable(True)
        ops.prof_collect()
        for _ in range(prof_steps):
            train_step(step_model, opt, X, y, 1.0)
        barrier()
    else:
        # timed region: HIP events only around the launches of the dominant kernel (plain gemm_nt: what `roofline.achieved`
        # is made of); the per-family breakdown comes from `prof_steps` more steps behind it, with every launch timed
        # (an event pair per launch costs the step 0.3-1.0 ms at ~400 launches)
        ops.prof_select("gemm_nt")
        ops.prof_collect()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            logits, loss, aux, gnorm = train_step(step_model, opt, X, y, 1.0, sync_grads=sync)
        barrier()
        dt = time.perf_counter() - t0
        ops.prof_enable(False)
        timed_gemm_nt = ops.prof_collect()["gemm_nt"]
        prof_steps = min(args.steps, 3)
        ops.prof_enable(True)
        for _ in range(prof_steps):
            train_step(step_model, opt, X, y, 1.0, sync_grads=sync)
        barrier()
    ops.prof_enable(False)
    prof = ops.prof_collect()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    if not torch.isfinite(loss).item():
        raise SystemExit("non-finite loss in the timed region")
    dist_info = None
    if world > 1:
        # what the communicator saw, so that a multi-GPU run answers "did RCCL see N ranks on N devices, was the
        # all-reduce hidden behind backward" by itself
        prop = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device_index": dev.index, "name": prop.name,
                "uuid": str(getattr(prop, "uuid", "")), "pci_bus_id": getattr(prop, "pci_bus_id", None),
                "exposed_comm_ms_per_step": None}
        ex = dp.exposed_ms()
        if ex:
            tail = prof_steps if timed_gemm_nt else 0       # (the fully timed steps behind the timed region are not counted)
            ex = ex[len(ex) - tail - args.steps:len(ex) - tail]
            mine["exposed_comm_ms_per_step"] = round(sum(ex) / len(ex), 3)
        every = [None] * world
        dist.all_gather_object(every, mine)
        desc = dp.describe()
        dist_info = {"backend": desc["backend"], "world_size_seen_by_communicator": dist.get_world_size(),
                     "collective": desc["collective"], "gemm_tile_schedule": desc["gemm_tile_schedule"],
                     "buckets": desc["buckets"], "bucket_bytes": desc["bucket_bytes"],
                     "grad_bytes_per_step": desc["grad_bytes_per_step"], "gradient_copies_total": desc["copies_total"],
                     "distinct_devices": len({(e["uuid"], e["pci_bus_id"], e["device_index"]) for e in every}),
                     "exposed_comm_ms_per_step_max_over_ranks": max((e["exposed_comm_ms_per_step"] or 0.0) for e in every),
                     "exposed_comm_what": "time the compute stream waits at the end of backward for the last gradient "
                                          "collective (HIP events around the end-of-backward waits); 0 = fully hidden",
                     "ranks": every}

    # stand-alone nvit_renorm_weights (the kernel north_star singles out; the step itself uses the fused optimizer):
    # 10 warm calls, outside the timed region
    renorm_warm = None
    if rank == 0:
        for _ in range(3):
            normalize_matrices(model)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            normalize_matrices(model)
        e1.record()
        torch.cuda.synchronize()
        n_mat = sum(getattr(blk, n).weight.numel() for blk in model.transformer.h
                    for n in ("query", "key", "value", "att_c_proj", "c_fc", "mlp_c_proj"))
        t_ms = e0.elapsed_time(e1) / 10
        renorm_warm = {"what": "stand-alone nvit_renorm_weights, 10 warm calls, 8 B/element (fp32 read + write)",
                       "bytes": 8.0 * n_mat, "ms": round(t_ms, 4), "GB/s": round(8.0 * n_mat / (t_ms * 1e-3) / 1e9, 1),
                       "frac_of_8TB/s": round(8.0 * n_mat / (t_ms * 1e-3) / 8e12, 3)}

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        gf = train_flops_per_image(cfg) / 1e9
        cus, mhz, peak = device_peak(dev)
        print(f"[bench] device: {torch.cuda.get_device_name(dev)}, {cus} CUs, max clock {mhz:.0f} MHz -> dense bf16 MFMA "
              f"peak {peak:.1f} TFLOP/s (nominal {NOMINAL_PEAK_BF16_TFLOPS})", file=sys.stderr, flush=True)

        traffic, fam_traffic, traffic_src = pmc_traffic()
        if not (args.config == "base" and args.batch == 128 and args.precision == "bf16"):
            traffic, fam_traffic, traffic_src = None, {}, None   # the committed PMC passes are of the default workload only

        def fam_of(key, label, executed_mult=1.0, pmc_key=None):
            """MFMA view (algorithmic FLOPs / HIP-event time vs the dense bf16 peak) and HBM view (algorithmic bytes
            declared at the launch / the same time vs 8 TB/s) of one kernel family; `bound` names the roofline the
            family sits closer to.  `traffic_over_algorithmic`: fabric-side bytes per launch from the committed PMC
            passes over the algorithmic bytes (well above 1 = re-fetched operands)."""
            f = prof.get(key)
            if not f or f["ms"] <= 0 or f["launches"] == 0:
                return None
            a = f["flops"] / (f["ms"] * 1e-3) / 1e12
            d = {"kernel": label, "achieved": round(a, 1), "frac": round(a / peak, 4),
                 "ms_per_step": round(f["ms"] / prof_steps, 3), "launches_per_step": round(f["launches"] / prof_steps, 1),
                 "avg_launch_ms": round(f["ms"] / f["launches"], 4)}
            if executed_mult != 1.0:
                d["executed_incl_recompute"] = round(a * executed_mult, 1)
            if f["bytes"] > 0:
                gbs = f["bytes"] / (f["ms"] * 1e-3) / 1e9
                d["algorithmic_bytes_per_launch"] = round(f["bytes"] / f["launches"])
                d["hbm_GB/s"] = round(gbs, 1)
                d["hbm_frac"] = round(gbs / PEAK_HBM_GBS, 4)
                d["bound"] = "hbm" if gbs / PEAK_HBM_GBS > a / peak else "mfma"
                t = fam_traffic.get(pmc_key or key)
                if t:
                    d["traffic_bytes_per_launch"] = round(t)
                    d["traffic_over_algorithmic"] = round(t / (f["bytes"] / f["launches"]), 3)
            return d

        g = timed_gemm_nt or prof["gemm_nt"]   # plain-epilogue bf16 NT GEMMs: the dominant kernel of the step
        ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        fused_keys = ("gemm_swiglu", "gemm_qknorm", "gemm_swiglu_bwd")
        fam_ms = prof["gemm_nt"]["ms"] + sum(prof[k]["ms"] for k in fused_keys)
        fam_fl = prof["gemm_nt"]["flops"] + sum(prof[k]["flops"] for k in fused_keys)
        fam = fam_fl / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else 0.0
        families = {
            "gemm_nt_plain": fam_of("gemm_nt", "gemm_nt_persistent EPI 1/2 (bf16 / fp32 store)", pmc_key="gemm_nt_plain"),
            "gemm_nt_swiglu": fam_of("gemm_swiglu", "gemm_nt_persistent EPI 3 (c_fc + suv + SwiGLU)", pmc_key="gemm_nt_epi3"),
            "gemm_nt_qknorm": fam_of("gemm_qknorm", "gemm_nt_persistent EPI 4 (qkv + per-head normalise + sqk)", pmc_key="gemm_nt_epi4"),
            "gemm_nt_swiglu_bwd": fam_of("gemm_swiglu_bwd", "gemm_nt_persistent EPI 5 (mlp_c_proj dgrad + SwiGLU backward)", pmc_key="gemm_nt_epi5"),
            "gemm_tn_wgrad": fam_of("gemm_tn", "gemm_tn_persistent + slab_reduce (weight gradients)", pmc_key="gemm_tn"),
            "attn_fwd": fam_of("attn_fwd", "attn_fwd_mfma (4*B*H*T^2*d)"),
            # algorithmic backward = 10*B*H*T^2*d (five products); the two-kernel form executes 14 (S and dP twice)
            "attn_bwd": fam_of("attn_bwd", "attn_bwd_dq_mfma + attn_bwd_dkv_mfma (10*B*H*T^2*d algorithmic)", 1.4),
        }
        if args.precision == "bf16":
            # fused dual patch embedding: executed flops = 3 bf16 products per element (hi*hi + lo*hi + hi*lo)
            families["patch_embed"] = fam_of("patchify", "patch_embed (image gather -> LDS, split-operand MFMA, bias+pos epilogue; "
                                                         "3 x 2*M*C*(Kl+Kg) executed)")
        families = {k: v for k, v in families.items() if v}
        weakest = min(families, key=lambda k: families[k]["frac"]) if families else None
        # algorithmic bytes of an average plain gemm_nt launch: A [M,K] bf16 + B [N,K] bf16 + C [M,N] (4 B: fp32 outputs dominate)
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
            "step_mfma_frac": round(value * gf / 1e3 / (world * peak), 4),
            "device": {"name": torch.cuda.get_device_name(dev), "compute_units": cus, "max_clock_mhz": round(mhz, 1),
                       "peak_bf16_tflops_from_props": round(peak, 1), "nominal_peak_bf16_tflops": NOMINAL_PEAK_BF16_TFLOPS},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt (persistent 256x256 tile, bf16 v_mfma_f32_16x16x32, LDS-DMA ring)",
                         "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "launches": g["launches"], "avg_launch_ms": round(g["ms"] / max(1, g["launches"]), 4),
                         "family_achieved_incl_fused_epilogues": round(fam, 1),
                         "families": families, "weakest_family": weakest},
            "kernel_ms_per_step": {k: round(v["ms"] / prof_steps, 3) for k, v in prof.items() if v["launches"]},
        }
        # HBM-bound kernels of the path: algorithmic bytes (as declared at each launch) / HIP-event time, vs 8 TB/s
        n_upd = opt._cache["n_elems"] if getattr(opt, "_cache", None) else sum(p.numel() for p in model.parameters())
        hbm = {}
        for fam, label in (("rowops", "row kernels (LERP/norm_skip fwd+bwd, reductions)"),
                           ("shadow", "bf16 operand copies of the weights")):
            f = prof.get(fam)
            if f and f["ms"] > 0 and f["bytes"] > 0:
                hbm[fam] = {"what": label, "GB/s": round(f["bytes"] / (f["ms"] * 1e-3) / 1e9, 1),
                            "frac_of_8TB/s": round(f["bytes"] / (f["ms"] * 1e-3) / 8e12, 3)}
        f = prof.get("optim")
        if f and f["ms"] > 0:
            by = 32.0 * n_upd * prof_steps   # p,g,m,v read + p,m,v written + g read again for the global norm
            hbm["optimizer+renorm"] = {"what": "clip + AdamW + normalize_matrices, 32 B/parameter",
                                       "GB/s": round(by / (f["ms"] * 1e-3) / 1e9, 1),
                                       "frac_of_8TB/s": round(by / (f["ms"] * 1e-3) / 8e12, 3)}
        if renorm_warm:
            hbm["renorm_standalone"] = renorm_warm
        out["hbm_kernels"] = hbm
        # plain gemm_nt launches of the step: algorithmic bytes = A + B read once, C written once, as each launch declared
        # them (the same figure the family entry uses, so the two ratios cannot disagree)
        if g["launches"] and g["bytes"] > 0:
            alg = g["bytes"] / g["launches"]
            out["roofline"]["algorithmic_bytes"] = round(alg)
            if traffic:
                out["roofline"]["traffic_over_algorithmic"] = round(traffic / alg, 3)
        out["roofline"]["traffic_measured_in_run"] = False   # PMC counters come from the committed rocprofv3 passes named above
        # eager: HIP events around the plain gemm_nt launches only inside the timed region (`roofline.achieved`, `launches`,
        # `avg_launch_ms` are from there); families / kernel_ms_per_step from `prof_steps` fully timed steps behind it
        out["timed_with_event_profiling"] = "gemm_nt launches only" if timed_gemm_nt else False
        out["family_breakdown_from"] = f"{prof_steps} fully timed steps behind the timed region"
        if dist_info is not None:
            out["dist"] = dist_info
        if world == 1:
            out["parity"] = run_parity(model, cfg, args, X)
        if args.check:
            out["check"] = run_check(model, cfg, args, X, dev)
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
        dp.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
