"""Where does the distance between the bf16 HIP path and the bf16-operand CPU emulation come from?  (GPU box.)

Runs the fp32 oracle, four CPU emulations that round every GEMM operand to bf16 -
    E_rowmax      probabilities rounded relative to the row maximum (the textbook flash form, `O.bf16_round`)
    E_bound       rounding points of the HIP kernels (`O.KernelRounding("bound")`: probabilities relative to the score
                  bound, row sum over the rounded values, q pre-scaled before its rounding)
    E_*_acc64     the same two with every matrix product accumulated in float64
- and the HIP model in bf16 mode on the same formula weights and inputs, then prints the max / rms logit distance of
every pair and, per layer, the residual-stream distance of HIP to E_bound and of E_bound to its acc64 twin.

Reading: E_x vs E_x_acc64 differ ONLY in summation order/precision (identical rounding points); that distance is the
floor below which two correct bf16-operand evaluations of this network cannot be expected to agree (each fp32 partial
sum that moves by an ulp flips some operand roundings downstream by 2^-9 relative).  E_rowmax vs E_bound isolates the
attention rounding convention.  HIP vs E_bound is what tests/test_gpu_model.py bounds.

  python tools/parity_attribution.py [base|large|base_k|tiny] [batch]
"""
from __future__ import annotations

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from nvit_amd.config import named_config
from nvit_amd.weights import formula_state_dict, synthetic_batch
from oracle import nvit_oracle as O


def rms(t):
    return t.double().pow(2).mean().sqrt().item()


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "base"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    cfg = named_config(name)
    X, y = synthetic_batch(cfg, B)

    def oracle(lowp):
        p = O.make_params(formula_state_dict(cfg))
        O.renorm_(p, cfg)
        taps = {}
        with torch.no_grad():
            logits, _ = O.forward(p, cfg, X, lowp, taps=taps, training=True, step=1)
        return logits, taps

    t0 = time.time()
    runs = {"fp32": oracle(None),
            "E_rowmax": oracle(O.KernelRounding("rowmax")),
            "E_bound": oracle(O.KernelRounding("bound")),
            "E_rowmax_acc64": oracle(O.KernelRounding("rowmax", acc64=True)),
            "E_bound_acc64": oracle(O.KernelRounding("bound", acc64=True))}
    print(f"[{name} B={B}] five CPU runs in {time.time() - t0:.0f}s, |logit|max {runs['fp32'][0].abs().max():.3f}")

    from nvit_amd.model import ViT
    from nvit_amd.train import normalize_matrices
    m = ViT(cfg)
    m.load_state_dict(formula_state_dict(cfg), strict=False)
    m = m.to("cuda:0").set_precision("bf16").train()
    normalize_matrices(m)
    taps = {}
    object.__setattr__(m, "_taps", taps)
    with torch.no_grad():
        lb, _ = m(X.cuda())
    object.__setattr__(m, "_taps", None)
    htaps = {k: v.cpu().reshape(B, -1, v.shape[-1]) for k, v in taps.items() if k.startswith("x")}
    runs["HIP"] = (lb.cpu(), htaps)
    if cfg.use_kohonen:
        for k in ("lidx", "gidx"):
            a, b = taps[k].cpu().reshape(-1), runs["fp32"][1][k].reshape(-1)
            print(f"   BMU {k}: {(a != b).sum().item()} of {a.numel()} tokens pick a different node than the fp32 oracle; "
                  f"E_bound vs fp32 oracle: {(runs['E_bound'][1][k].reshape(-1) != b).sum().item()}")

    names = list(runs)
    print("   max|dlogit| (upper triangle) / rms (lower triangle):")
    print("   " + " " * 16 + " ".join(f"{n:>15s}" for n in names))
    for i, a in enumerate(names):
        row = []
        for j, b in enumerate(names):
            if i == j:
                row.append(" " * 14 + "-")
            elif i < j:
                row.append(f"{(runs[a][0] - runs[b][0]).abs().max().item():15.2e}")
            else:
                row.append(f"{rms(runs[a][0] - runs[b][0]):15.2e}")
        print(f"   {a:>16s}" + " ".join(row))
    L = cfg.n_layer
    print("   residual stream, max|dx| per layer (x0 = cross-attention output):")
    for tag, (a, b) in (("HIP     - E_bound      ", ("HIP", "E_bound")),
                        ("HIP     - E_rowmax     ", ("HIP", "E_rowmax")),
                        ("E_bound - E_bound_acc64", ("E_bound", "E_bound_acc64")),
                        ("E_bound - E_rowmax     ", ("E_bound", "E_rowmax")),
                        ("E_bound - fp32         ", ("E_bound", "fp32"))):
        print(f"      {tag}: " + " ".join(
            f"{(runs[a][1][f'x{i}'] - runs[b][1][f'x{i}']).abs().max().item():.1e}" for i in range(L + 1)))
    floor = (runs["E_bound"][0] - runs["E_bound_acc64"][0]).abs().max().item()
    conv = (runs["E_bound"][0] - runs["E_rowmax"][0]).abs().max().item()
    hip = (runs["HIP"][0] - runs["E_bound"][0]).abs().max().item()
    hip_r = (runs["HIP"][0] - runs["E_rowmax"][0]).abs().max().item()
    print(f"   SUMMARY {name}: HIP-E_bound {hip:.2e} | HIP-E_rowmax {hip_r:.2e} | summation-order floor {floor:.2e} | "
          f"attention rounding convention {conv:.2e} | E_bound-fp32 {(runs['E_bound'][0] - runs['fp32'][0]).abs().max().item():.2e}")


if __name__ == "__main__":
    main()
