"""Where does the bf16-operand error of the logits come from?  CPU-only experiment on the oracle.

Runs the fp32 oracle forward, then re-runs it with bf16 rounding applied to ONE operand family at a
time (weights only / activations only / both) and prints max|dlogit| against the fp32 run.  Used to
decide which GEMMs of the HIP path get split-operand (hi+lo) treatment.

  python tools/bf16_budget.py [config] [batch]
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

ACTIVE = set()      # families whose operands are rounded
SIDE = "both"       # "w", "x", "both"
fam_of = {}

_linear = O.linear
_attend = O.attend


def linear(x, w, b, lowp):
    fam = fam_of.get(id(w))
    if lowp is None or fam not in ACTIVE:
        return _linear(x, w, b, None)
    xx = lowp(x) if SIDE in ("x", "both") else x
    ww = lowp(w) if SIDE in ("w", "both") else w
    y = xx @ ww.t()
    return y if b is None else y + b


def attend(q, k, v, s_eff, H, lowp):
    return _attend(q, k, v, s_eff, H, lowp if "attn" in ACTIVE else None)


O.linear = linear
O.attend = attend


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "base"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    torch.set_num_threads(8)
    cfg = named_config(name)
    X, y = synthetic_batch(cfg, B)
    p = O.make_params(formula_state_dict(cfg))
    O.renorm_(p, cfg)
    for n, t in p.items():
        if n.startswith("cross_attention."):
            fam_of[id(t)] = "cross"
        elif n.endswith(("query.weight", "key.weight", "value.weight")):
            fam_of[id(t)] = "qkv"
        elif n.endswith("att_c_proj.weight"):
            fam_of[id(t)] = "o"
        elif n.endswith("c_fc.weight"):
            fam_of[id(t)] = "fc"
        elif n.endswith("mlp_c_proj.weight"):
            fam_of[id(t)] = "p"
    with torch.no_grad():
        t0 = time.time()
        ref, _ = O.forward(p, cfg, X, None, training=False)
        print(f"{name} B={B}: fp32 forward {time.time() - t0:.1f}s, |logit|max {ref.abs().max():.3f}")
        fams = ["qkv", "attn", "o", "fc", "p", "cross"]
        for side in ("both", "w", "x"):
            global SIDE
            SIDE = side
            for act in [set(fams)] + [{f} for f in fams]:
                if side != "both" and act == {"attn"}:
                    continue
                ACTIVE.clear()
                ACTIVE.update(act)
                l, _ = O.forward(p, cfg, X, O.bf16_round, training=False)
                tag = "ALL" if len(act) > 1 else next(iter(act))
                print(f"  side={side:4s} round {tag:6s}: max|dlogit| {(l - ref).abs().max().item():.3e}")


if __name__ == "__main__":
    main()
