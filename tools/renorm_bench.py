"""Stand-alone nvit_renorm_weights (Trainer.normalize_matrices as one launch), warm, Base and Large weight sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops
dev = torch.device("cuda:0")
for name, C, L in (("base", 768, 12), ("large", 1024, 24)):
    mats = []
    for _ in range(L):
        for shape, dim in (((C, C), 1), ((C, C), 1), ((C, C), 1), ((C, C), 0), ((8 * C, C), 1), ((C, 4 * C), 0)):
            mats.append((torch.randn(*shape, device=dev), dim))
    table, items = ops.renorm_table(mats, dev)
    n = sum(w.numel() for w, _ in mats)
    for sub, sel in (("all", None), ("rows only (dim=1)", 1), ("columns only (dim=0)", 0)):
        ms_ = [m for m in mats if sel is None or m[1] == sel]
        t, it = ops.renorm_table(ms_, dev)
        nn = sum(w.numel() for w, _ in ms_)
        for _ in range(3): ops.renorm_weights(t, it)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.renorm_weights(t, it)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:5s} {sub:22s}: {nn * 8 / 1e6:8.1f} MB  {ms * 1e3:7.1f} us  {nn * 8 / ms / 1e9:7.2f} TB/s")
    w, d = mats[3]
    assert (w.norm(dim=0) - 1).abs().max().item() < 1e-5 and (mats[0][0].norm(dim=1) - 1).abs().max().item() < 1e-5
