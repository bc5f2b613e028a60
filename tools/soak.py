import torch, sys, time
sys.path.insert(0, ".")
from nvit_amd.config import named_config
from nvit_amd.model import ViT
from nvit_amd.weights import load_formula_weights
from nvit_amd.train import train_step, normalize_matrices
cfg = named_config("base")
m = ViT(cfg); load_formula_weights(m, cfg, perturb_scalars=False); m = m.cuda().set_precision("bf16").train(); normalize_matrices(m)
opt = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
g = torch.Generator(device="cuda").manual_seed(0)
B = 64
t0 = time.time()
for i in range(150):
    X = torch.rand(B, 3, 224, 224, device="cuda", generator=g) * 2 - 1
    y = torch.randint(0, 1000, (B,), device="cuda", generator=g)
    l, loss, aux, gn = train_step(m, opt, X, y, 1.0)
    if i % 25 == 0 or i == 149:
        torch.cuda.synchronize()
        print(i, round(loss.item(), 4), round(gn.item(), 4), "mem GB", round(torch.cuda.max_memory_allocated() / 2**30, 1), "t", round(time.time() - t0, 1), flush=True)
assert torch.isfinite(loss).item()
w = m.transformer.h[3].c_fc.weight
print("row norms", w.norm(dim=1).min().item(), w.norm(dim=1).max().item(), "sqk mean", m.transformer.h[3].sqk.mean().item())
