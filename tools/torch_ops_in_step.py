import sys, os
sys.path.insert(0, os.getcwd())
import torch
from nvit_amd import ViT
from nvit_amd.config import named_config
from nvit_amd.train import normalize_matrices, train_step
from nvit_amd.weights import load_formula_weights, synthetic_batch
cfg = named_config("base")
dev = torch.device("cuda:0")
m = ViT(cfg); load_formula_weights(m, cfg, perturb_scalars=False)
m = m.to(dev).set_precision("bf16").train(); normalize_matrices(m)
opt = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
X, y = synthetic_batch(cfg, 128, seed=1); X, y = X.to(dev), y.to(dev)
for _ in range(2): train_step(m, opt, X, y, 1.0)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    train_step(m, opt, X, y, 1.0)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f"{e.key:40s} calls {e.count:4d}  device {e.device_time_total:9.1f} us  cpu {e.cpu_time_total:9.1f} us")
