import torch, sys
sys.path.insert(0, ".")
from nvit_amd.config import named_config
from nvit_amd.model import ViT
from nvit_amd.weights import load_formula_weights, synthetic_batch
from nvit_amd.train import train_step, normalize_matrices, GraphedTrainStep
name = sys.argv[1]; pre = int(sys.argv[2])
cfg = named_config(name)
m = ViT(cfg); load_formula_weights(m, cfg, perturb_scalars=False); m = m.cuda().set_precision("bf16").train(); normalize_matrices(m)
opt = m.configure_optimizers(0.1, 1e-3, (0.9, 0.95), "cuda")
X, y = synthetic_batch(cfg, 32, seed=1234); X, y = X.cuda(), y.cuda()
for i in range(pre):
    l, loss, aux, g = train_step(m, opt, X, y, 1.0)
    print("eager", i, loss.item(), g.item(), flush=True)
gs = GraphedTrainStep(m, opt, X, y, 1.0, warmup=1)
for i in range(6):
    l, loss, aux, g = gs(X, y)
torch.cuda.synchronize()
print("graph x6 nosync", loss.item(), g.item(), torch.isfinite(l).all().item(), flush=True)
from nvit_amd import ops
print("after replays loss", loss.item())
ops.prof_enable(True); ops.prof_collect()
for i in range(3):
    l2, loss2, _, g2 = train_step(m, opt, X, y, 1.0)
    print("post eager", i, loss2.item(), g2.item(), "static loss now", loss.item(), flush=True)
