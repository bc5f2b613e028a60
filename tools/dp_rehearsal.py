"""2-rank data-parallel rehearsal on ONE GPU (gloo backend, both ranks on cuda:0): checks that the bucketed
all-reduce of nvit_amd.parallel.DataParallel around the HIP-backed model yields the single-process gradients
of the concatenated batch.  Launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1
--master-port 29511 tools/dp_rehearsal.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nvit_amd.config import named_config
from nvit_amd.model import ViT
from nvit_amd.parallel import DataParallel
from nvit_amd.weights import formula_state_dict, synthetic_batch

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="env://")
cfg = named_config("mini")
m = ViT(cfg); m.load_state_dict(formula_state_dict(cfg)); m = m.to("cuda:0").set_precision("fp32").train()
dp = DataParallel(m, bucket_cap_mb=0.25)
X, y = synthetic_batch(cfg, 8)
xs, ys = X.chunk(world)[rank].cuda(), y.chunk(world)[rank].cuda()
res = []
for step in range(3):
    for p in m.parameters(): p.grad = None
    logits, _ = dp(xs)
    torch.nn.functional.cross_entropy(logits, ys).backward()
    torch.cuda.synchronize()
    res.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
if rank == 0:
    ref = ViT(cfg); ref.load_state_dict(formula_state_dict(cfg)); ref = ref.to("cuda:0").set_precision("fp32").train()
    logits, _ = ref(X.cuda())
    torch.nn.functional.cross_entropy(logits, y.cuda()).backward()
    worst = 0.0
    for step in range(3):
        for n, p in ref.named_parameters():
            if p.grad is None: continue
            e = (res[step][n] - p.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-12)
            worst = max(worst, e)
    print(f"dp rehearsal: buckets={dp.num_buckets} worst relative grad error vs single process = {worst:.3e}")
    assert worst < 1e-4
dist.barrier()
dist.destroy_process_group()
