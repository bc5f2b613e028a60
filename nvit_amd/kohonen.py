"""KohonenMap on HIP kernels — mirror of /root/reference/nvit/kohonen.py:30-165 (same constructor, parameter
`nodes`, buffers `locations` / `offsets`, `forward(x) -> (node_repr, idx)`, `update_nodes(x, idx, lr)`).

The reference's `update_nodes` is a Python loop of ~242 torch kernels per sample; here the whole batch is two
launches (nvit_som_update), reproducing the reference literally, including its pairing quirk (only B steps; step i
uses the BMU of flat token i and sample i mean-pooled to C values — SURVEY.md §9.1-Q13)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn

from . import ops

Tensor = torch.Tensor


class _GatherFn(torch.autograd.Function):
    """repr = nodes[idx] (kohonen.py:117); backward = fixed-order segmented row sum into the node gradient."""

    @staticmethod
    def forward(ctx, nodes, idx):
        ctx.save_for_backward(idx)
        ctx.n = nodes.shape[0]
        return ops.gather_rows(nodes.detach().contiguous(), idx)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        return ops.scatter_rows(dout.contiguous(), idx, ctx.n), None


class CosConsistencyFn(torch.autograd.Function):
    """1 - mean cosine(a, b) over rows (reference model.py:482-491)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        loss, stats = ops.cos_consistency_fwd(a, b)
        ctx.save_for_backward(a, b, stats)
        return loss

    @staticmethod
    def backward(ctx, g):
        a, b, stats = ctx.saved_tensors
        return ops.cos_consistency_bwd(a, b, stats, g.contiguous().reshape(1))


class HuberFn(torch.autograd.Function):
    """F.huber_loss(a, b), delta=1, mean (reference model.py:441-442)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return ops.huber_fwd(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return ops.huber_bwd(a, b, g.contiguous().reshape(1))


class MapSmoothnessFn(torch.autograd.Function):
    """mean over tokens and the 8 periodic grid neighbours of ||nodes[idx] - nodes[nb]|| (model.py:538-561)."""

    @staticmethod
    def forward(ctx, nodes, idx, map_size):
        nd = nodes.detach().contiguous()
        loss, cnt, D = ops.som_smooth_fwd(nd, idx, map_size)
        ctx.save_for_backward(nd.clone(), cnt, D)   # nodes are mutated in place by later SOM updates
        ctx.M, ctx.ms = idx.numel(), map_size
        return loss

    @staticmethod
    def backward(ctx, g):
        nd, cnt, D = ctx.saved_tensors
        return ops.som_smooth_bwd(nd, D, cnt, g.contiguous().reshape(1), ctx.M, ctx.ms), None, None


class KohonenMap(nn.Module):
    def __init__(self, input_dim: int, num_nodes: int, alpha: float = 0.01, sigma: Optional[float] = None,
                 periodic: bool = True) -> None:
        super().__init__()
        self.m = int(num_nodes ** 0.5)
        self.n = num_nodes // self.m
        self.grid_size = self.m * self.n
        self.input_dim = input_dim
        self.alpha = alpha
        self.periodic = periodic
        self.nodes = nn.Parameter(torch.randn(self.grid_size, input_dim))
        locs = torch.tensor([[i, j] for i in range(self.m) for j in range(self.n)], dtype=torch.long)
        self.register_buffer("locations", locs)
        self.sigma = (self.m * self.n) ** 0.5 / 2.0 if sigma is None else float(sigma)
        if periodic:   # (the reference registers the wrap offsets only for the periodic map, kohonen.py:70-78)
            offsets = [[-self.m, -self.n], [self.m, self.n], [-self.m, 0], [self.m, 0], [0, -self.n], [0, self.n],
                       [-self.m, self.n], [self.m, -self.n]]
            self.register_buffer("offsets", torch.tensor(offsets))

    def get_neighborhood_distances(self, bmu_loc: Tensor) -> Tensor:
        """Squared grid distance of every node to `bmu_loc` ([2]: row, col); on the periodic map the minimum over the
        un-shifted grid and its 8 wrapped copies (reference kohonen.py:80-98).  Index arithmetic on [m*n, 2] integers
        (the SOM update kernel recomputes the same quantity per node in registers, kohonen.hip); returns fp32 [m*n]."""
        loc = self.locations.float()
        bmu = bmu_loc.to(loc.device).float().reshape(1, 2)
        if not self.periodic:
            return ((loc - bmu) ** 2).sum(-1)
        shifts = torch.cat((torch.zeros(1, 2, device=loc.device), self.offsets.float()), dim=0)   # [9, 2]
        d = loc.unsqueeze(0) + shifts.unsqueeze(1) - bmu.unsqueeze(0)                              # [9, m*n, 2]
        return (d * d).sum(-1).min(dim=0).values

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        """x [..., C] -> (node representations [..., C], winning indices [...])."""
        if not x.is_cuda:
            raise RuntimeError("KohonenMap runs only on the HIP device (no CPU fallback)")
        if x.dim() == 1:
            x = x.unsqueeze(0)
        lead = x.shape[:-1]
        x2 = x.detach().reshape(-1, x.shape[-1]).contiguous().float()
        idx = ops.som_bmu(x2, self.nodes.detach().contiguous())
        node_repr = _GatherFn.apply(self.nodes, idx)
        return node_repr.reshape(*lead, x.shape[-1]), idx.reshape(lead)

    @torch.no_grad()
    def update_nodes(self, x: Tensor, winning_indices: Tensor, learning_rate: float) -> None:
        if not self.training:
            return
        if x.dim() != 3:
            raise ValueError("update_nodes expects x of shape [B, T, C]")
        B, T, _ = x.shape
        ops.som_update(self.nodes.data, x.detach().contiguous().float(), winning_indices.reshape(-1).contiguous(),
                       float(learning_rate) * float(self.alpha), self.sigma, self.m, self.n, B, T, periodic=self.periodic)
