"""CPU oracle for the nViT hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (nvit_amd/) never does and fails loudly when the
HIP library is missing.

This is a from-scratch restatement of the reference's algorithm in plain fp32
PyTorch (CPU), written from the mathematics in SURVEY.md §9.2 as explicit
matrix products / softmax / norms over a flat {state_dict name -> tensor}
dictionary.  It is not a transliteration of the reference's nn.Modules: there
are no modules, no einops, no SDPA and no conv2d calls here.

What each function restates (citations into /root/reference/):
  nrm            nvit/model.py:43-44      x / ||x||_2, no eps
  im2col         nvit/model.py:286-304    Conv2d(k=P,s=P_l) (+ReflectionPad2d) as a GEMM operand
  attend         nvit/model.py:104-127    per-head cosine-normalised attention (SDPA branch :124)
  lerp           nvit/model.py:134-142    normalised LERP residual
  cross_block    nvit/model.py:219-275
  block          nvit/model.py:92-169 + norm_skip :84-87 (called at :452)
  forward        nvit/model.py:403-470 (both branches: plain cross-attention and the Kohonen head :419-444)
  bmu/som_update_/map_smoothness/consistency/huber   nvit/kohonen.py:80-165, model.py:482-561
  renorm_        nvit/train.py:461-480    post-step weight re-normalisation
  param_groups   nvit/model.py:369-385    AdamW groups
  train_step     nvit/train.py:898-946,989-990

Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
golden vectors produced by importing the real reference on CPU
(oracle/make_golden.py, run in the build container; fixtures in tests/golden/),
including the known-answer record of SURVEY.md §9.3.

`lowp` hook (bf16_round, or a KernelRounding instance): when not None it is applied to every GEMM operand (activations and
weights) and to the un-normalised softmax probabilities exp(s - rowmax) (the
operand a fused attention kernel feeds its second product), emulating "bf16 MFMA
operands, fp32 accumulate" so the bf16 HIP path can be compared at a tight
tolerance.  With lowp=None nothing changes (plain softmax; pinned by the goldens).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Tuple

import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]
LowP = Optional[Callable[[Tensor], Tensor]]


def bf16_round(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


class KernelRounding:
    """bf16-operand emulation that rounds at the points where the HIP kernels round (still a CPU computation over the
    reference's algorithm; what changes is WHICH fp32 value each bf16 operand is the rounding of):

      * attention (head dim 64, the MFMA kernels): `nvit_attn_fwd_bounded` takes the probabilities relative to the
        score bound of the head, tb = sqrt(d)*log2(e)*max_d(sqk*c_q)^2, instead of the row maximum: the second
        product's operand is bf16(exp2(s*log2e - tb)), and the row sum that normalises the output is the fp32 sum of
        those ROUNDED values (it comes out of the matrix pipe as one extra ones-row).  The factor between the two
        conventions is not a power of two, so every probability rounds independently of the row-maximum form.
      * on the fused path (n_embd % 256 == 0) the q projection leaves its GEMM epilogue pre-scaled by sqrt(d)*log2(e)
        (one rounding of q_hat*prescale instead of rounding q_hat and scaling the fp32 score).
      * `y_bf16`: the branch outputs y (att_c_proj / mlp_c_proj / out_proj results, the second LERP input) are stored in
        bf16, as the reference's own autocast nn.Linear returns them.
    `acc64=True` additionally accumulates every matrix product in float64 (same operand roundings, different summation
    order/accuracy): the distance between the two is the noise floor of 'same rounding points, other summation order'.
    With attn_conv="rowmax", acc64=False and y_bf16=False this is exactly `bf16_round`."""

    def __init__(self, attn_conv: str = "bound", acc64: bool = False, y_bf16: bool = True) -> None:
        assert attn_conv in ("bound", "rowmax")
        self.attn_conv = attn_conv
        self.acc64 = acc64
        self.y_bf16 = y_bf16

    def __call__(self, t: Tensor) -> Tensor:
        return bf16_round(t)


def _ylo(y: Tensor, lowp: LowP) -> Tensor:
    """Storage rounding of a branch output (see KernelRounding.y_bf16)."""
    return lowp(y) if getattr(lowp, "y_bf16", False) else y


def _mm(a: Tensor, b: Tensor, lowp: LowP) -> Tensor:
    """a @ b with the accumulation precision the rounding mode asks for."""
    if getattr(lowp, "acc64", False):
        return (a.double() @ b.double()).float()
    return a @ b


def _lp(t: Tensor, lowp: LowP) -> Tensor:
    return t if lowp is None else lowp(t)


def nrm(x: Tensor) -> Tensor:
    return x / torch.sqrt((x * x).sum(dim=-1, keepdim=True))


def linear(x: Tensor, w: Tensor, b: Optional[Tensor], lowp: LowP) -> Tensor:
    y = _mm(_lp(x, lowp), _lp(w, lowp).t(), lowp)
    return y if b is None else y + b


def reflect_index(i: Tensor, n: int) -> Tensor:
    """ReflectionPad2d index map: -1 -> 1, n -> n-2 (no edge repeat)."""
    i = torch.where(i < 0, -i, i)
    return torch.where(i >= n, 2 * (n - 1) - i, i)


def im2col(img: Tensor, P: int, stride: int, pad: int) -> Tensor:
    """[B,ch,S,S] -> [B,T,ch*P*P], column order (c, ph, pw); reflect padding by index map."""
    B, ch, S, _ = img.shape
    G = (S + 2 * pad - P) // stride + 1
    base = torch.arange(G) * stride - pad
    off = torch.arange(P)
    rows = reflect_index(base[:, None] + off[None, :], S)          # [G,P]
    cols = rows
    # gather: out[b, gy, gx, c, ph, pw] = img[b, c, rows[gy,ph], cols[gx,pw]]
    x = img[:, :, rows.reshape(-1), :]                              # [B,ch,G*P,S]
    x = x[:, :, :, cols.reshape(-1)]                                # [B,ch,G*P,G*P]
    x = x.reshape(B, ch, G, P, G, P).permute(0, 2, 4, 1, 3, 5)      # [B,G,G,ch,P,P]
    return x.reshape(B, G * G, ch * P * P)


def heads(x: Tensor, H: int) -> Tensor:
    B, T, C = x.shape
    return x.reshape(B, T, H, C // H).permute(0, 2, 1, 3)           # [B,H,T,d]


def attend(q: Tensor, k: Tensor, v: Tensor, s_eff: Tensor, H: int, lowp: LowP) -> Tensor:
    """softmax(sqrt(d) * qh kh^T) v with qh = s*nrm(q) per head; returns [B,T,C]."""
    B, T, C = q.shape
    d = C // H
    s = s_eff.reshape(1, H, 1, d)
    qh = s * nrm(heads(q, H))
    kh = s * nrm(heads(k, H))
    vh = heads(v, H)
    if lowp is not None and getattr(lowp, "attn_conv", "rowmax") == "bound" and d == 64:
        # the MFMA kernels' rounding points (see KernelRounding): scores in log2 units, probabilities relative to the bound
        log2e = 1.4426950408889634
        c2t = math.sqrt(d) * log2e
        qpre = c2t if C % 256 == 0 else 1.0           # fused q/k-normalise GEMM epilogue folds the factor into q's scale
        tb = c2t * s_eff.reshape(H, d).abs().max(dim=-1).values ** 2          # [H]
        if float(tb.detach().max()) <= 60.0:
            qs = lowp(nrm(heads(q, H)) * (s * qpre))
            z = _mm(qs, lowp(kh).transpose(-1, -2), lowp) * (c2t / qpre) - tb.reshape(1, H, 1, 1)
            pt = lowp(torch.exp2(z))
            o = _mm(pt, lowp(vh), lowp) / pt.sum(dim=-1, keepdim=True)
            return o.permute(0, 2, 1, 3).reshape(B, T, C)
    scores = _mm(_lp(qh, lowp), _lp(kh, lowp).transpose(-1, -2), lowp) * math.sqrt(d)
    if lowp is None:
        o = torch.softmax(scores, dim=-1) @ vh
    else:
        # bf16-operand emulation: the second product's low-precision operand is the UN-normalised probability
        # exp(s - rowmax), as in every fused (flash-style) attention kernel incl. the reference's SDPA under autocast;
        # the row sum that normalises the output is taken in fp32 from the unrounded values
        pt = torch.exp(scores - scores.max(dim=-1, keepdim=True).values)
        o = _mm(lowp(pt), lowp(vh), lowp) / pt.sum(dim=-1, keepdim=True)
    return o.permute(0, 2, 1, 3).reshape(B, T, C)


def lerp(h: Tensor, y: Tensor, alpha: Tensor, c_a: float) -> Tensor:
    lam = torch.abs(alpha * c_a)
    a = nrm(h)
    b = nrm(y)
    return nrm(a + lam * (b - a))


def _b(p: Params, name: str) -> Optional[Tensor]:
    return p.get(name)


def cross_block(p: Params, cfg, local: Tensor, global_: Tensor, lowp: LowP) -> Tensor:
    pre = "cross_attention."
    c_q = 1.0 / cfg.base_scale
    c_a = 0.05 / cfg.base_scale
    q = linear(local, p[pre + "q_local.weight"], _b(p, pre + "q_local.bias"), lowp)
    k = linear(global_, p[pre + "k_global.weight"], _b(p, pre + "k_global.bias"), lowp)
    v = linear(global_, p[pre + "v_global.weight"], _b(p, pre + "v_global.bias"), lowp)
    o = attend(q, k, v, p[pre + "sqk"] * c_q, cfg.n_head, lowp)
    o = linear(o, p[pre + "proj.weight"], _b(p, pre + "proj.bias"), lowp)
    C = cfg.n_embd
    u, g = o[..., :C], o[..., C:]
    o = u * (g * torch.sigmoid(g))
    o = _ylo(linear(o, p[pre + "out_proj.weight"], _b(p, pre + "out_proj.bias"), lowp), lowp)
    return lerp(local, o, p[pre + "attn_alpha"], c_a)


def block(p: Params, cfg, i: int, x: Tensor, lowp: LowP) -> Tensor:
    pre = f"transformer.h.{i}."
    C = cfg.n_embd
    c_q = 1.0 / cfg.base_scale
    c_a = 0.05 / cfg.base_scale
    q = linear(x, p[pre + "query.weight"], _b(p, pre + "query.bias"), lowp)
    k = linear(x, p[pre + "key.weight"], _b(p, pre + "key.bias"), lowp)
    v = linear(x, p[pre + "value.weight"], _b(p, pre + "value.bias"), lowp)
    o = attend(q, k, v, p[pre + "sqk"] * c_q, cfg.n_head, lowp)
    y = _ylo(linear(o, p[pre + "att_c_proj.weight"], _b(p, pre + "att_c_proj.bias"), lowp), lowp)
    h1 = lerp(x, y, p[pre + "attn_alpha"], c_a)
    uv = linear(h1, p[pre + "c_fc.weight"], _b(p, pre + "c_fc.bias"), lowp)
    uv = uv * (p[pre + "suv"] * (1.0 * math.sqrt(C)))
    u, g = uv[..., : 4 * C], uv[..., 4 * C:]
    xm = u * (g * torch.sigmoid(g))
    y2 = _ylo(linear(xm, p[pre + "mlp_c_proj.weight"], _b(p, pre + "mlp_c_proj.bias"), lowp), lowp)
    h2 = lerp(h1, y2, p[pre + "mlp_alpha"], c_a)
    return nrm(h2 * p[pre + "skip_param"] + x)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def embed(p: Params, cfg, img: Tensor, lowp: LowP) -> Tuple[Tensor, Tensor, Tensor]:
    Pl, Pg = cfg.local_patch_size, cfg.global_patch_size
    C = cfg.n_embd
    A_l = im2col(img, Pl, Pl, 0)
    A_g = im2col(img, Pg, Pl, (Pg - Pl) // 2)
    # (the HIP path's bf16 mode keeps the two patch-embedding GEMMs in exact fp32, so the emulation does too)
    loc = linear(A_l, p["local_patch_embed.weight"].reshape(C, -1), p["local_patch_embed.bias"], None)
    glo = linear(A_g, p["global_patch_embed.1.weight"].reshape(C, -1), p["global_patch_embed.1.bias"], None)
    return loc + p["local_pos_embed"], glo + p["global_pos_embed"], A_l


# ---- Kohonen head (BASELINE config C5): /root/reference/nvit/kohonen.py:30-165, model.py:419-444,482-561 ----
def som_dims(num_nodes: int) -> Tuple[int, int]:
    m = int(num_nodes ** 0.5)          # kohonen.py:52-54
    return m, num_nodes // m


def bmu(x: Tensor, nodes: Tensor) -> Tensor:
    """index of the nearest node (L2) for every token: kohonen.py:111-114 (cdist + argmin)."""
    return torch.argmin(torch.cdist(x, nodes, p=2), dim=-1)


def som_neighborhood_d2(bmu_loc: Tensor, m: int, n: int, periodic: bool = True) -> Tensor:
    """squared grid distance of every node to bmu_loc (kohonen.py:80-98): periodic map = min over the 9 wrapped copies,
    otherwise the plain distance."""
    ii, jj = torch.meshgrid(torch.arange(m), torch.arange(n), indexing="ij")
    loc = torch.stack([ii.reshape(-1), jj.reshape(-1)], dim=1).float()
    if not periodic:
        return ((loc - bmu_loc.float()[None, :]) ** 2).sum(-1)
    offs = torch.tensor([[0, 0], [-m, -n], [m, n], [-m, 0], [m, 0], [0, -n], [0, n], [-m, n], [m, -n]]).float()
    d = loc[None, :, :] + offs[:, None, :] - bmu_loc.float()[None, None, :]
    return (d * d).sum(-1).min(dim=0).values


@torch.no_grad()
def som_update_(nodes: Tensor, x: Tensor, idx: Tensor, lr: float, alpha: float, periodic: bool = True) -> None:
    """kohonen.py:121-165 literally, including its pairing quirk: only B iterations;
    iteration i takes the BMU of FLAT token i and the whole image i pooled T*C -> C by means of T consecutive
    elements of the flattened [T, C] buffer; updates are sequential."""
    N, C = nodes.shape
    m, n = som_dims(N)
    sigma = (m * n) ** 0.5 / 2.0
    flat_idx = idx.reshape(-1)
    B = x.shape[0]
    for i in range(B):
        w = int(flat_idx[i])
        d2 = som_neighborhood_d2(torch.tensor([w // n, w % n]), m, n, periodic)
        strength = lr * alpha * torch.exp(-d2 / (2 * sigma * sigma))
        v = x[i].reshape(-1)
        sz = v.numel()
        if sz > C:
            v = v.view(-1, sz // C).mean(dim=1)
        elif sz < C:
            v = v.repeat(C // sz)
        nodes.add_(strength[:, None] * (v[None, :] - nodes))


def neighbor_indices(idx: Tensor, nodes_per_map: int) -> Tensor:
    """8-neighbourhood on the periodic map_size x map_size grid (model.py:503-536)."""
    ms = int(math.sqrt(nodes_per_map))
    if ms * ms != nodes_per_map:
        raise ValueError("nodes per map must be a perfect square (SURVEY.md §9.1-Q2)")
    offs = torch.tensor([[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 1], [1, -1], [1, 0], [1, 1]])
    row = (idx // ms)[..., None] + offs[:, 0]
    col = (idx % ms)[..., None] + offs[:, 1]
    return (row % ms) * ms + (col % ms)


def map_smoothness(nodes: Tensor, idx: Tensor) -> Tensor:
    nb = neighbor_indices(idx, nodes.shape[0])
    cur = nodes[idx]
    d = cur[..., None, :] - nodes[nb]
    return torch.sqrt((d * d).sum(-1)).mean()


def consistency(a: Tensor, b: Tensor) -> Tensor:
    return 1.0 - (nrm(a) * nrm(b)).sum(-1).mean()


def huber(a: Tensor, b: Tensor) -> Tensor:
    d = a - b
    ad = d.abs()
    return torch.where(ad < 1.0, 0.5 * d * d, ad - 0.5).mean()


def kohonen_lr(cfg, step: int) -> float:
    if not cfg.kohonen_scheduler_enabled:
        return cfg.kohonen_alpha
    w, dcy = cfg.kohonen_scheduler_warmup_steps, cfg.kohonen_scheduler_decay_steps
    lo, hi = cfg.kohonen_scheduler_min_lr, cfg.kohonen_alpha
    if step < w:
        return lo + (hi - lo) * (step / w)
    if step > dcy:
        return lo
    return lo + 0.5 * (1.0 + math.cos(math.pi * (step - w) / (dcy - w))) * (hi - lo)


def forward(p: Params, cfg, img: Tensor, lowp: LowP = None, taps: Optional[dict] = None, training: bool = True,
            step: int = 1):
    """-> (logits [B,ncls], aux dict).  `step` is the value of ViT.step AFTER its increment (model.py:404-405)."""
    assert cfg.use_nvit
    loc, glo, A_l = embed(p, cfg, img, lowp)
    aux = {}
    if cfg.use_kohonen:
        alpha = cfg.kohonen_alpha if not cfg.kohonen_scheduler_enabled else cfg.kohonen_scheduler_min_lr
        lr = kohonen_lr(cfg, step)
        Ln, Gn = p["local_kohonen.nodes"], p["global_kohonen.nodes"]
        lidx, gidx = bmu(loc.detach(), Ln.detach()), bmu(glo.detach(), Gn.detach())
        lrepr, grepr = Ln[lidx], Gn[gidx]
        if training:
            som_update_(Ln, loc.detach(), lidx, lr, alpha)
            som_update_(Gn, glo.detach(), gidx, lr, alpha)
        local_new = cross_block(p, cfg, lrepr, loc, lowp)
        global_new = cross_block(p, cfg, grepr, glo, lowp)
        aux["kohonen_consistency"] = consistency(lrepr, grepr)
        aux["kohonen_smoothness"] = map_smoothness(Ln, lidx) + map_smoothness(Gn, gidx)
        aux["local_quantization"] = huber(lrepr, loc)
        aux["global_quantization"] = huber(grepr, glo)
        x = cross_block(p, cfg, local_new, global_new, lowp)
        if taps is not None:
            taps["lidx"], taps["gidx"] = lidx, gidx
    else:
        x = cross_block(p, cfg, loc, glo, lowp)
    if taps is not None:
        taps["loc"], taps["glo"], taps["x0"] = loc, glo, x
    for i in range(cfg.n_layer):
        x = block(p, cfg, i, x, lowp)
        if taps is not None:
            taps[f"x{i + 1}"] = x
    pooled = x.mean(dim=1)
    ln = layer_norm(pooled, p["mlp_head.0.weight"], p["mlp_head.0.bias"])
    # the classifier GEMM ([B,C]x[C,ncls], negligible work) stays fp32 in the HIP path's bf16 mode, so the bf16-operand
    # emulation does not round its operands either
    logits = linear(ln, p["mlp_head.1.weight"], p["mlp_head.1.bias"], None)
    logits = logits * (p["sz"] * (cfg.sz_init_value / cfg.sz_init_scaling))
    rec = torch.tanh(linear(x, p["reconstruction_head.0.weight"], p["reconstruction_head.0.bias"], lowp))
    aux["reconstruction"] = ((rec - A_l) ** 2).mean()
    return logits, aux


def total_loss(cfg, logits: Tensor, aux, y: Tensor, consistency_weight: float = 0.1, smoothness_weight: float = 0.1):
    """train.py:906-926: CE, plus the weighted aux losses only when the Kohonen head is on."""
    loss = cross_entropy(logits, y)
    if cfg.use_kohonen:
        loss = (loss + consistency_weight * aux["kohonen_consistency"] + smoothness_weight * aux["kohonen_smoothness"]
                + cfg.local_quantization_weight * aux["local_quantization"]
                + cfg.global_quantization_weight * aux["global_quantization"]
                + cfg.reconstruction_weight * aux["reconstruction"])
    return loss


def cross_entropy(logits: Tensor, y: Tensor) -> Tensor:
    lse = torch.logsumexp(logits, dim=-1)
    return (lse - logits.gather(1, y[:, None]).squeeze(1)).mean()


RENORM_ROWS = ("query", "key", "value", "c_fc")          # dim=1 (train.py:475-477,479)
RENORM_COLS = ("att_c_proj", "mlp_c_proj")               # dim=0 (train.py:478,480)


@torch.no_grad()
def renorm_(p: Params, cfg) -> None:
    for i in range(cfg.n_layer):
        for n in RENORM_ROWS:
            w = p[f"transformer.h.{i}.{n}.weight"]
            w.copy_(w / torch.sqrt((w * w).sum(dim=1, keepdim=True)))
        for n in RENORM_COLS:
            w = p[f"transformer.h.{i}.{n}.weight"]
            w.copy_(w / torch.sqrt((w * w).sum(dim=0, keepdim=True)))


def param_groups(p: Params, weight_decay: float):
    """model.py:369-385 nViT branch: decay >=2-D non-'sz' params; no decay for <2-D and sz."""
    decay = [t for n, t in p.items() if "sz" not in n and t.dim() >= 2]
    nodecay = [t for n, t in p.items() if "sz" not in n and t.dim() < 2]
    return [
        {"params": decay, "weight_decay": weight_decay},
        {"params": nodecay, "weight_decay": 0.0},
        {"params": [p["sz"]], "weight_decay": 0.0},
    ]


def make_params(state: Dict[str, Tensor]) -> Params:
    return {n: t.detach().clone().float().requires_grad_(True) for n, t in state.items()}


def make_optimizer(p: Params, lr: float = 1e-3, weight_decay: float = 0.1, betas=(0.9, 0.95)):
    return torch.optim.AdamW(param_groups(p, weight_decay), lr=lr, betas=betas)


def loss_and_grads(p: Params, cfg, X: Tensor, y: Tensor, lowp: LowP = None, step: int = 1, want_aux: bool = False):
    for t in p.values():
        t.grad = None
    logits, aux = forward(p, cfg, X, lowp, training=True, step=step)
    loss = total_loss(cfg, logits, aux, y)
    loss.backward()
    if want_aux:
        return logits.detach(), loss.detach(), {k: v.detach() for k, v in aux.items()}
    return logits.detach(), loss.detach(), aux["reconstruction"].detach()


def total_grad_norm(p: Params) -> Tensor:
    gs = [t.grad for t in p.values() if t.grad is not None]
    return torch.sqrt(sum((g * g).sum() for g in gs))


def train_step(p: Params, cfg, opt, X: Tensor, y: Tensor, grad_clip: float = 1.0, lowp: LowP = None, step: int = 1):
    """forward -> CE -> backward -> clip -> AdamW -> zero_grad -> renorm (train.py:898-946,989-990)."""
    logits, loss, recon = loss_and_grads(p, cfg, X, y, lowp, step=step)
    gnorm = torch.nn.utils.clip_grad_norm_([t for t in p.values() if t.grad is not None], grad_clip)
    opt.step()
    opt.zero_grad(set_to_none=True)
    renorm_(p, cfg)
    return logits, loss, recon, gnorm
