"""Closed-form, counter-based parameter values (no torch RNG).

The GPU box has no copy of the reference, and torch's RNG streams are not a
stable contract, so every test / bench weight is a pure function of
(parameter name, element index).  The same function is used to fill the
reference module (in the build container, see oracle/make_golden.py), the CPU
oracle and the HIP-backed model, so all three start from bit-identical fp32
parameters.

Parameter names and shapes follow the reference's state_dict
(/root/reference/nvit/model.py:279-356).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Tuple

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def unit_uniform(name: str, numel: int, salt: int = 0) -> np.ndarray:
    """float64 values in [-1, 1), a pure function of (name, index, salt)."""
    seed = np.uint64(((zlib.crc32(name.encode()) & 0xFFFFFFFF) * 0x100000001B3 + salt) & 0xFFFFFFFFFFFFFFFF)
    idx = np.arange(numel, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + seed)
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24 bits -> [0,1)
    return u * 2.0 - 1.0


def formula_tensor(name: str, shape: Tuple[int, ...], std: float, mean: float = 0.0, salt: int = 0) -> torch.Tensor:
    """fp32 tensor, uniform with the given std (bound = std*sqrt(3)) around mean."""
    n = int(np.prod(shape)) if len(shape) else 1
    v = unit_uniform(name, n, salt) * (std * math.sqrt(3.0)) + mean
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def param_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """state_dict parameter names -> shapes for an nViT-mode config."""
    C, L = cfg.n_embd, cfg.n_layer
    Pl, Pg, ch = cfg.local_patch_size, cfg.global_patch_size, cfg.channels
    T = (cfg.image_size // Pl) ** 2
    Kl = ch * Pl * Pl
    s: Dict[str, Tuple[int, ...]] = {}
    s["local_pos_embed"] = (1, T, C)
    s["global_pos_embed"] = (1, T, C)
    s["sz"] = (cfg.num_classes,)
    s["local_patch_embed.weight"] = (C, ch, Pl, Pl)
    s["local_patch_embed.bias"] = (C,)
    s["global_patch_embed.1.weight"] = (C, ch, Pg, Pg)
    s["global_patch_embed.1.bias"] = (C,)
    s["cross_attention.attn_alpha"] = (C,)
    s["cross_attention.sqk"] = (C,)
    for n in ("q_local", "k_global", "v_global", "out_proj"):
        s[f"cross_attention.{n}.weight"] = (C, C)
        if cfg.bias:
            s[f"cross_attention.{n}.bias"] = (C,)
    s["cross_attention.proj.weight"] = (2 * C, C)
    if cfg.bias:
        s["cross_attention.proj.bias"] = (2 * C,)
    s["reconstruction_head.0.weight"] = (Kl, C)
    s["reconstruction_head.0.bias"] = (Kl,)
    for i in range(L):
        p = f"transformer.h.{i}."
        s[p + "skip_param"] = (1,)
        s[p + "attn_alpha"] = (C,)
        s[p + "mlp_alpha"] = (C,)
        s[p + "sqk"] = (C,)
        s[p + "suv"] = (8 * C,)
        for n in ("key", "query", "value", "att_c_proj"):
            s[p + n + ".weight"] = (C, C)
            if cfg.bias:
                s[p + n + ".bias"] = (C,)
        s[p + "c_fc.weight"] = (8 * C, C)
        s[p + "mlp_c_proj.weight"] = (C, 4 * C)
        if cfg.bias:
            s[p + "c_fc.bias"] = (8 * C,)
            s[p + "mlp_c_proj.bias"] = (C,)
        s[p + "rmsnorm_att.weight"] = (C,)
        s[p + "rmsnorm_mlp.weight"] = (C,)
    if cfg.use_kohonen:
        m = int((cfg.kohonen_nodes // 2) ** 0.5)
        n = (cfg.kohonen_nodes // 2) // m
        s["local_kohonen.nodes"] = (m * n, C)
        s["global_kohonen.nodes"] = (m * n, C)
        s["map_balance"] = ()
    s["mlp_head.0.weight"] = (C,)
    s["mlp_head.0.bias"] = (C,)
    s["mlp_head.1.weight"] = (cfg.num_classes, C)
    s["mlp_head.1.bias"] = (cfg.num_classes,)
    return s


def formula_state_dict(cfg, perturb_scalars: bool = True, salt: int = 0) -> Dict[str, torch.Tensor]:
    """Full nViT-mode state_dict from the closed-form formula.

    Magnitudes mimic the reference's init (model.py:354-367: Linear N(0,0.02),
    *c_proj N(0,0.02/sqrt(2L)), Conv2d default bound 1/sqrt(fan_in)); learned
    scale vectors sit at their init value (model.py:68-81), optionally perturbed
    by +-10 % so that tests exercise every per-channel gradient path.
    """
    L = cfg.n_layer
    bs = float(cfg.base_scale)
    jit = 0.1 if perturb_scalars else 0.0
    out: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        leaf = name.split(".")[-1]
        if name == "map_balance":
            t = torch.tensor(float(cfg.map_balance_weight))
        elif name.endswith("kohonen.nodes"):
            t = formula_tensor(name, shape, 1.0, salt=salt)      # reference: randn (kohonen.py:60)
        elif name.endswith("pos_embed"):
            t = formula_tensor(name, shape, 0.02 if perturb_scalars else 0.0, salt=salt)
        elif name == "sz":
            t = formula_tensor(name, shape, jit * cfg.sz_init_scaling, cfg.sz_init_value, salt)
        elif leaf in ("attn_alpha", "mlp_alpha", "sqk"):
            t = formula_tensor(name, shape, jit * bs, bs, salt)
        elif leaf == "suv":
            t = formula_tensor(name, shape, jit, 1.0, salt)
        elif leaf == "skip_param":
            t = formula_tensor(name, shape, jit, 1.0, salt)
        elif "rmsnorm" in name:
            t = torch.ones(shape)
        elif name == "mlp_head.0.weight":
            t = formula_tensor(name, shape, jit, 1.0, salt)
        elif name == "mlp_head.0.bias":
            t = formula_tensor(name, shape, 0.02 if perturb_scalars else 0.0, salt=salt)
        elif "patch_embed" in name:
            w_shape = shape if leaf == "weight" else param_shapes(cfg)[name[: -len("bias")] + "weight"]
            fan_in = w_shape[1] * w_shape[2] * w_shape[3]
            bound = 1.0 / math.sqrt(fan_in)
            t = formula_tensor(name, shape, bound / math.sqrt(3.0), salt=salt)
        elif leaf == "bias":
            t = formula_tensor(name, shape, 0.02 if perturb_scalars else 0.0, salt=salt)
        elif name.endswith("c_proj.weight"):
            t = formula_tensor(name, shape, 0.02 / math.sqrt(2 * L), salt=salt)
        else:
            t = formula_tensor(name, shape, 0.02, salt=salt)
        out[name] = t
    return out


def synthetic_batch(cfg, batch: int, seed: int = 1234, salt: int = 0):
    """X = 2*u-1 in [-1,1) fp32 [B,ch,S,S], y int64 [B], formula-generated."""
    S = cfg.image_size
    X = formula_tensor(f"X{seed}", (batch, cfg.channels, S, S), 1.0 / math.sqrt(3.0), salt=salt)
    u = (unit_uniform(f"y{seed}", batch, salt) + 1.0) * 0.5
    y = torch.from_numpy(np.minimum((u * cfg.num_classes).astype(np.int64), cfg.num_classes - 1))
    return X, y


def load_formula_weights(model, cfg, **kw) -> None:
    """model.load_state_dict(formula_state_dict(cfg)); only the Kohonen index buffers (locations / offsets, which
    the module builds itself) may be absent from the formula dictionary."""
    res = model.load_state_dict(formula_state_dict(cfg, **kw), strict=False)
    bad = [k for k in res.missing_keys if not k.endswith((".locations", ".offsets"))]
    if bad or res.unexpected_keys:
        raise RuntimeError(f"formula weights do not match the module: missing {bad}, unexpected {res.unexpected_keys}")
