"""ViTConfig: field-for-field mirror of the reference dataclass.

Reference: /root/reference/nvit/model.py:13-40 (26 fields, same names, order and
defaults, so `ViTConfig(**model_args)` built by the reference trainer at
train.py:398-422 constructs unchanged).
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class ViTConfig:
    image_size: int = 224
    n_layer: int = 12
    n_head: int = 12
    n_embd: int = 1024
    base_scale: float = 1.0 / (1024.0 ** 0.5)
    use_nvit: bool = False
    flash_attn: bool = False
    sz_init_value: float = 1.00
    sz_init_scaling: float = 1.0
    dropout: float = 0.0
    bias: bool = False
    channels: int = 3
    num_classes: int = 1000
    local_patch_size: int = 8
    global_patch_size: int = 16
    kohonen_nodes: int = 512
    kohonen_alpha: float = 0.01
    use_kohonen: bool = False
    reconstruction_weight: float = 0.1
    map_balance_weight: float = 0.5
    kohonen_scheduler_enabled: bool = False
    kohonen_scheduler_warmup_steps: int = 1000
    kohonen_scheduler_decay_steps: int = 10000
    kohonen_scheduler_min_lr: float = 0.001
    local_quantization_weight: float = 0.1
    global_quantization_weight: float = 0.1


def named_config(name: str, **over) -> ViTConfig:
    """The concrete configurations of SURVEY.md §8.0 (BASELINE.json `configs`)."""
    common = dict(use_nvit=True, flash_attn=False, bias=False, dropout=0.0)
    table = {
        # C1: nViT-Tiny/16 CIFAR-10 (CPU reference config)
        "tiny": dict(image_size=32, n_embd=192, n_layer=12, n_head=3, num_classes=10),
        # C1': as-shipped settings.yaml model (parity only)
        "micro": dict(image_size=32, n_embd=64, n_layer=2, n_head=2, num_classes=100, bias=True, dropout=0.15),
        # C2: nViT-Base/16 224 px, reference patch geometry 8/16 (T=784)
        "base": dict(image_size=224, n_embd=768, n_layer=12, n_head=12, num_classes=1000),
        # C2': conventional 16-px token grid (T=196)
        "base_p16": dict(image_size=224, n_embd=768, n_layer=12, n_head=12, num_classes=1000,
                         local_patch_size=16, global_patch_size=32),
        # C4: nViT-Large/16
        "large": dict(image_size=224, n_embd=1024, n_layer=24, n_head=16, num_classes=1000),
        "large_p16": dict(image_size=224, n_embd=1024, n_layer=24, n_head=16, num_classes=1000,
                          local_patch_size=16, global_patch_size=32),
        # small parity configs with head dim 64 and a ragged token count
        "mini": dict(image_size=56, n_embd=128, n_layer=2, n_head=2, num_classes=16),
        # C5-style parity configs: Kohonen head on (nodes per map must be a perfect square)
        "micro_k": dict(image_size=32, n_embd=64, n_layer=2, n_head=2, num_classes=100, bias=True, dropout=0.15,
                        use_kohonen=True, kohonen_nodes=128, kohonen_alpha=0.02),
        "mini_k": dict(image_size=56, n_embd=128, n_layer=2, n_head=2, num_classes=16, use_kohonen=True,
                       kohonen_nodes=32, kohonen_alpha=0.05, reconstruction_weight=0.5),
        # C5: nViT-Base + Kohonen head (512 nodes = 2 maps of 16x16)
        "base_k": dict(image_size=224, n_embd=768, n_layer=12, n_head=12, num_classes=1000, use_kohonen=True,
                       kohonen_nodes=512),
    }
    kw = dict(common)
    kw.update(table[name])
    kw.update(over)
    return ViTConfig(**kw)


def train_flops_per_image(cfg: ViTConfig) -> float:
    """Algorithmic train FLOPs per image = 3 x forward (SURVEY.md §8.0 formula, 2*MAC)."""
    C, L = cfg.n_embd, cfg.n_layer
    T = (cfg.image_size // cfg.local_patch_size) ** 2
    Kl = cfg.channels * cfg.local_patch_size ** 2
    Kg = cfg.channels * cfg.global_patch_size ** 2
    fwd = (2 * T * Kl * C + 2 * T * Kg * C + (12 * T * C * C + 4 * T * T * C)
           + L * (32 * T * C * C + 4 * T * T * C) + 2 * C * cfg.num_classes + 2 * T * C * Kl)
    return 3.0 * fwd
