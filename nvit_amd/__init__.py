"""nvit_amd: MI355X-native nViT hot path (HIP kernels behind the reference's Python API)."""
from .config import ViTConfig, named_config, train_flops_per_image  # noqa: F401


def __getattr__(name):
    # model/ops import torch and bind the HIP library lazily
    if name in ("ViT", "Block", "CrossAttentionBlock", "RMSNorm"):
        from . import model
        return getattr(model, name)
    if name in ("normalize_matrices", "train_step"):
        from . import train
        return getattr(train, name)
    if name == "FusedAdamW":
        from . import optim
        return optim.FusedAdamW
    raise AttributeError(name)
