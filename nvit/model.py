"""`nvit.model` of the reference (nvit/model.py), served by the MI355X implementation: same names, constructor and
forward() signatures, parameter names and shapes.  See nvit_amd/model.py."""
from nvit_amd.config import ViTConfig  # noqa: F401
from nvit_amd.kohonen import KohonenMap  # noqa: F401
from nvit_amd.model import RMSNorm, Block, CrossAttentionBlock, ViT, justnorm  # noqa: F401
