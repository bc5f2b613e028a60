"""Import shim: lets the reference trainer run UNCHANGED on the MI355X path.

The reference builds its model with `from nvit.model import ViT, ViTConfig` (nvit/train.py:37) and
`from nvit.kohonen import KohonenMap` (nvit/model.py:10).  With this repository placed BEFORE the reference on
PYTHONPATH, `nvit.model` / `nvit.kohonen` resolve to the modules below (re-exports of `nvit_amd`), while every other
submodule (`nvit.train`, `nvit.debug`, ...) still resolves to the reference's own files: this package extends its
search path over all `nvit/` directories on sys.path.

    PYTHONPATH=/path/to/this/repo:/path/to/reference python -m nvit.train

Nothing here computes anything; the product code lives in `nvit_amd/`.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
