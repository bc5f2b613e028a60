"""Numeric check of the benchmarked shape itself (nViT-Base, B=128: M = 100 352, B*H = 1536): every GEMM shape and
epilogue of the step (NT plain / EPI 1-5, TN incl. the interleaved-row variant), attention forward/backward and the
LERP row kernel, against fp32/fp64 torch math on the GPU over rows sampled from every 256-row tile
(tools/fullshape_check.py; the same routine runs under `bench.py --check`)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_every_kernel_of_the_b128_step_at_full_shape():
    import fullshape_check
    rep = fullshape_check.check_all(B=128, T=784, C=768, H=12, verbose=True)
    bad = {k: v for k, v in rep.rows.items() if not v["ok"]}
    assert not bad, bad
    assert len(rep.rows) >= 40


def test_large_shapes_tile_walk():
    """Same routine at the Large (C4) per-GPU shape: B=64, C=1024, H=16 (M = 50 176, 1024-wide rows)."""
    import fullshape_check
    rep = fullshape_check.check_all(B=64, T=784, C=1024, H=16, verbose=True)
    bad = {k: v for k, v in rep.rows.items() if not v["ok"]}
    assert not bad, bad
