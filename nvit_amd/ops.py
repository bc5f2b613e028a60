"""Thin Python wrappers over the C ABI (include/nvit_hip.h).

PyTorch is used here only as plumbing: device memory (caching allocator), the current HIP
stream, and dtype bookkeeping.  Every function enqueues hand-written HIP kernels on
`torch.cuda.current_stream()`; nothing falls back to torch operators.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import BF16, F32, check

Tensor = torch.Tensor


def tdtype(dt: int) -> torch.dtype:
    return torch.float32 if dt == F32 else torch.bfloat16


def dt_of(t: Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _s():
    return torch.cuda.current_stream().cuda_stream


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("nvit_amd ops need device tensors: the nViT hot path has no CPU fallback")


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def bk_of(dt: int) -> int:
    return 32 if dt == F32 else 64


# ----------------------------------------------------------------------------- GEMMs
def gemm_nt(A: Tensor, B: Tensor, M: int, N: int, K: int, out: Optional[Tensor] = None,
            out_dtype: torch.dtype = torch.float32, bias: Optional[Tensor] = None,
            colscale: Optional[Tensor] = None, rowadd: Optional[Tensor] = None, rowadd_period: int = 0,
            accumulate: bool = False, lda: Optional[int] = None, ldb: Optional[int] = None,
            ldc: Optional[int] = None) -> Tensor:
    """out[M,N] = A[M,K] @ B[N,K]^T (+epilogue). A,B same dtype (fp32 or bf16)."""
    _chk_dev(A, B)
    dt = dt_of(A)
    assert dt_of(B) == dt
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=out_dtype)
    ldc = out.stride(0) if ldc is None else ldc
    check(_lib.load().nvit_gemm_nt(dt, _p(A), lda, _p(B), ldb, _p(out), ldc, dt_of(out), M, N, K, _p(bias),
                                   _p(colscale), _p(rowadd), rowadd_period, int(accumulate), _s()), "nvit_gemm_nt")
    return out


FUSE_MIN_ELEMS = 256 * 256 * 64   # below this output size the fused persistent kernel is not worth a launch


def fusable(dt: int, M: int, N: int, K: int) -> bool:
    """True when the fused-epilogue persistent GEMM (nvit_gemm_nt_swiglu / _qknorm) can and should be used."""
    return bool(_lib.load().nvit_gemm_nt_fusable(dt, M, N, K)) and M * N >= FUSE_MIN_ELEMS


def gemm_nt_swiglu(A: Tensor, B: Tensor, M: int, F: int, K: int, gs: Optional[Tensor], gscale: float):
    """uv [M,2F] (raw, interleaved) and xm [M,F] = swiglu(uv) in one launch (bf16)."""
    _chk_dev(A, B)
    uv = torch.empty((M, 2 * F), device=A.device, dtype=torch.bfloat16)
    xm = torch.empty((M, F), device=A.device, dtype=torch.bfloat16)
    check(_lib.load().nvit_gemm_nt_swiglu(dt_of(A), _p(A), A.stride(0), _p(B), B.stride(0), _p(uv), _p(xm), M, F, K,
                                          _p(gs), gscale, _s()), "nvit_gemm_nt_swiglu")
    return uv, xm


def gemm_nt_swiglu_bwd(A: Tensor, B: Tensor, uv: Tensor, M: int, F: int, K: int, gs: Optional[Tensor], gscale: float):
    """duv [M,2F] (interleaved) and the d(suv) partials from dy [M,K] and W^T [F,K] in one launch (bf16)."""
    _chk_dev(A, B, uv)
    duv = torch.empty((M, 2 * F), device=A.device, dtype=torch.bfloat16)
    part = (torch.empty((2 * math.ceil(M / 256), 2 * F), device=A.device, dtype=torch.float32)
            if gs is not None else None)
    check(_lib.load().nvit_gemm_nt_swiglu_bwd(dt_of(A), _p(A), A.stride(0), _p(B), B.stride(0), _p(uv), _p(duv),
                                              _p(part), M, F, K, _p(gs), gscale, _s()), "nvit_gemm_nt_swiglu_bwd")
    return duv, part


def qk_buffers(dt: int, B: int, T: int, H: int, d: int, device):
    td = tdtype(dt)
    qh = torch.empty((B, H, T, d), device=device, dtype=td)
    return (qh, torch.empty_like(qh), torch.empty_like(qh),
            torch.empty((B * T, H), device=device, dtype=torch.float32),
            torch.empty((B * T, H), device=device, dtype=torch.float32))


LOG2E = 1.4426950408889634


def attn_q_prescale(d: int) -> float:
    """The factor that, folded into q_hat by its producer, turns the MFMA result into the score in log2 units:
    sqrt(d) * log2(e) - the exponent of the attention kernels then needs no multiply (nvit_attn_fwd_bounded)."""
    return math.sqrt(d) * LOG2E


def gemm_nt_qknorm(A: Tensor, B: Tensor, M: int, K: int, nparts: int, part0: int, sqk: Tensor, c_q: float, Bsz: int,
                   T: int, H: int, d: int, bufs=None, q_prescale: float = 1.0):
    """q/k/v projection(s) with the per-head normalise, sqk scale and head split fused (bf16, d=64).
    q_prescale: extra factor folded into the q part (pass the same value to attn_fwd / attn_bwd_qknorm)."""
    _chk_dev(A, B)
    if bufs is None:
        bufs = qk_buffers(dt_of(A), Bsz, T, H, d, A.device)
    qh, kh, vh, rq, rk = bufs
    check(_lib.load().nvit_gemm_nt_qknorm(dt_of(A), _p(A), A.stride(0), _p(B), B.stride(0), M, K, nparts, part0,
                                          _p(sqk), c_q, q_prescale, _p(qh), _p(kh), _p(vh), _p(rq), _p(rk), T, H, d,
                                          _s()), "nvit_gemm_nt_qknorm")
    return bufs


_ws_cache = {}


def _workspace(nbytes: int, device) -> Tensor:
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(max(nbytes // 4 + 1, 1 << 20), device=device, dtype=torch.float32)
        _ws_cache[key] = ws
    return ws


def tn_splits(Mred: int, N: int, K: int, dt: int) -> int:
    if N % 256 == 0 and K % 256 == 0 and Mred >= 4096:
        # persistent 256x256 kernel: (tiles x splits) work items dealt over 256 workgroups; pick the
        # split count that fills whole rounds best (fewest splits on ties: less slab traffic)
        tiles = (N // 256) * (K // 256)
        flops = 2.0 * Mred * N * K
        best, best_t = 1, float("inf")
        for s in range(1, 65):
            if Mred // s < 1024:
                break
            items = tiles * s
            eff = items / (math.ceil(items / 256) * 256)
            t = flops / (1.0e15 * eff) + 2.0 * s * N * K * 4 / 4.0e12  # MFMA time + slab write/read
            if t < best_t:
                best, best_t = s, t
        return best
    tiles = math.ceil(N / 128) * math.ceil(K / 128)
    rb = bk_of(dt)
    want = max(1, math.ceil(1024 / tiles))
    return max(1, min(want, math.ceil(Mred / (4 * rb)), 64))


def gemm_tn(A: Tensor, B: Tensor, G: Tensor, Mred: int, N: int, K: int, perm: int = 0, accumulate: bool = False,
            lda: Optional[int] = None, ldb: Optional[int] = None, ldg: Optional[int] = None) -> Tensor:
    """G[N,K] (+)= A[Mred,N]^T @ B[Mred,K]; G fp32."""
    _chk_dev(A, B, G)
    dt = dt_of(A)
    assert dt_of(B) == dt and G.dtype == torch.float32
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    ldg = G.stride(0) if ldg is None else ldg
    splits = tn_splits(Mred, N, K, dt)
    nbytes = splits * N * K * 4 + 256
    ws = _workspace(nbytes, A.device)
    check(_lib.load().nvit_gemm_tn(dt, _p(A), lda, _p(B), ldb, _p(G), ldg, Mred, N, K, splits, _p(ws),
                                   ws.numel() * 4, perm, int(accumulate), _s()), "nvit_gemm_tn")
    return G


# ----------------------------------------------------------------------------- row ops
_PART_BLOCKS_ENV = "NVIT_PART_BLOCKS" in os.environ
_LERP_BWD_BLOCKS: dict = {}
PART_BLOCKS = int(os.environ.get("NVIT_PART_BLOCKS", "1024"))   # workgroups of the backward row kernels (experiments: env)


def lerp_fwd(dt: int, h: Tensor, y: Tensor, alpha: Tensor, c_a: float, skip_x: Optional[Tensor] = None,
             skip: Optional[Tensor] = None, want_lo: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
    M, Cc = h.shape
    out = torch.empty_like(h)
    out_lo = torch.empty((M, Cc), device=h.device, dtype=tdtype(dt)) if want_lo else None
    check(_lib.load().nvit_lerp_fwd(dt, _p(h), _p(y), dt_of(y), _p(alpha), c_a, _p(skip_x), _p(skip), _p(out),
                                    _p(out_lo), M, Cc, _s()), "nvit_lerp_fwd")
    return out, out_lo


def lerp_bwd(dt: int, dout: Tensor, h: Tensor, y: Tensor, alpha: Tensor, c_a: float, skip_x: Optional[Tensor],
             skip: Optional[Tensor], dh: Optional[Tensor], accum_dh: bool, want_dy_f32: bool, want_dy_lo: bool,
             dout_add: Optional[Tensor] = None):
    """-> dh, dy_f32|None, dy_lo|None, dskip_x|None, part_dlam [4*nblk,C], part_dskip|None.
    dout_add: optional bf16 [M,C] added to dout inside the kernel (a data-gradient GEMM's output)."""
    if dout_add is not None and (dout_add.dtype != torch.bfloat16 or dout_add.shape != h.shape or not dout_add.is_contiguous()):
        raise ValueError("lerp_bwd: dout_add must be a contiguous bf16 [M,C] tensor")
    M, Cc = h.shape
    dev = h.device
    if dh is None:
        dh = torch.empty_like(h)
        accum_dh = False
    # grid = the workgroups resident at once for this kernel variant (a second, partly filled round of a 1024-block
    # grid cost 1.8 ms per Base step); NVIT_PART_BLOCKS overrides
    key = (dt, dt_of(y), Cc, dout_add is not None, skip_x is not None, bool(accum_dh))
    nres = _LERP_BWD_BLOCKS.get(key)
    if nres is None:
        nres = int(_lib.load().nvit_lerp_bwd_blocks(dt, dt_of(y), Cc, int(key[3]), int(key[4]), int(key[5]))) or PART_BLOCKS
        _LERP_BWD_BLOCKS[key] = nres
    nblk = min(PART_BLOCKS if _PART_BLOCKS_ENV else nres, math.ceil(M / 4))
    dy = torch.empty_like(h) if want_dy_f32 else None
    dy_lo = torch.empty((M, Cc), device=dev, dtype=tdtype(dt)) if want_dy_lo else None
    dskip_x = torch.empty_like(h) if skip_x is not None else None
    part = torch.empty((4 * nblk, Cc), device=dev, dtype=torch.float32)      # one row of column partials per wave
    pskip = torch.empty((4 * nblk,), device=dev, dtype=torch.float32) if skip_x is not None else None
    check(_lib.load().nvit_lerp_bwd(dt, _p(dout), _p(dout_add), _p(h), _p(y), dt_of(y), _p(alpha), c_a, _p(skip_x), _p(skip),
                                    _p(dh), int(accum_dh), _p(dy), _p(dy_lo), _p(dskip_x), _p(part), _p(pskip),
                                    nblk, M, Cc, _s()), "nvit_lerp_bwd")
    return dh, dy, dy_lo, dskip_x, part, pskip


def rmsnorm_fwd(x: Tensor, w: Tensor, eps: float):
    M, Cc = x.shape
    out = torch.empty_like(x)
    rstd = torch.empty((M,), device=x.device, dtype=torch.float32)
    check(_lib.load().nvit_rmsnorm_fwd(_p(x), _p(w), eps, _p(out), _p(rstd), M, Cc, _s()), "nvit_rmsnorm_fwd")
    return out, rstd


def rmsnorm_bwd(dout: Tensor, x: Tensor, w: Tensor, rstd: Tensor):
    """-> dx [M,C], dw [C]"""
    M, Cc = x.shape
    nblk = min(PART_BLOCKS, math.ceil(M / 4))
    dx = torch.empty_like(x)
    part = torch.empty((nblk, Cc), device=x.device, dtype=torch.float32)
    check(_lib.load().nvit_rmsnorm_bwd(_p(dout), _p(x), _p(w), _p(rstd), _p(dx), _p(part), nblk, M, Cc, _s()),
          "nvit_rmsnorm_bwd")
    dw = torch.empty((Cc,), device=x.device, dtype=torch.float32)
    colsum_reduce(part, dw, False)
    return dx, dw


def norm_skip_fwd(src: Tensor, tgt: Optional[Tensor], skip: Tensor) -> Tensor:
    """nrm(src*skip + tgt); tgt None = justnorm(src*skip)."""
    M, Cc = src.shape
    out = torch.empty_like(src)
    check(_lib.load().nvit_norm_skip_fwd(_p(src), _p(tgt), _p(skip), _p(out), M, Cc, _s()), "nvit_norm_skip_fwd")
    return out


def norm_skip_bwd(dout: Tensor, src: Tensor, tgt: Optional[Tensor], skip: Tensor):
    M, Cc = src.shape
    nblk = min(PART_BLOCKS, math.ceil(M / 4))
    dsrc = torch.empty_like(src)
    dtgt = torch.empty_like(src) if tgt is not None else None
    part = torch.empty((nblk,), device=src.device, dtype=torch.float32)
    check(_lib.load().nvit_norm_skip_bwd(_p(dout), _p(src), _p(tgt), _p(skip), _p(dsrc), _p(dtgt), _p(part), nblk, M,
                                         Cc, _s()), "nvit_norm_skip_bwd")
    dskip = torch.empty_like(skip)
    colsum_reduce(part, dskip, False)
    return dsrc, dtgt, dskip


def qknorm_fwd(dt: int, q: Tensor, ldq: int, k: Tensor, ldk: int, v: Tensor, ldv: int, sqk: Tensor, c_q: float,
               B: int, T: int, H: int, d: int):
    dev = sqk.device
    td = tdtype(BF16 if dt == _lib.BF16_F32IN else dt)   # BF16_F32IN: fp32 projection outputs in, bf16 head tensors out
    qh = torch.empty((B, H, T, d), device=dev, dtype=td)
    kh = torch.empty_like(qh)
    vh = torch.empty_like(qh)
    rq = torch.empty((B * T, H), device=dev, dtype=torch.float32)
    rk = torch.empty_like(rq)
    check(_lib.load().nvit_qknorm_fwd(dt, _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(sqk), c_q, _p(qh), _p(kh), _p(vh),
                                      _p(rq), _p(rk), B, T, H, d, _s()), "nvit_qknorm_fwd")
    return qh, kh, vh, rq, rk


def qknorm_bwd(dt: int, dqh, dkh, dvh, qh, kh, rq, rk, sqk, c_q: float, dq: Tensor, ldq: int, dk: Tensor, ldk: int,
               dv: Tensor, ldv: int, B: int, T: int, H: int, d: int) -> Tensor:
    nblk = min(PART_BLOCKS, math.ceil(B * T / 4))
    part = torch.empty((nblk, H * d), device=sqk.device, dtype=torch.float32)
    check(_lib.load().nvit_qknorm_bwd(dt, _p(dqh), _p(dkh), _p(dvh), _p(qh), _p(kh), _p(rq), _p(rk), _p(sqk), c_q,
                                      _p(dq), ldq, _p(dk), ldk, _p(dv), ldv, _p(part), nblk, B, T, H, d, _s()),
          "nvit_qknorm_bwd")
    return part


def swiglu_fwd(dt: int, uv: Tensor, suv: Optional[Tensor], gscale: float, M: int, F: int) -> Tensor:
    """dt = BF16_F32IN: fp32 pre-activations in, bf16 gated output."""
    x = torch.empty((M, F), device=uv.device, dtype=tdtype(BF16 if dt == _lib.BF16_F32IN else dt))
    check(_lib.load().nvit_swiglu_fwd(dt, _p(uv), _p(suv), gscale, _p(x), M, F, _s()), "nvit_swiglu_fwd")
    return x


def swiglu_bwd(dt: int, dx: Tensor, uv: Tensor, suv: Optional[Tensor], gscale: float, M: int, F: int):
    duv = torch.empty((M, 2 * F), device=uv.device, dtype=tdtype(dt))
    colblocks = math.ceil(F / 1024)
    rowblocks = max(1, min(M, math.ceil(2048 / colblocks)))
    rpb = math.ceil(M / rowblocks)
    nrb = math.ceil(M / rpb)
    part = torch.empty((nrb, 2 * F), device=uv.device, dtype=torch.float32) if suv is not None else None
    check(_lib.load().nvit_swiglu_bwd(dt, _p(dx), _p(uv), _p(suv), gscale, _p(duv), _p(part), rpb, M, F, _s()),
          "nvit_swiglu_bwd")
    return duv, part


def colsum_reduce(part: Tensor, out: Tensor, accumulate: bool, kind: int = 0, ref: Optional[Tensor] = None,
                  scale: float = 1.0) -> None:
    nblk, N = part.shape if part.dim() == 2 else (part.shape[0], 1)
    check(_lib.load().nvit_colsum_reduce(_p(part), nblk, N, _p(out), int(accumulate), kind, _p(ref), scale, _s()),
          "nvit_colsum_reduce")


class ReduceBatch:
    """Column-partial reductions collected during one backward node and issued as ONE launch (`flush`): items are
    (part [rows, N], out [N], accumulate, kind, ref, scale) as for `colsum_reduce`, optionally with a second partial
    array that is reduced, scaled and added after the first (`part_b`)."""

    def __init__(self) -> None:
        self.items = []

    def add(self, part: Tensor, out: Tensor, accumulate: bool = False, kind: int = 0, ref: Optional[Tensor] = None,
            scale: float = 1.0, part_b: Optional[Tensor] = None) -> None:
        self.items.append((part, out, accumulate, kind, ref, scale, part_b))
        if len(self.items) == 8:
            self.flush()

    def flush(self) -> None:
        items, self.items = self.items, []
        n = len(items)
        if n == 0:
            return
        if n == 1 and items[0][6] is None:
            part, out, acc, kind, ref, scale, _ = items[0]
            colsum_reduce(part, out, acc, kind, ref, scale)
            return
        I64, I32, F32_ = C.c_int64 * n, C.c_int * n, C.c_float * n
        rows = lambda t: t.shape[0]
        cols = lambda t: t.shape[1] if t.dim() == 2 else 1
        check(_lib.load().nvit_colsum_reduce_multi(
            I64(*[it[0].data_ptr() for it in items]), I32(*[rows(it[0]) for it in items]),
            I64(*[(it[6].data_ptr() if it[6] is not None else 0) for it in items]),
            I32(*[(rows(it[6]) if it[6] is not None else 0) for it in items]),
            I32(*[cols(it[0]) for it in items]), I64(*[it[1].data_ptr() for it in items]),
            I32(*[int(it[2]) for it in items]), I32(*[it[3] for it in items]),
            I64(*[(it[4].data_ptr() if it[4] is not None else 0) for it in items]), F32_(*[float(it[5]) for it in items]),
            n, _s()), "nvit_colsum_reduce_multi")


def colsum(a: Tensor, R: int, N: int, out: Tensor, accumulate: bool, b: Optional[Tensor] = None, period: int = 0,
           scale: float = 1.0, lda: Optional[int] = None, ldb: Optional[int] = None) -> None:
    lda = a.stride(0) if lda is None else lda
    ldb = (b.stride(0) if ldb is None else ldb) if b is not None else 0
    check(_lib.load().nvit_colsum(_p(a), dt_of(a), lda, _p(b), dt_of(b) if b is not None else F32, ldb, R, N,
                                  period, _p(out), int(accumulate), scale, _s()), "nvit_colsum")


def colsum_big(a: Tensor, R: int, N: int, out: Tensor, accumulate: bool, lda: Optional[int] = None) -> None:
    """column sums over many rows: fold rows modulo 512 first, then reduce the 512 partial rows."""
    if R <= 1024:
        colsum(a, R, N, out, accumulate, lda=lda)
        return
    part = torch.empty((512, N), device=a.device, dtype=torch.float32)
    colsum(a, R, N, part, False, period=512, lda=lda)
    colsum(part, 512, N, out, accumulate)


def cast(src: Tensor, dt: int) -> Tensor:
    dst = torch.empty(src.shape, device=src.device, dtype=tdtype(dt))
    check(_lib.load().nvit_cast(_p(src), _p(dst), dt, src.numel(), _s()), "nvit_cast")
    return dst


def normalize_images(x: Tensor, mean: float = 0.5, std: float = 0.5) -> Tensor:
    """ToTensor + Normalize(mean, std) on the device (reference train.py:1084-1090, the deterministic part of its input
    pipeline): uint8 [B,H,W,C] (scaled by 1/255) or fp32 [B,C,H,W] in [0,1] -> fp32 [B,C,H,W]."""
    _chk_dev(x)
    x = x.contiguous()
    if x.dtype == torch.uint8:
        B, H, W, Cc = x.shape
        u8 = 1
    elif x.dtype == torch.float32:
        B, Cc, H, W = x.shape
        u8 = 0
    else:
        raise TypeError("normalize_images: uint8 [B,H,W,C] or float32 [B,C,H,W]")
    out = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32)
    check(_lib.load().nvit_normalize_images(_p(x), u8, _p(out), B, Cc, H, W, mean, std, _s()), "nvit_normalize_images")
    return out


def scale_cols(a: Tensor, s: Tensor, c: float, R: int, N: int, out: Tensor, lda: Optional[int] = None,
               ldo: Optional[int] = None) -> Tensor:
    lda = a.stride(0) if lda is None else lda
    ldo = out.stride(0) if ldo is None else ldo
    check(_lib.load().nvit_scale_cols(_p(a), lda, _p(s), c, _p(out), dt_of(out), ldo, R, N, _s()), "nvit_scale_cols")
    return out


# ----------------------------------------------------------------------------- attention
def attn_fwd(dt: int, impl: int, qh: Tensor, kh: Tensor, vh: Tensor, scale: float, sqk: Optional[Tensor] = None,
             c_q: float = 0.0, q_prescale: float = 1.0):
    """sqk/c_q given: q and k are (sqk*c_q) * unit vectors per head (the nViT call sites) -> bounded-score kernel path;
    q_prescale: qh holds q_prescale * q_hat (only with sqk)."""
    B, H, Tq, d = qh.shape
    Tk = kh.shape[2]
    o = torch.empty((B * Tq, H * d), device=qh.device, dtype=tdtype(dt))
    lse = torch.empty((B, H, Tq), device=qh.device, dtype=torch.float32)
    if sqk is not None:
        check(_lib.load().nvit_attn_fwd_bounded(dt, impl, _p(qh), _p(kh), _p(vh), scale, _p(sqk), c_q, q_prescale, _p(o),
                                                _p(lse), B, H, Tq, Tk, d, _s()), "nvit_attn_fwd_bounded")
    else:
        assert q_prescale == 1.0
        check(_lib.load().nvit_attn_fwd(dt, impl, _p(qh), _p(kh), _p(vh), scale, _p(o), _p(lse), B, H, Tq, Tk, d, _s()),
              "nvit_attn_fwd")
    return o, lse


def attn_bwd(dt: int, impl: int, dout: Tensor, qh: Tensor, kh: Tensor, vh: Tensor, o: Tensor, lse: Tensor,
             scale: float):
    B, H, Tq, d = qh.shape
    Tk = kh.shape[2]
    dqh = torch.empty_like(qh)
    dkh = torch.empty_like(kh)
    dvh = torch.empty_like(vh)
    delta = torch.empty((2, B, H, Tq), device=qh.device, dtype=torch.float32)   # workspace: -delta | -lse*log2(e)
    check(_lib.load().nvit_attn_bwd(dt, impl, _p(dout), _p(qh), _p(kh), _p(vh), _p(o), _p(lse), scale, _p(dqh),
                                    _p(dkh), _p(dvh), _p(delta), B, H, Tq, Tk, d, _s()), "nvit_attn_bwd")
    return dqh, dkh, dvh


def attn_bwd_qknorm(dout: Tensor, qh: Tensor, kh: Tensor, vh: Tensor, o: Tensor, lse: Tensor, scale: float,
                    rq: Tensor, rk: Tensor, sqk: Tensor, c_q: float, dq: Tensor, ldq: int, dk: Tensor, dv: Tensor,
                    ldkv: int, q_prescale: float = 1.0):
    """MFMA attention backward + q/k-normalise backward in one pass (bf16, d=64).
    Writes dq/dk/dv token-major; returns the partial sums (part_q, part_k) of d/d(sqk*c_q)."""
    B, H, Tq, d = qh.shape
    Tk = kh.shape[2]
    dev = qh.device
    part_q = torch.empty((B * math.ceil(Tq / 128), H * d), device=dev, dtype=torch.float32)
    part_k = torch.empty((B * math.ceil(Tk / 128), H * d), device=dev, dtype=torch.float32)
    delta = torch.empty((2, B, H, Tq), device=dev, dtype=torch.float32)   # workspace: -delta | -lse*log2(e)
    check(_lib.load().nvit_attn_bwd_qknorm(BF16, _p(dout), _p(qh), _p(kh), _p(vh), _p(o), _p(lse), scale, _p(rq),
                                           _p(rk), _p(sqk), c_q, q_prescale, _p(dq), ldq, _p(dk), _p(dv), ldkv, _p(part_q),
                                           _p(part_k), _p(delta), B, H, Tq, Tk, d, _s()), "nvit_attn_bwd_qknorm")
    return part_q, part_k


# ----------------------------------------------------------------------------- embed / head
def im2col(dt: int, img: Tensor, Pl: int, Pg: int):
    B, ch, S, _ = img.shape
    T = (S // Pl) ** 2
    td = tdtype(dt)
    A_l = torch.empty((B * T, ch * Pl * Pl), device=img.device, dtype=td)
    A_g = torch.empty((B * T, ch * Pg * Pg), device=img.device, dtype=td)
    check(_lib.load().nvit_im2col(dt, _p(img), _p(A_l), _p(A_g), B, ch, S, Pl, Pg, _s()), "nvit_im2col")
    return A_l, A_g


def patch_kp(K: int) -> int:
    """patch length padded to whole 32-element stages of nvit_patch_embed_fwd"""
    return round_up(K, 32)


def patch_embed_fwd(img: Tensor, w_l: Tensor, b_l: Optional[Tensor], pos_l: Tensor, w_g: Tensor, b_g: Optional[Tensor],
                    pos_g: Tensor, Pl: int, Pg: int, Cc: int, save_rows: bool = True, twins: bool = False):
    """Fused dual patch embedding of the bf16 mode -> loc, glo fp32 [M, C], (twins) their bf16 copies, and (save_rows)
    the bf16 patch rows a_l [Mpad, Kp_l], a_g [Mpad, Kp_g] for the weight gradients.
    w_l / w_g: split images [C, 2*Kp] (shadow perm 2).  Returns (loc, glo, a_l, a_g, loc_lo, glo_lo)."""
    B, ch, S, _ = img.shape
    T = (S // Pl) ** 2
    M = B * T
    Kpl, Kpg = patch_kp(ch * Pl * Pl), patch_kp(ch * Pg * Pg)
    assert w_l.shape == (Cc, 2 * Kpl) and w_g.shape == (Cc, 2 * Kpg) and w_l.dtype == torch.bfloat16
    assert pos_l.shape == (T, Cc) and pos_g.shape == (T, Cc) and img.dtype == torch.float32 and img.is_contiguous()
    _chk_dev(img, w_l, w_g, pos_l, pos_g)
    dev = img.device
    loc = torch.empty((M, Cc), device=dev, dtype=torch.float32)
    glo = torch.empty((M, Cc), device=dev, dtype=torch.float32)
    a_l = a_g = lo_l = lo_g = None
    if save_rows:
        Mpad = round_up(M, 256)
        a_l = torch.empty((Mpad, Kpl), device=dev, dtype=torch.bfloat16)
        a_g = torch.empty((Mpad, Kpg), device=dev, dtype=torch.bfloat16)
    if twins:
        lo_l = torch.empty((M, Cc), device=dev, dtype=torch.bfloat16)
        lo_g = torch.empty((M, Cc), device=dev, dtype=torch.bfloat16)
    check(_lib.load().nvit_patch_embed_fwd(_p(img), _p(w_l), _p(b_l), _p(pos_l), _p(loc), _p(lo_l), _p(a_l), _p(w_g),
                                           _p(b_g), _p(pos_g), _p(glo), _p(lo_g), _p(a_g), B, ch, S, Pl, Pg, Cc, _s()),
          "nvit_patch_embed_fwd")
    return loc, glo, a_l, a_g, lo_l, lo_g


def pool_ln_fwd(dt: int, x: Tensor, w: Tensor, b: Tensor, eps: float, B: int, T: int, Cc: int):
    dev = x.device
    nchunk = max(1, min(T, math.ceil(1024 / B)))
    pooled = torch.empty((B, Cc), device=dev, dtype=torch.float32)
    ln = torch.empty_like(pooled)
    ln_lo = torch.empty((B, Cc), device=dev, dtype=tdtype(dt))
    stats = torch.empty((B, 2), device=dev, dtype=torch.float32)
    ws = torch.empty((B, nchunk, Cc), device=dev, dtype=torch.float32)
    check(_lib.load().nvit_pool_ln_fwd(dt, _p(x), _p(w), _p(b), eps, _p(pooled), _p(ln), _p(ln_lo), _p(stats), _p(ws),
                                       nchunk, B, T, Cc, _s()), "nvit_pool_ln_fwd")
    return pooled, ln, ln_lo, stats


def pool_ln_bwd(dln: Tensor, pooled: Tensor, w: Tensor, stats: Tensor, dw: Tensor, db: Tensor, accumulate: bool,
                B: int, T: int, Cc: int) -> Tensor:
    dx = torch.empty((B * T, Cc), device=dln.device, dtype=torch.float32)
    check(_lib.load().nvit_pool_ln_bwd(_p(dln), _p(pooled), _p(w), _p(stats), _p(dx), _p(dw), _p(db),
                                       int(accumulate), B, T, Cc, _s()), "nvit_pool_ln_bwd")
    return dx


def recon_loss(raw: Tensor, img: Tensor, P: int) -> Tensor:
    B, ch, S, _ = img.shape
    nblk = 1024
    part = torch.empty((nblk,), device=img.device, dtype=torch.float32)
    loss = torch.empty((1,), device=img.device, dtype=torch.float32)
    check(_lib.load().nvit_recon_loss(_p(raw), _p(img), _p(part), nblk, _p(loss), B, ch, S, P, _s()),
          "nvit_recon_loss")
    return loss.reshape(())


# ----------------------------------------------------------------------------- Kohonen head (config C5)
def som_bmu(x: Tensor, nodes: Tensor) -> Tensor:
    """x [M,C] fp32, nodes [N,C] fp32 -> idx [M] int64 (exact-f32 MFMA score GEMM + argmin kernel)."""
    M, Cc = x.shape
    N = nodes.shape[0]
    scores = gemm_nt(x, nodes, M, N, Cc)                      # fp32 operands -> v_mfma_f32_16x16x4_f32
    nn_ws = torch.empty((N,), device=x.device, dtype=torch.float32)
    idx = torch.empty((M,), device=x.device, dtype=torch.int64)
    check(_lib.load().nvit_som_bmu(_p(scores), _p(nodes), _p(nn_ws), M, N, Cc, _p(idx), _s()), "nvit_som_bmu")
    return idx


def gather_rows(nodes: Tensor, idx: Tensor) -> Tensor:
    M, Cc = idx.numel(), nodes.shape[1]
    out = torch.empty((M, Cc), device=nodes.device, dtype=torch.float32)
    check(_lib.load().nvit_gather_rows(_p(nodes), _p(idx), _p(out), M, Cc, _s()), "nvit_gather_rows")
    return out


def scatter_rows(dout: Tensor, idx: Tensor, N: int) -> Tensor:
    """dnodes[n] = sum of dout rows with idx == n.  All tokens may pick the same node (they do at init: the BMU of a
    small-norm patch is the smallest-norm node), so instead of a per-node loop this is onehot(idx)^T . dout on the
    exact-f32 weight-gradient GEMM: load balanced and deterministic whatever the histogram."""
    M, Cc = dout.shape
    dn = torch.empty((N, Cc), device=dout.device, dtype=torch.float32)
    if N % 4 != 0 or Cc % 4 != 0:
        check(_lib.load().nvit_scatter_rows(_p(dout), _p(idx), _p(dn), M, N, Cc, _s()), "nvit_scatter_rows")
        return dn
    oh = torch.empty((M, N), device=dout.device, dtype=torch.float32)
    check(_lib.load().nvit_onehot(_p(idx), _p(oh), M, N, _s()), "nvit_onehot")
    return gemm_tn(oh, dout.float().contiguous(), dn, M, N, Cc)


def som_update(nodes: Tensor, x: Tensor, idx: Tensor, lr_alpha: float, sigma: float, gm: int, gn: int, B: int,
               T: int, periodic: bool = True) -> None:
    Cc = nodes.shape[1]
    v_ws = torch.empty((B, Cc), device=nodes.device, dtype=torch.float32)
    s_ws = torch.empty((B, gm * gn), device=nodes.device, dtype=torch.float32)
    check(_lib.load().nvit_som_update(_p(nodes), _p(x), _p(idx), lr_alpha, sigma, gm, gn, int(periodic), _p(v_ws), _p(s_ws),
                                      B, T, Cc, _s()), "nvit_som_update")


def cos_consistency_fwd(a: Tensor, b: Tensor):
    M, Cc = a.shape
    nblk = min(1024, math.ceil(M / 4))
    stats = torch.empty((M, 3), device=a.device, dtype=torch.float32)
    part = torch.empty((nblk,), device=a.device, dtype=torch.float32)
    loss = torch.empty((1,), device=a.device, dtype=torch.float32)
    check(_lib.load().nvit_cos_consistency_fwd(_p(a), _p(b), _p(stats), _p(part), nblk, _p(loss), M, Cc, _s()),
          "nvit_cos_consistency_fwd")
    return loss.reshape(()), stats


def cos_consistency_bwd(a: Tensor, b: Tensor, stats: Tensor, g: Tensor):
    M, Cc = a.shape
    da, db = torch.empty_like(a), torch.empty_like(b)
    check(_lib.load().nvit_cos_consistency_bwd(_p(a), _p(b), _p(stats), _p(g), _p(da), _p(db), M, Cc, _s()),
          "nvit_cos_consistency_bwd")
    return da, db


def huber_fwd(a: Tensor, b: Tensor) -> Tensor:
    nblk = 1024
    part = torch.empty((nblk,), device=a.device, dtype=torch.float32)
    loss = torch.empty((1,), device=a.device, dtype=torch.float32)
    check(_lib.load().nvit_huber_fwd(_p(a), _p(b), _p(part), nblk, _p(loss), a.numel(), _s()), "nvit_huber_fwd")
    return loss.reshape(())


def huber_bwd(a: Tensor, b: Tensor, g: Tensor):
    da, db = torch.empty_like(a), torch.empty_like(b)
    check(_lib.load().nvit_huber_bwd(_p(a), _p(b), _p(g), _p(da), _p(db), a.numel(), _s()), "nvit_huber_bwd")
    return da, db


def som_smooth_fwd(nodes: Tensor, idx: Tensor, map_size: int):
    Nn, Cc = nodes.shape
    cnt = torch.empty((Nn,), device=nodes.device, dtype=torch.int32)
    D = torch.empty((Nn, 8), device=nodes.device, dtype=torch.float32)
    loss = torch.empty((1,), device=nodes.device, dtype=torch.float32)
    check(_lib.load().nvit_som_smooth_fwd(_p(nodes), _p(idx), _p(cnt), _p(D), _p(loss), idx.numel(), Nn, Cc, map_size,
                                          _s()), "nvit_som_smooth_fwd")
    return loss.reshape(()), cnt, D


def som_smooth_bwd(nodes: Tensor, D: Tensor, cnt: Tensor, g: Tensor, M: int, map_size: int) -> Tensor:
    Nn, Cc = nodes.shape
    dn = torch.empty_like(nodes)
    check(_lib.load().nvit_som_smooth_bwd(_p(nodes), _p(D), _p(cnt), _p(g), _p(dn), 0, M, Nn, Cc, map_size, _s()),
          "nvit_som_smooth_bwd")
    return dn


def recon_bwd(dt: int, raw: Tensor, img: Tensor, g: Tensor, P: int) -> Tensor:
    B, ch, S, _ = img.shape
    draw = torch.empty(raw.shape, device=raw.device, dtype=tdtype(dt))
    check(_lib.load().nvit_recon_bwd(dt, _p(raw), _p(img), _p(g), _p(draw), B, ch, S, P, _s()), "nvit_recon_bwd")
    return draw


# ----------------------------------------------------------------------------- weights
def renorm_table(mats, device) -> Tuple[Tensor, int]:
    """mats: list of (tensor fp32 [rows, cols] contiguous, dim). -> device table, total_items"""
    rows_, first = [], 0
    for w, dim in mats:
        assert w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 2
        r, c = w.shape
        if dim == 0 and r > 1152:
            raise RuntimeError(f"renorm: column-normalised matrix with {r} rows exceeds the LDS slab (1152)")
        items = math.ceil(r / _lib.RENORM_ROWS_PER_ITEM) if dim == 1 else math.ceil(c / _lib.RENORM_COLS_PER_ITEM)
        rows_.append([w.data_ptr(), r, c, dim, first])
        first += items
    return torch.tensor(rows_, dtype=torch.int64).to(device), first


def renorm_weights(table: Tensor, total_items: int) -> None:
    check(_lib.load().nvit_renorm_weights(_p(table), table.shape[0], total_items, _s()), "nvit_renorm_weights")


def shadow_table(entries, device) -> Tuple[Tensor, int]:
    """entries: list of (src fp32 2-D view, dst|None, dst_ld, dst_cols, dstT|None, dstT_ld, dstT_cols, perm)."""
    rows_, first = [], 0
    for src, dst, dst_ld, dst_cols, dstT, dstT_ld, dstT_cols, perm in entries:
        assert src.dtype == torch.float32 and src.is_contiguous()
        r, c = src.shape[0], src.numel() // src.shape[0]
        tiles_c = math.ceil(max(c, dst_cols if dst is not None else 0) / 64)
        tiles_r = math.ceil(max(r, dstT_cols if dstT is not None else 0) / 64)
        rows_.append([src.data_ptr(), r, c, dst.data_ptr() if dst is not None else 0, dst_ld, dst_cols,
                      dstT.data_ptr() if dstT is not None else 0, dstT_ld, dstT_cols, perm, first, tiles_c])
        first += tiles_c * tiles_r
    return torch.tensor(rows_, dtype=torch.int64).to(device), first


def shadow_weights(table: Tensor, total_items: int, dt: int) -> None:
    check(_lib.load().nvit_shadow_weights(_p(table), table.shape[0], total_items, dt, _s()), "nvit_shadow_weights")


# ----------------------------------------------------------------------------- profiling
def prof_enable(on: bool) -> None:
    _lib.load().nvit_prof_enable(int(on))


def prof_select(*families: str) -> None:
    """time only the launches of the named kernel families (names: _lib.KID_NAMES)"""
    mask = 0
    for f in families:
        mask |= 1 << _lib.KID_NAMES.index(f)
    _lib.load().nvit_prof_select(mask)


def prof_collect():
    n = len(_lib.KID_NAMES)
    ms = (C.c_double * n)()
    fl = (C.c_double * n)()
    by = (C.c_double * n)()
    ln = (C.c_int64 * n)()
    check(_lib.load().nvit_prof_collect(ms, fl, by, ln), "nvit_prof_collect")
    return {_lib.KID_NAMES[i]: {"ms": ms[i], "flops": fl[i], "bytes": by[i], "launches": ln[i]} for i in range(n)}
