"""ctypes binding of libnvit_hip.so (the C ABI declared in include/nvit_hip.h).

The product path has NO CPU or PyTorch fallback: if the library is missing or a call
fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (NVIT_LIB: experiments with an alternative build of the same sources, e.g. tools/overlap_probe.py; never set in product use)
LIB_PATH = os.environ.get("NVIT_LIB") or os.path.join(_HERE, "libnvit_hip.so")

F32, BF16, BF16_F32IN = 0, 1, 3
KID_NAMES = ["gemm_nt", "gemm_tn", "attn_fwd", "attn_bwd", "rowops", "renorm", "shadow", "patchify", "misc", "gemm_f32",
             "gemm_swiglu", "gemm_qknorm", "gemm_swiglu_bwd", "optim"]
RENORM_ROWS_PER_ITEM = 64
RENORM_COLS_PER_ITEM = 64

_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "nvit_version": [],
    "nvit_last_error": [],
    "nvit_prof_enable": [_i],
    "nvit_prof_select": [C.c_uint],
    "nvit_prof_collect": [_vp, _vp, _vp, _vp],
    "nvit_prof_name": [_i],
    "nvit_renorm_weights": [_vp, _i, _i, _vp],
    "nvit_shadow_weights": [_vp, _i, _i, _i, _vp],
    "nvit_grad_sqnorm": [_vp, _i, _i, _vp, _i, _vp],
    "nvit_adamw_renorm": [_vp, _i, _i, _i, _f, _f, _f, C.c_double, C.c_double, _vp, _i, _f, _vp, _vp, _vp],
    "nvit_adamw_tick": [_vp, C.c_double, C.c_double, _vp],
    "nvit_set_gemm_sched": [_i],
    "nvit_set_gemm_impl": [_i, _i],
    "nvit_set_tn_order": [_i],
    "nvit_ce_loss": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "nvit_gemm_nt": [_i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp],
    "nvit_gemm_nt_fusable": [_i, _i, _i, _i],
    "nvit_gemm_nt_swiglu": [_i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _f, _vp],
    "nvit_gemm_nt_swiglu_bwd": [_i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _f, _vp],
    "nvit_gemm_nt_qknorm": [_i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "nvit_gemm_tn": [_i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _i64, _i, _i, _vp],
    "nvit_lerp_fwd": [_i, _vp, _vp, _i, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "nvit_lerp_bwd_blocks": [_i, _i, _i, _i, _i, _i],
    "nvit_lerp_bwd": [_i, _vp, _vp, _vp, _vp, _i, _vp, _f, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "nvit_norm_skip_fwd": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "nvit_norm_skip_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "nvit_qknorm_fwd": [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "nvit_qknorm_bwd": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i,
                        _i, _i, _vp],
    "nvit_swiglu_fwd": [_i, _vp, _vp, _f, _vp, _i, _i, _vp],
    "nvit_swiglu_bwd": [_i, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _vp],
    "nvit_colsum_reduce_multi": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "nvit_colsum_reduce": [_vp, _i, _i, _vp, _i, _i, _vp, _f, _vp],
    "nvit_colsum": [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _f, _vp],
    "nvit_cast": [_vp, _vp, _i, _i64, _vp],
    "nvit_normalize_images": [_vp, _i, _vp, _i, _i, _i, _i, _f, _f, _vp],
    "nvit_scale_cols": [_vp, _i, _vp, _f, _vp, _i, _i, _i, _i, _vp],
    "nvit_attn_fwd": [_i, _i, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "nvit_attn_fwd_bounded": [_i, _i, _vp, _vp, _vp, _f, _vp, _f, _f, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "nvit_attn_bwd": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "nvit_attn_bwd_qknorm": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _f, _f, _vp, _i, _vp, _vp, _i, _vp, _vp,
                             _vp, _i, _i, _i, _i, _i, _vp],
    "nvit_im2col": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "nvit_patch_embed_kp": [_i],
    "nvit_patch_embed_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "nvit_pool_ln_fwd": [_i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "nvit_pool_ln_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "nvit_som_bmu": [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp],
    "nvit_gather_rows": [_vp, _vp, _vp, _i64, _i, _vp],
    "nvit_scatter_rows": [_vp, _vp, _vp, _i64, _i, _i, _vp],
    "nvit_onehot": [_vp, _vp, _i64, _i, _vp],
    "nvit_rmsnorm_fwd": [_vp, _vp, _f, _vp, _vp, _i, _i, _vp],
    "nvit_rmsnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "nvit_som_update": [_vp, _vp, _vp, _f, _f, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp],
    "nvit_cos_consistency_fwd": [_vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _vp],
    "nvit_cos_consistency_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp],
    "nvit_huber_fwd": [_vp, _vp, _vp, _i, _vp, _i64, _vp],
    "nvit_huber_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "nvit_som_smooth_fwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "nvit_som_smooth_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp],
    "nvit_recon_bwd": [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "nvit_recon_loss": [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp],
    "nvit_xgmi_chunk": [_i64, _i],
    "nvit_xgmi_reduce_scatter": [_vp, _i, _i, _i64, _f, _vp],
    "nvit_xgmi_all_gather": [_vp, _i, _i, _i64, _vp],
    "nvit_xgmi_flag_bytes": [_i],
    "nvit_xgmi_flags_alloc": [_i, _vp, _vp],
    "nvit_xgmi_flags_open": [_vp, _vp],
    "nvit_xgmi_flags_close": [_vp],
    "nvit_xgmi_flags_free": [_vp],
    "nvit_xgmi_flags_error": [_vp, _i, _vp, _vp],
    "nvit_xgmi_reduce_scatter_sync": [_vp, _vp, _i, _i, _i, _i, C.c_uint, _i64, _i64, _f, _vp],
    "nvit_xgmi_all_gather_sync": [_vp, _vp, _i, _i, _i, _i, C.c_uint, _i64, _i64, _vp],
    "nvit_xgmi_wait_gathered": [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp],
    "nvit_xgmi_set_timeout": [C.c_double],
    "nvit_xgmi_errword_alloc": [_vp, _vp],
    "nvit_xgmi_errword_free": [_vp],
    "nvit_set_attn_dkv_asm": [_i],
}
_RESTYPES = {"nvit_last_error": C.c_char_p, "nvit_prof_name": C.c_char_p, "nvit_prof_enable": None, "nvit_prof_select": None,
             "nvit_xgmi_chunk": C.c_int64, "nvit_xgmi_flag_bytes": C.c_int64}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raise loudly when it is absent (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"nvit_amd: {LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the nViT hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().nvit_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
