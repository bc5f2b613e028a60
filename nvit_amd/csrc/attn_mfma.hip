// impl 1: MFMA flash attention (bf16).  Placeholder until the kernels land: fails loudly.
#include "common.h"

int nvit_attn_fwd_mfma(const void*, const void*, const void*, float, void*, float*, int, int, int, int, int,
                       hipStream_t) {
  NVIT_FAIL(NVIT_EINVAL, "attn_fwd: MFMA kernel not built");
}
int nvit_attn_bwd_mfma(const void*, const void*, const void*, const void*, const float*, const float*, float, void*,
                       void*, void*, int, int, int, int, int, hipStream_t) {
  NVIT_FAIL(NVIT_EINVAL, "attn_bwd: MFMA kernel not built");
}
