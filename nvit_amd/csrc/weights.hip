// Weight maintenance kernels.
//
// renorm_kernel: Trainer.normalize_matrices (/root/reference/nvit/train.py:461-480) as ONE
// persistent launch over a device-side table of matrices.  fp32, in place, algorithmic
// minimum traffic (one HBM read + one write per element):
//   dim=1 (row norms, query/key/value/c_fc): one wave per row; the second pass over the row
//         (<= 16 KiB) is served by L1/L2.
//   dim=0 (column norms, att_c_proj/mlp_c_proj): a workgroup keeps a [rows x 64-column] panel
//         in registers (rows <= 1152), so the column pass needs no re-read.
//
// shadow_kernel: builds the private MFMA-operand copies of the fp32 masters: W (optionally
// row-permuted for the SwiGLU interleave) and W^T, cast to bf16 (or kept fp32 for the exact
// mode), zero padded to the requested leading dimensions.
#include "common.h"

namespace {

// row of the device table whose work-item range contains `item` (first_item column `col`, `stride` int64 per row):
// binary search - a linear scan costs one dependent scalar load per matrix (72 for Base) in front of every item
__device__ __forceinline__ int table_row(const int64_t* table, int n, int stride, int col, int item) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)table[mid * stride + col] <= item) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// 1024-thread workgroups (16 waves per CU keep ~48 KiB of 16-byte loads in flight, which is what an HBM-bound
// stream needs on this chip).  dim=1: one wave per row, the row stays in registers between the norm and the
// scaled store (one read + one write per element, no second pass).  dim=0: [rows x 64] column panel in registers.
constexpr int RENORM_THREADS = 1024;
constexpr int RENORM_CL = NVIT_RENORM_COLS_PER_ITEM / 4;   // lanes across a panel row (16-byte pieces)
constexpr int RENORM_RG = RENORM_THREADS / RENORM_CL;   // row groups of the column pass
constexpr int RENORM_NV = 8;                    // float4 per lane held in registers: rows of up to 2048 columns
constexpr int RENORM_NR = 1152 / RENORM_RG;      // rows per thread of a column panel: matrices of up to 1152 rows

__global__ __launch_bounds__(RENORM_THREADS) void renorm_kernel(const int64_t* table, int n, int total_items) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int item = blockIdx.x; item < total_items; item += gridDim.x) {
    const int mi = table_row(table, n, 5, 4, item);
    float* W = reinterpret_cast<float*>(table[mi * 5 + 0]);
    const int rows = (int)table[mi * 5 + 1], cols = (int)table[mi * 5 + 2], dim = (int)table[mi * 5 + 3];
    const int local = item - (int)table[mi * 5 + 4];
    if (dim == 1) {
      for (int r = local * NVIT_RENORM_ROWS_PER_ITEM + wid; r < rows && r < (local + 1) * NVIT_RENORM_ROWS_PER_ITEM;
           r += RENORM_THREADS / 64) {
        float* row = W + (size_t)r * cols;
        if ((cols & 3) == 0 && cols <= RENORM_NV * 256) {
          f32x4 v[RENORM_NV];
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < RENORM_NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            v[i] = c < cols ? *reinterpret_cast<const f32x4*>(row + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
            s += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
          }
          s = wave_sum(s);
          const float nrm = sqrtf(s);
#pragma unroll
          for (int i = 0; i < RENORM_NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < cols) *reinterpret_cast<f32x4*>(row + c) = v[i] / nrm;
          }
        } else {
          float s = 0.f;
          for (int c = lane; c < cols; c += 64) s += row[c] * row[c];
          s = wave_sum(s);
          const float nrm = sqrtf(s);
          for (int c = lane; c < cols; c += 64) row[c] = row[c] / nrm;
        }
      }
    } else {
      // [rows x 64]-column panel (256-byte row segments) held in REGISTERS (thread = 4 columns x every 64th row):
      // all loads go out before the reduction, only the partial sums cross LDS
      constexpr int PC = NVIT_RENORM_COLS_PER_ITEM;
      float* red = reinterpret_cast<float*>(smem);  // [RENORM_RG][PC]
      const int c0 = local * PC;
      const int cg = (tid % RENORM_CL) * 4, rg = tid / RENORM_CL;  // RENORM_CL threads cover a panel row; RENORM_RG row groups
      const bool vec = c0 + cg + 3 < cols;
      f32x4 v[RENORM_NR];
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < RENORM_NR; ++i) {
        const int r = rg + i * RENORM_RG;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
          if (vec)
            v[i] = *reinterpret_cast<const f32x4*>(W + (size_t)r * cols + c0 + cg);
          else
            for (int e = 0; e < 4; ++e)
              if (c0 + cg + e < cols) v[i][e] = W[(size_t)r * cols + c0 + cg + e];
        }
        acc += v[i] * v[i];
      }
      *reinterpret_cast<f32x4*>(red + rg * PC + cg) = acc;
      __syncthreads();
      // fixed-order two-level sum of the row-group partials: 8 parts per column, then 8
      float* red2 = red + RENORM_RG * PC;  // [8][PC] + [PC]
      if (tid < 8 * PC) {
        const int c = tid % PC, part = tid / PC;
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < RENORM_RG / 8; ++g) s += red[(part * (RENORM_RG / 8) + g) * PC + c];
        red2[part * PC + c] = s;
      }
      __syncthreads();
      if (tid < PC) {
        float s = 0.f;
#pragma unroll
        for (int part = 0; part < 8; ++part) s += red2[part * PC + tid];
        red2[8 * PC + tid] = sqrtf(s);
      }
      __syncthreads();
      const f32x4 nrm = *reinterpret_cast<const f32x4*>(red2 + 8 * PC + cg);
#pragma unroll
      for (int i = 0; i < RENORM_NR; ++i) {
        const int r = rg + i * RENORM_RG;
        if (r < rows) {
          const f32x4 o = v[i] / nrm;
          if (vec)
            *reinterpret_cast<f32x4*>(W + (size_t)r * cols + c0 + cg) = o;
          else
            for (int e = 0; e < 4; ++e)
              if (c0 + cg + e < cols) W[(size_t)r * cols + c0 + cg + e] = o[e];
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ int shadow_perm_row(int perm, int s, int F) {
  if (perm == 0) return s;
  const int q = s >> 5, w = s & 31;
  return w < 16 ? q * 16 + w : F + q * 16 + (w - 16);
}

// table row: {src, rows, cols, dst, dst_ld, dst_cols, dstT, dstT_ld, dstT_cols, perm, first_item, tiles_c}
template <typename T>
__global__ __launch_bounds__(256) void shadow_kernel(const int64_t* table, int n, int total_items) {
  __shared__ float tile[64][65];
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  for (int item = blockIdx.x; item < total_items; item += gridDim.x) {
    const int mi = table_row(table, n, 12, 10, item);
    const int64_t* e = table + mi * 12;
    const float* src = reinterpret_cast<const float*>(e[0]);
    const int rows = (int)e[1], cols = (int)e[2];
    T* dst = reinterpret_cast<T*>(e[3]);
    const int dst_ld = (int)e[4], dst_cols = (int)e[5];
    T* dstT = reinterpret_cast<T*>(e[6]);
    const int dstT_ld = (int)e[7], dstT_cols = (int)e[8];
    const int perm = (int)e[9];
    const int local = item - (int)e[10], tiles_c = (int)e[11];
    const int r0 = (local / tiles_c) * 64, c0 = (local % tiles_c) * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = r0 + ty + 4 * i, c = c0 + tx;
      float v = 0.f;
      if (perm == 2) {
        // split-precision image of a patch-embedding weight for patch_embed.hip: per 32 patch elements one 128-byte
        // slice [hi32 | lo32] with hi = bf16(w), lo = bf16(w - hi); dst_ld = 2 * Kp, the zero padding of the last
        // slice is written once by the host (the buffer is allocated zeroed)
        if constexpr (sizeof(T) == 2) {
          if (r < rows && c < cols) {
            v = src[(size_t)r * cols + c];
            const T hi = (T)v;
            const T lo = (T)(v - (float)hi);
            T* d = dst + (size_t)r * dst_ld + (c >> 5) * 64 + (c & 31);
            d[0] = hi;
            d[32] = lo;
          }
        }
        continue;
      }
      if (r < rows && c < cols) v = src[(size_t)shadow_perm_row(perm, r, rows / 2) * cols + c];
      tile[ty + 4 * i][tx] = v;
      if (dst && r < rows && c < dst_cols) dst[(size_t)r * dst_ld + c] = (T)v;
    }
    __syncthreads();
    if (dstT && perm != 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < cols && r < dstT_cols) dstT[(size_t)c * dstT_ld + r] = (T)tile[tx][ty + 4 * i];
      }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int nvit_renorm_weights(const int64_t* table, int n, int total_items, void* stream) {
  NVIT_REQUIRE(n > 0 && total_items > 0, "renorm: empty table");
  hipStream_t s = (hipStream_t)stream;
  // LDS for the largest column slab: caller guarantees rows <= 1152 for dim=0 matrices.
  static_assert(NVIT_RENORM_ROWS_PER_ITEM % (RENORM_THREADS / 64) == 0, "whole rounds of one row per wave");
  static const int kMaxLds = (RENORM_RG + 9) * NVIT_RENORM_COLS_PER_ITEM * 4;   // partial sums only (~18 KiB)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)renorm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) NVIT_FAIL((int)e, "renorm: cannot raise LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  int grid = total_items < 2048 ? total_items : 2048;
  ProfScope ps(NVIT_KID_RENORM, 0.0, 0.0, s);
  hipLaunchKernelGGL(renorm_kernel, dim3(grid), dim3(RENORM_THREADS), kMaxLds, s, table, n, total_items);
  NVIT_CHECK_LAUNCH("renorm");
  return NVIT_OK;
}

extern "C" int nvit_shadow_weights(const int64_t* table, int n, int total_items, int dt, void* stream) {
  NVIT_REQUIRE(n > 0 && total_items > 0, "shadow: empty table");
  NVIT_REQUIRE(dt == NVIT_F32 || dt == NVIT_BF16, "shadow: bad dt");
  hipStream_t s = (hipStream_t)stream;
  int grid = total_items < 4096 ? total_items : 4096;
  ProfScope ps(NVIT_KID_SHADOW, 0.0, 0.0, s);
  if (dt == NVIT_F32)
    hipLaunchKernelGGL(shadow_kernel<float>, dim3(grid), dim3(256), 0, s, table, n, total_items);
  else
    hipLaunchKernelGGL(shadow_kernel<bf16>, dim3(grid), dim3(256), 0, s, table, n, total_items);
  NVIT_CHECK_LAUNCH("shadow");
  return NVIT_OK;
}
