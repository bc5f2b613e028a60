// Optimizer step fused with the weight re-normalisation.
//
// Replaces, for every parameter of the model, the reference's per-step sequence
//   torch.nn.utils.clip_grad_norm_(params, grad_clip)          /root/reference/nvit/train.py:935-941
//   AdamW.step()  (groups of model.py:369-385)                 /root/reference/nvit/train.py:942-944
//   Trainer.normalize_matrices()                               /root/reference/nvit/train.py:461-480, 989-990
// by two launches over a device-side parameter table:
//   grad_sqnorm_kernel : per-workgroup partial sums of ||g||^2 over all gradients (fixed order, no atomics);
//   adamw_renorm_kernel: every workgroup re-reduces those partials (same order -> same clip factor everywhere),
//                        then walks its work items: AdamW on (p, g*clip, m, v) and, for the six matrices per block
//                        that normalize_matrices touches, the row / column L2 normalisation of the UPDATED weights
//                        before they are written - each of p, g, m, v is read once and p, m, v written once
//                        (28 B per parameter; the unfused sequence moves ~52 B and needs ~20 launches).
// Item kinds: -1 = plain chunk of 8192 elements; 1 = 16 rows, one wave per row (row norm, dim=1);
//             0 = slab of all rows x 32 columns kept in LDS (column norm, dim=0).
// AdamW arithmetic follows torch's (decoupled weight decay, bias corrections passed in from the host in double):
//   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).
#include "common.h"

namespace {

constexpr int OPT_COLS = 10;       // table columns (int64)
constexpr int OPT_CHUNK = 8192;    // elements per plain item and per grad-norm chunk
constexpr int OPT_ROWS_PER_ITEM = 16;  // one wave per row
constexpr int OPT_SLAB_COLS = 32;

struct OptRow {
  float* p;
  const float* g;
  float* m;
  float* v;
  int rows, cols, kind;
  int first_item, first_chunk;
  float lr, wd;
};

__device__ __forceinline__ OptRow opt_row(const int64_t* t) {
  OptRow r;
  r.p = reinterpret_cast<float*>(t[0]);
  r.g = reinterpret_cast<const float*>(t[1]);
  r.m = reinterpret_cast<float*>(t[2]);
  r.v = reinterpret_cast<float*>(t[3]);
  r.rows = (int)t[4];
  r.cols = (int)t[5];
  r.kind = (int)t[6];
  r.first_item = (int)t[7];
  r.first_chunk = (int)t[8];
  const unsigned long long h = (unsigned long long)t[9];
  r.lr = __uint_as_float((unsigned)(h & 0xffffffffull));
  r.wd = __uint_as_float((unsigned)(h >> 32));
  return r;
}

// binary search of the table row whose [first, next first) range holds `item` (column `col` holds the firsts)
__device__ __forceinline__ int opt_find(const int64_t* table, int n, int col, int item) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)table[mid * OPT_COLS + col] <= item) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <int NW>
__device__ __forceinline__ float block_sum(float s, float* red) {
  s = wave_sum(s);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = s;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) t += red[w];
  return t;
}

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const int64_t* table, int n, int total_chunks,
                                                          float* partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (int item = blockIdx.x; item < total_chunks; item += gridDim.x) {
    const int mi = opt_find(table, n, 8, item);
    const OptRow r = opt_row(table + mi * OPT_COLS);
    const long long numel = (long long)r.rows * r.cols;
    const long long e0 = (long long)(item - r.first_chunk) * OPT_CHUNK;
    const long long e1 = e0 + OPT_CHUNK < numel ? e0 + OPT_CHUNK : numel;
    if ((numel & 3) == 0) {
      for (long long e = e0 + threadIdx.x * 4; e < e1; e += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(r.g + e);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      }
    } else {
      for (long long e = e0 + threadIdx.x; e < e1; e += 256) s += r.g[e] * r.g[e];
    }
  }
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

struct AdamArgs {
  float b1, b2, eps, inv_bc1, inv_sqrt_bc2, max_norm;
  int npart;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a, float lr, float wd,
                                      float clip) {
  g *= clip;
  p *= 1.0f - lr * wd;
  m = a.b1 * m + (1.0f - a.b1) * g;
  v = a.b2 * v + (1.0f - a.b2) * g * g;
  const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
  p -= (lr * a.inv_bc1) * (m / denom);
}

__device__ __forceinline__ void adam4(f32x4& p, const f32x4& g, f32x4& m, f32x4& v, const AdamArgs& a, float lr,
                                      float wd, float clip) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float pe = p[e], me = m[e], ve = v[e];
    adam1(pe, g[e], me, ve, a, lr, wd, clip);
    p[e] = pe;
    m[e] = me;
    v[e] = ve;
  }
}

// 1024 threads = 16 waves, one workgroup per CU (the column slab takes most of the LDS), so the streaming items
// still have 16 waves x 4 tensors of loads in flight per CU.
__global__ __launch_bounds__(1024) void adamw_renorm_kernel(const int64_t* table, int n, int total_items,
                                                            const float* partial, AdamArgs a, float* gnorm_out,
                                                            const float* hyper) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (hyper) {  // bias corrections of the device-side step counter (nvit_adamw_tick): hipGraph replays stay correct
    a.inv_bc1 = hyper[1];
    a.inv_sqrt_bc2 = hyper[2];
  }
  // global gradient norm -> clip factor (every workgroup sums the same partials in the same order)
  float clip = 1.0f;
  if (partial) {
    float s = 0.f;
    for (int i = tid; i < a.npart; i += 1024) s += partial[i];
    s = block_sum<16>(s, red);
    const float nrm = sqrtf(s);
    if (a.max_norm > 0.f) {
      const float c = a.max_norm / (nrm + 1e-6f);
      clip = c < 1.0f ? c : 1.0f;
    }
    if (gnorm_out && blockIdx.x == 0 && tid == 0) gnorm_out[0] = nrm;
  }
  for (int item = blockIdx.x; item < total_items; item += gridDim.x) {
    const int mi = opt_find(table, n, 7, item);
    const OptRow r = opt_row(table + mi * OPT_COLS);
    const int local = item - r.first_item;
    if (r.kind < 0) {
      const long long numel = (long long)r.rows * r.cols;
      const long long e0 = (long long)local * OPT_CHUNK;
      const long long e1 = e0 + OPT_CHUNK < numel ? e0 + OPT_CHUNK : numel;
      if ((numel & 3) == 0) {
        for (long long e = e0 + tid * 4; e < e1; e += 4096) {
          f32x4 p = *reinterpret_cast<const f32x4*>(r.p + e);
          const f32x4 g = *reinterpret_cast<const f32x4*>(r.g + e);
          f32x4 m = *reinterpret_cast<const f32x4*>(r.m + e);
          f32x4 v = *reinterpret_cast<const f32x4*>(r.v + e);
          adam4(p, g, m, v, a, r.lr, r.wd, clip);
          *reinterpret_cast<f32x4*>(r.p + e) = p;
          *reinterpret_cast<f32x4*>(r.m + e) = m;
          *reinterpret_cast<f32x4*>(r.v + e) = v;
        }
      } else {
        for (long long e = e0 + tid; e < e1; e += 1024) {
          float p = r.p[e], m = r.m[e], v = r.v[e];
          adam1(p, r.g[e], m, v, a, r.lr, r.wd, clip);
          r.p[e] = p;
          r.m[e] = m;
          r.v[e] = v;
        }
      }
    } else if (r.kind == 1) {
      // rows: one wave per row; the updated row stays in registers (cols <= 1536, multiple of 4) for the norm
      const int row = local * OPT_ROWS_PER_ITEM + wid;
      if (row < r.rows) {
        const size_t off = (size_t)row * r.cols;
        const int kmax = (r.cols + 255) >> 8;
        f32x4 pw[6];
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          if (k < kmax) {
            const int c = (k * 64 + lane) * 4;
            const int cc = c < r.cols ? c : 0;  // clamped load, masked contribution, guarded stores
            f32x4 p = *reinterpret_cast<const f32x4*>(r.p + off + cc);
            const f32x4 g = *reinterpret_cast<const f32x4*>(r.g + off + cc);
            f32x4 m = *reinterpret_cast<const f32x4*>(r.m + off + cc);
            f32x4 v = *reinterpret_cast<const f32x4*>(r.v + off + cc);
            adam4(p, g, m, v, a, r.lr, r.wd, clip);
            if (c < r.cols) {
              *reinterpret_cast<f32x4*>(r.m + off + c) = m;
              *reinterpret_cast<f32x4*>(r.v + off + c) = v;
              ss += p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3];
            }
            pw[k] = p;
          }
        }
        ss = wave_sum(ss);
        const float nrm = sqrtf(ss);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int c = (k * 64 + lane) * 4;
          if (k < kmax && c < r.cols) *reinterpret_cast<f32x4*>(r.p + off + c) = pw[k] / nrm;
        }
      }
    } else {
      // column slab [rows][32] of updated weights in LDS; 8 threads cover 32 columns, 128 row groups
      float* slab = reinterpret_cast<float*>(smem);
      float* cred = slab + (size_t)r.rows * OPT_SLAB_COLS;  // [128][32]
      const int c0 = local * OPT_SLAB_COLS;
      const int cg = (tid & 7) * 4, rg = tid >> 3;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const bool full = c0 + cg + 3 < r.cols;
      for (int row = rg; row < r.rows; row += 128) {
        const size_t off = (size_t)row * r.cols + c0 + cg;
        f32x4 p = {0.f, 0.f, 0.f, 0.f};
        if (full) {
          p = *reinterpret_cast<const f32x4*>(r.p + off);
          const f32x4 g = *reinterpret_cast<const f32x4*>(r.g + off);
          f32x4 m = *reinterpret_cast<const f32x4*>(r.m + off);
          f32x4 v = *reinterpret_cast<const f32x4*>(r.v + off);
          adam4(p, g, m, v, a, r.lr, r.wd, clip);
          *reinterpret_cast<f32x4*>(r.m + off) = m;
          *reinterpret_cast<f32x4*>(r.v + off) = v;
        } else {
          for (int e = 0; e < 4; ++e)
            if (c0 + cg + e < r.cols) {
              float pe = r.p[off + e], me = r.m[off + e], ve = r.v[off + e];
              adam1(pe, r.g[off + e], me, ve, a, r.lr, r.wd, clip);
              r.m[off + e] = me;
              r.v[off + e] = ve;
              p[e] = pe;
            }
        }
        *reinterpret_cast<f32x4*>(slab + row * 32 + cg) = p;
        acc += p * p;
      }
      *reinterpret_cast<f32x4*>(cred + rg * 32 + cg) = acc;
      __syncthreads();
      if (tid < 32) {
        float s = 0.f;
        for (int gidx = 0; gidx < 128; ++gidx) s += cred[gidx * 32 + tid];
        cred[tid] = sqrtf(s);  // row 0 of cred is only read by thread `tid` above before this write
      }
      __syncthreads();
      const f32x4 nrm = *reinterpret_cast<const f32x4*>(cred + cg);
      for (int row = rg; row < r.rows; row += 128) {
        const size_t off = (size_t)row * r.cols + c0 + cg;
        const f32x4 p = *reinterpret_cast<const f32x4*>(slab + row * 32 + cg) / nrm;
        if (full)
          *reinterpret_cast<f32x4*>(r.p + off) = p;
        else
          for (int e = 0; e < 4; ++e)
            if (c0 + cg + e < r.cols) r.p[off + e] = p[e];
      }
      __syncthreads();
    }
  }
}

// hyper[0] = step count t (after the increment), hyper[1] = 1/(1-b1^t), hyper[2] = 1/sqrt(1-b2^t)
__global__ void adamw_tick_kernel(float* hyper, double b1, double b2) {
  const double t = (double)hyper[0] + 1.0;
  hyper[0] = (float)t;
  hyper[1] = (float)(1.0 / (1.0 - pow(b1, t)));
  hyper[2] = (float)(1.0 / sqrt(1.0 - pow(b2, t)));
}

}  // namespace

extern "C" int nvit_adamw_tick(float* hyper, double beta1, double beta2, void* stream) {
  NVIT_REQUIRE(hyper && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "adamw_tick: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, s, hyper, beta1, beta2);
  NVIT_CHECK_LAUNCH("adamw_tick");
  return NVIT_OK;
}

extern "C" int nvit_grad_sqnorm(const int64_t* table, int n, int total_chunks, float* partial, int npart,
                                void* stream) {
  NVIT_REQUIRE(table && partial && n > 0 && total_chunks > 0 && npart > 0 && npart <= 4096,
               "grad_sqnorm: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_OPTIM, 0.0, (double)total_chunks * OPT_CHUNK * 4.0, s);
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(npart), dim3(256), 0, s, table, n, total_chunks, partial);
  NVIT_CHECK_LAUNCH("grad_sqnorm");
  return NVIT_OK;
}

extern "C" int nvit_adamw_renorm(const int64_t* table, int n, int total_items, int max_slab_rows, float beta1,
                                 float beta2, float eps, double bias_correction1, double bias_correction2,
                                 const float* partial, int npart, float max_norm, float* gnorm_out,
                                 const float* hyper, void* stream) {
  NVIT_REQUIRE(table && n > 0 && total_items > 0, "adamw_renorm: empty table");
  NVIT_REQUIRE(max_slab_rows >= 0 && max_slab_rows <= 1152,
               "adamw_renorm: column-normalised matrix with %d rows exceeds the LDS slab (1152)", max_slab_rows);
  NVIT_REQUIRE(hyper || (bias_correction1 > 0.0 && bias_correction2 > 0.0),
               "adamw_renorm: bias corrections must be > 0");
  NVIT_REQUIRE(!partial || (npart > 0 && npart <= 4096), "adamw_renorm: bad npart");
  AdamArgs a;
  a.b1 = beta1;
  a.b2 = beta2;
  a.eps = eps;
  a.inv_bc1 = hyper ? 0.f : (float)(1.0 / bias_correction1);
  a.inv_sqrt_bc2 = hyper ? 0.f : (float)(1.0 / sqrt(bias_correction2));
  a.max_norm = max_norm;
  a.npart = npart;
  hipStream_t s = (hipStream_t)stream;
  const int lds = max_slab_rows > 0 ? (max_slab_rows * OPT_SLAB_COLS + 128 * 32) * 4 : 0;
  static int lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)adamw_renorm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) NVIT_FAIL((int)e, "adamw_renorm: cannot raise LDS limit: %s", hipGetErrorString(e));
    lds_set = lds;
  }
  int grid = total_items < 2048 ? total_items : 2048;
  ProfScope ps(NVIT_KID_OPTIM, 0.0, 0.0, s);
  hipLaunchKernelGGL(adamw_renorm_kernel, dim3(grid), dim3(1024), lds, s, table, n, total_items, partial, a, gnorm_out, hyper);
  NVIT_CHECK_LAUNCH("adamw_renorm");
  return NVIT_OK;
}
