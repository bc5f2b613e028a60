// impl 0: scalar-FMA attention kernels (any element type, fp32 math).  They carry the exact-f32
// parity mode and serve as the on-device cross-check of the MFMA flash kernels (attn_mfma.hip).
// One thread owns one query (fwd, dq) or one key (dk/dv); the other side is streamed through
// LDS in tiles of 32 rows and read by broadcast.
#include "common.h"

namespace {

constexpr int TILE = 32;

template <typename T, int D>
__global__ __launch_bounds__(64) void attn_fwd_ref(const T* qh, const T* kh, const T* vh, float scale, T* o,
                                                    float* lse, int H, int Tq, int Tk) {
  __shared__ float ks[TILE][D], vs[TILE][D];
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int qi = blockIdx.x * 64 + threadIdx.x;
  const bool ok = qi < Tq;
  const T* qp = qh + ((size_t)bh * Tq + (ok ? qi : 0)) * D;
  float q[D], acc[D];
#pragma unroll
  for (int e = 0; e < D; ++e) {
    q[e] = (float)qp[e] * scale;
    acc[e] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < Tk; k0 += TILE) {
    const int nk = min(TILE, Tk - k0);
    for (int i = threadIdx.x; i < TILE * D; i += 64) {
      const int r = i / D, e = i % D;
      const bool in = r < nk;
      ks[r][e] = in ? (float)kh[((size_t)bh * Tk + k0 + r) * D + e] : 0.f;
      vs[r][e] = in ? (float)vh[((size_t)bh * Tk + k0 + r) * D + e] : 0.f;
    }
    __syncthreads();
    for (int j = 0; j < nk; ++j) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < D; ++e) s += q[e] * ks[j][e];
      const float mn = fmaxf(m, s);
      const float corr = expf(m - mn), p = expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int e = 0; e < D; ++e) acc[e] = acc[e] * corr + p * vs[j][e];
      m = mn;
    }
    __syncthreads();
  }
  if (ok) {
    const float inv = 1.0f / l;
    T* op = o + ((size_t)b * Tq + qi) * (H * D) + h * D;
#pragma unroll
    for (int e = 0; e < D; ++e) op[e] = (T)(acc[e] * inv);
    lse[(size_t)bh * Tq + qi] = m + logf(l);
  }
}

// delta[b,h,t] = sum_e dO[b,t,h*D+e] * O[b,t,h*D+e].  One thread per 8 consecutive channels of a
// token row (coalesced 16/32-byte loads); the D/8 lanes of a head are adjacent and reduce by shuffles.
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* dout, const T* o, float* delta, int B, int H,
                                                          int Tq, int D) {
  const int C8 = (H * D) >> 3, G = D >> 3;  // chunks per row, lanes per head (4 or 8)
  const long long total = (long long)B * Tq * C8;
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  float s = 0.f;
  long long m = 0;
  int c8 = 0;
  if (idx < total) {
    m = idx / C8;
    c8 = (int)(idx % C8);
    const size_t off = (size_t)m * (H * D) + (size_t)c8 * 8;
    const f32x4 a0 = load4<T>(dout + off), a1 = load4<T>(dout + off + 4);
    const f32x4 b0 = load4<T>(o + off), b1 = load4<T>(o + off + 4);
    s = a0[0] * b0[0] + a0[1] * b0[1] + a0[2] * b0[2] + a0[3] * b0[3] + a1[0] * b1[0] + a1[1] * b1[1] +
        a1[2] * b1[2] + a1[3] * b1[3];
  }
  s = group_sum_dyn(s, G);
  if (idx < total && (c8 % G) == 0) {
    const int h = c8 / G;
    const int b = (int)(m / Tq), t = (int)(m % Tq);
    delta[((size_t)b * H + h) * Tq + t] = s;
  }
}

template <typename T, int D>
__global__ __launch_bounds__(64) void attn_bwd_dq_ref(const T* dout, const T* qh, const T* kh, const T* vh,
                                                       const float* lse, const float* delta, float scale, T* dqh,
                                                       int H, int Tq, int Tk) {
  __shared__ float ks[TILE][D], vs[TILE][D];
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int qi = blockIdx.x * 64 + threadIdx.x;
  const bool ok = qi < Tq;
  const int qc = ok ? qi : 0;
  const T* qp = qh + ((size_t)bh * Tq + qc) * D;
  const T* gp = dout + ((size_t)b * Tq + qc) * (H * D) + h * D;
  float q[D], g[D], acc[D];
#pragma unroll
  for (int e = 0; e < D; ++e) {
    q[e] = (float)qp[e] * scale;
    g[e] = (float)gp[e];
    acc[e] = 0.f;
  }
  const float L = lse[(size_t)bh * Tq + qc], dl = delta[(size_t)bh * Tq + qc];
  for (int k0 = 0; k0 < Tk; k0 += TILE) {
    const int nk = min(TILE, Tk - k0);
    for (int i = threadIdx.x; i < TILE * D; i += 64) {
      const int r = i / D, e = i % D;
      const bool in = r < nk;
      ks[r][e] = in ? (float)kh[((size_t)bh * Tk + k0 + r) * D + e] : 0.f;
      vs[r][e] = in ? (float)vh[((size_t)bh * Tk + k0 + r) * D + e] : 0.f;
    }
    __syncthreads();
    for (int j = 0; j < nk; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int e = 0; e < D; ++e) {
        s += q[e] * ks[j][e];
        dp += g[e] * vs[j][e];
      }
      const float p = expf(s - L);
      const float ds = p * (dp - dl) * scale;
#pragma unroll
      for (int e = 0; e < D; ++e) acc[e] += ds * ks[j][e];
    }
    __syncthreads();
  }
  if (ok) {
    T* op = dqh + ((size_t)bh * Tq + qi) * D;
#pragma unroll
    for (int e = 0; e < D; ++e) op[e] = (T)acc[e];
  }
}

template <typename T, int D>
__global__ __launch_bounds__(64) void attn_bwd_dkv_ref(const T* dout, const T* qh, const T* kh, const T* vh,
                                                        const float* lse, const float* delta, float scale, T* dkh,
                                                        T* dvh, int H, int Tq, int Tk) {
  __shared__ float qs[TILE][D], gs[TILE][D];
  __shared__ float ls[TILE], ds_[TILE];
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int ki = blockIdx.x * 64 + threadIdx.x;
  const bool ok = ki < Tk;
  const int kc = ok ? ki : 0;
  float k[D], v[D], dk[D], dv[D];
#pragma unroll
  for (int e = 0; e < D; ++e) {
    k[e] = (float)kh[((size_t)bh * Tk + kc) * D + e];
    v[e] = (float)vh[((size_t)bh * Tk + kc) * D + e];
    dk[e] = 0.f;
    dv[e] = 0.f;
  }
  for (int q0 = 0; q0 < Tq; q0 += TILE) {
    const int nq = min(TILE, Tq - q0);
    for (int i = threadIdx.x; i < TILE * D; i += 64) {
      const int r = i / D, e = i % D;
      const bool in = r < nq;
      qs[r][e] = in ? (float)qh[((size_t)bh * Tq + q0 + r) * D + e] : 0.f;
      gs[r][e] = in ? (float)dout[((size_t)b * Tq + q0 + r) * (H * D) + h * D + e] : 0.f;
    }
    if (threadIdx.x < TILE) {
      const bool in = threadIdx.x < nq;
      ls[threadIdx.x] = in ? lse[(size_t)bh * Tq + q0 + threadIdx.x] : 0.f;
      ds_[threadIdx.x] = in ? delta[(size_t)bh * Tq + q0 + threadIdx.x] : 0.f;
    }
    __syncthreads();
    for (int j = 0; j < nq; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int e = 0; e < D; ++e) {
        s += qs[j][e] * k[e];
        dp += gs[j][e] * v[e];
      }
      const float p = expf(s * scale - ls[j]);
      const float dsv = p * (dp - ds_[j]) * scale;
#pragma unroll
      for (int e = 0; e < D; ++e) {
        dv[e] += p * gs[j][e];
        dk[e] += dsv * qs[j][e];
      }
    }
    __syncthreads();
  }
  if (ok) {
#pragma unroll
    for (int e = 0; e < D; ++e) {
      dkh[((size_t)bh * Tk + ki) * D + e] = (T)dk[e];
      dvh[((size_t)bh * Tk + ki) * D + e] = (T)dv[e];
    }
  }
}

}  // namespace

int nvit_attn_fwd_mfma(const void* qh, const void* kh, const void* vh, float scale, float qpre, const float* sqk,
                       float c_q, void* o, float* lse, int B, int H, int Tq, int Tk, int d, hipStream_t s);
int nvit_attn_bwd_mfma(const void* dout, const void* qh, const void* kh, const void* vh, const void* o, const float* lse,
                       float* delta, float scale, void* dqh, void* dkh, void* dvh, int B, int H, int Tq, int Tk,
                       int d, hipStream_t s);

int nvit_attn_bwd_mfma_fused(const void* dout, const void* qh, const void* kh, const void* vh, const void* o,
                             const float* lse, float* delta, float scale, const float* rq, const float* rk, const float* sqk,
                             float c_q, float qpre, void* dq, int ldq, void* dk, void* dv, int ldkv, float* part_q,
                             float* part_k, int B, int H, int Tq, int Tk, int d, hipStream_t s);

static int attn_fwd_impl(int dt, int impl, const void* qh, const void* kh, const void* vh, float scale, const float* sqk,
                         float c_q, float qpre, void* o, float* lse, int B, int H, int Tq, int Tk, int d, void* stream);

extern "C" int nvit_attn_fwd(int dt, int impl, const void* qh, const void* kh, const void* vh, float scale, void* o,
                             float* lse, int B, int H, int Tq, int Tk, int d, void* stream) {
  return attn_fwd_impl(dt, impl, qh, kh, vh, scale, nullptr, 0.f, 1.0f, o, lse, B, H, Tq, Tk, d, stream);
}

// nViT call sites: q and k are (sqk*c_q) * unit vectors per head, which bounds every score; the MFMA kernel then skips
// the running maximum (see attn_mfma.hip).  Same result as nvit_attn_fwd up to rounding.  q_prescale: qh holds
// q_prescale * q_hat (the producer folded the factor into the learned scale); 1 = plain.  With q_prescale =
// scale * log2(e) the MFMA kernel's exponent needs no multiply (the fused training path).
extern "C" int nvit_attn_fwd_bounded(int dt, int impl, const void* qh, const void* kh, const void* vh, float scale,
                                     const float* sqk, float c_q, float q_prescale, void* o, float* lse, int B, int H,
                                     int Tq, int Tk, int d, void* stream) {
  NVIT_REQUIRE(sqk != nullptr, "attn_fwd_bounded: sqk is NULL (use nvit_attn_fwd)");
  NVIT_REQUIRE(q_prescale > 0.f, "attn_fwd_bounded: q_prescale must be positive");
  return attn_fwd_impl(dt, impl, qh, kh, vh, scale, sqk, c_q, q_prescale, o, lse, B, H, Tq, Tk, d, stream);
}

static int attn_fwd_impl(int dt, int impl, const void* qh, const void* kh, const void* vh, float scale, const float* sqk,
                         float c_q, float qpre, void* o, float* lse, int B, int H, int Tq, int Tk, int d, void* stream) {
  NVIT_REQUIRE(d == 32 || d == 64, "attn_fwd: head dim %d unsupported (32 or 64)", d);
  NVIT_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "attn_fwd: empty problem");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_ATTN_FWD, 4.0 * B * H * (double)Tq * Tk * d, 0.0, s);
  if (impl == 1) {
    NVIT_REQUIRE(dt == NVIT_BF16, "attn_fwd: MFMA kernel needs bf16");
    return nvit_attn_fwd_mfma(qh, kh, vh, scale, qpre, sqk, c_q, o, lse, B, H, Tq, Tk, d, s);
  }
  scale = scale / qpre;   // the scalar kernels take the multiplier of q.k directly
  dim3 grid(cdiv(Tq, 64), B * H);
#define L(T, D) \
  hipLaunchKernelGGL((attn_fwd_ref<T, D>), grid, dim3(64), 0, s, (const T*)qh, (const T*)kh, (const T*)vh, scale, (T*)o, lse, H, Tq, Tk)
  if (dt == NVIT_F32) { if (d == 32) L(float, 32); else L(float, 64); }
  else { if (d == 32) L(bf16, 32); else L(bf16, 64); }
#undef L
  NVIT_CHECK_LAUNCH("attn_fwd_ref");
  return NVIT_OK;
}

extern "C" int nvit_attn_bwd(int dt, int impl, const void* dout, const void* qh, const void* kh, const void* vh,
                             const void* o, const float* lse, float scale, void* dqh, void* dkh, void* dvh,
                             float* delta, int B, int H, int Tq, int Tk, int d, void* stream) {
  NVIT_REQUIRE(d == 32 || d == 64, "attn_bwd: head dim %d unsupported (32 or 64)", d);
  NVIT_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "attn_bwd: empty problem");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_ATTN_BWD, 10.0 * B * H * (double)Tq * Tk * d, 0.0, s);
  if (impl == 1) {
    // the MFMA dq kernel computes delta = rowsum(dO * O) itself and leaves it in `delta` for the dk/dv kernel
    NVIT_REQUIRE(dt == NVIT_BF16, "attn_bwd: MFMA kernel needs bf16");
    return nvit_attn_bwd_mfma(dout, qh, kh, vh, o, lse, delta, scale, dqh, dkh, dvh, B, H, Tq, Tk, d, s);
  }
  const long long total = (long long)B * Tq * ((H * d) / 8);
  if (dt == NVIT_F32)
    hipLaunchKernelGGL(attn_delta_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float*)dout,
                       (const float*)o, delta, B, H, Tq, d);
  else
    hipLaunchKernelGGL(attn_delta_kernel<bf16>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const bf16*)dout,
                       (const bf16*)o, delta, B, H, Tq, d);
  NVIT_CHECK_LAUNCH("attn_delta");
  dim3 gq(cdiv(Tq, 64), B * H), gk(cdiv(Tk, 64), B * H);
#define L(T, D)                                                                                                       \
  do {                                                                                                                \
    hipLaunchKernelGGL((attn_bwd_dq_ref<T, D>), gq, dim3(64), 0, s, (const T*)dout, (const T*)qh, (const T*)kh,       \
                       (const T*)vh, lse, delta, scale, (T*)dqh, H, Tq, Tk);                                          \
    hipLaunchKernelGGL((attn_bwd_dkv_ref<T, D>), gk, dim3(64), 0, s, (const T*)dout, (const T*)qh, (const T*)kh,      \
                       (const T*)vh, lse, delta, scale, (T*)dkh, (T*)dvh, H, Tq, Tk);                                 \
  } while (0)
  if (dt == NVIT_F32) { if (d == 32) L(float, 32); else L(float, 64); }
  else { if (d == 32) L(bf16, 32); else L(bf16, 64); }
#undef L
  NVIT_CHECK_LAUNCH("attn_bwd_ref");
  return NVIT_OK;
}

// MFMA attention backward (bf16, d = 64) with the q/k-normalise backward fused into the epilogues.
extern "C" int nvit_attn_bwd_qknorm(int dt, const void* dout, const void* qh, const void* kh, const void* vh,
                                    const void* o, const float* lse, float scale, const float* rq, const float* rk,
                                    const float* sqk, float c_q, float q_prescale, void* dq, int ldq, void* dk, void* dv,
                                    int ldkv, float* part_q, float* part_k, float* delta, int B, int H, int Tq, int Tk,
                                    int d, void* stream) {
  NVIT_REQUIRE(dt == NVIT_BF16 && d == 64, "attn_bwd_qknorm: needs bf16 and head dim 64");
  NVIT_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "attn_bwd_qknorm: empty problem");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_ATTN_BWD, 10.0 * B * H * (double)Tq * Tk * d, 0.0, s);
  return nvit_attn_bwd_mfma_fused(dout, qh, kh, vh, o, lse, delta, scale, rq, rk, sqk, c_q, q_prescale, dq, ldq, dk, dv,
                                  ldkv, part_q, part_k, B, H, Tq, Tk, d, s);
}
