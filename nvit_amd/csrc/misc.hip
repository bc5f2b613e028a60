// Patch extraction (im2col with reflect padding), token mean-pool + LayerNorm head, and the
// reconstruction loss.  All HBM-bound helpers around the GEMMs.
#include "common.h"

namespace {

__device__ __forceinline__ int reflect(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

template <typename T>
struct ColWriter {
  static constexpr int MUL = 1;
  static __device__ __forceinline__ void put(T* row, int K, int k, f32x4 v) { store4<T>(row + k, v); }
};

// one thread -> 4 consecutive pw of one (token, c, ph); column order (c, ph, pw)
template <typename W, typename T>
__global__ void im2col_kernel(const float* img, T* A_l, T* A_g, int B, int ch, int S, int Pl, int Pg) {
  const int G = S / Pl, Tn = G * G;
  const int Kl = ch * Pl * Pl, Kg = ch * Pg * Pg, pad = (Pg - Pl) / 2;
  const long long nl = (long long)B * Tn * (Kl / 4), ng = (long long)B * Tn * (Kg / 4);
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < nl + ng;
       idx += (long long)gridDim.x * blockDim.x) {
    if (idx < nl) {
      const int k = (int)(idx % (Kl / 4)) * 4;
      const long long m = idx / (Kl / 4);
      const int b = (int)(m / Tn), t = (int)(m % Tn), ty = t / G, tx = t % G;
      const int c = k / (Pl * Pl), ph = (k / Pl) % Pl, pw = k % Pl;
      const float* src = img + (((size_t)b * ch + c) * S + ty * Pl + ph) * S + tx * Pl + pw;
      ColWriter<W>::put(A_l + (size_t)m * Kl * ColWriter<W>::MUL, Kl, k, *reinterpret_cast<const f32x4*>(src));
    } else {
      const long long j = idx - nl;
      const int k = (int)(j % (Kg / 4)) * 4;
      const long long m = j / (Kg / 4);
      const int b = (int)(m / Tn), t = (int)(m % Tn), ty = t / G, tx = t % G;
      const int c = k / (Pg * Pg), ph = (k / Pg) % Pg, pw = k % Pg;
      const int y = reflect(ty * Pl - pad + ph, S);
      const float* row = img + (((size_t)b * ch + c) * S + y) * S;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = row[reflect(tx * Pl - pad + pw + e, S)];
      ColWriter<W>::put(A_g + (size_t)m * Kg * ColWriter<W>::MUL, Kg, k, v);
    }
  }
}

// partial token sums: grid (nchunk, B); ws[b][chunk][c]
__global__ __launch_bounds__(256) void pool_partial_kernel(const float* x, float* ws, int nchunk, int T, int C) {
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int per = (T + nchunk - 1) / nchunk;
  const int t0 = chunk * per, t1 = min(T, t0 + per);
  for (int c = threadIdx.x * 4; c < C; c += 1024) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int t = t0; t < t1; ++t) s += *reinterpret_cast<const f32x4*>(x + ((size_t)b * T + t) * C + c);
    *reinterpret_cast<f32x4*>(ws + ((size_t)b * nchunk + chunk) * C + c) = s;
  }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

template <typename T>
__global__ __launch_bounds__(256) void pool_ln_kernel(const float* ws, int nchunk, int Tn, int C, const float* w,
                                                       const float* bias, float eps, float* pooled, float* ln,
                                                       T* ln_lo, float* stats) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + 256 * i;
    float a = 0.f;
    if (c < C) {
      for (int k = 0; k < nchunk; ++k) a += ws[((size_t)b * nchunk + k) * C + c];
      a = a / (float)Tn;
      pooled[(size_t)b * C + c] = a;
    }
    v[i] = a;
    s += a;
  }
  const float mean = block_sum(s, red) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < C) q += (v[i] - mean) * (v[i] - mean);
  }
  const float var = block_sum(q, red) / (float)C;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < C) {
      const float o = (v[i] - mean) * rstd * w[c] + bias[c];
      ln[(size_t)b * C + c] = o;
      if (ln_lo) ln_lo[(size_t)b * C + c] = (T)o;
    }
  }
  if (threadIdx.x == 0) {
    stats[b * 2] = mean;
    stats[b * 2 + 1] = rstd;
  }
}

// grid (nchunk, B): LayerNorm backward for row b, then dx[b, t, :] = dpooled / T for the chunk's tokens
__global__ __launch_bounds__(256) void pool_ln_bwd_kernel(const float* dln, const float* pooled, const float* w,
                                                           const float* stats, float* dx, int nchunk, int Tn, int C) {
  __shared__ float red[4];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const float mean = stats[b * 2], rstd = stats[b * 2 + 1];
  float g[8], xh[8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + 256 * i;
    g[i] = 0.f;
    xh[i] = 0.f;
    if (c < C) {
      g[i] = dln[(size_t)b * C + c] * w[c];
      xh[i] = (pooled[(size_t)b * C + c] - mean) * rstd;
    }
    s1 += g[i];
    s2 += g[i] * xh[i];
  }
  const float m1 = block_sum(s1, red) / (float)C;
  const float m2 = block_sum(s2, red) / (float)C;
  const float invT = 1.0f / (float)Tn;
  const int per = (Tn + nchunk - 1) / nchunk;
  const int t0 = chunk * per, t1 = min(Tn, t0 + per);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < C) {
      const float d = rstd * (g[i] - m1 - xh[i] * m2) * invT;
      for (int t = t0; t < t1; ++t) dx[((size_t)b * Tn + t) * C + c] = d;
    }
  }
}

__global__ void ln_param_grad_kernel(const float* dln, const float* pooled, const float* stats, float* dw, float* db,
                                     int accumulate, int B, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float sw = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) {
    const float d = dln[(size_t)b * C + c];
    sw += d * (pooled[(size_t)b * C + c] - stats[b * 2]) * stats[b * 2 + 1];
    sb += d;
  }
  dw[c] = accumulate ? dw[c] + sw : sw;
  db[c] = accumulate ? db[c] + sb : sb;
}

__global__ __launch_bounds__(256) void recon_partial_kernel(const float* raw, const float* img, float* part, int B,
                                                             int ch, int S, int P) {
  __shared__ float red[4];
  const int G = S / P, Tn = G * G, K = ch * P * P;
  const long long total = (long long)B * Tn * K;
  float s = 0.f;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % K);
    const long long m = idx / K;
    const int b = (int)(m / Tn), t = (int)(m % Tn), ty = t / G, tx = t % G;
    const int c = k / (P * P), ph = (k / P) % P, pw = k % P;
    const float tgt = img[(((size_t)b * ch + c) * S + ty * P + ph) * S + tx * P + pw];
    const float d = tanhf(raw[idx]) - tgt;
    s += d * d;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void recon_final_kernel(const float* part, int nblk, float inv_n, float* loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += part[i];
    loss[0] = s * inv_n;
  }
}

}  // namespace

extern "C" int nvit_im2col(int dt, const float* img, void* A_l, void* A_g, int B, int ch, int S, int Pl, int Pg,
                           void* stream) {
  NVIT_REQUIRE(Pl % 4 == 0 && Pg % 4 == 0 && S % Pl == 0 && Pg >= Pl && (Pg - Pl) % 2 == 0,
               "im2col: unsupported patch geometry S=%d Pl=%d Pg=%d", S, Pl, Pg);
  NVIT_REQUIRE((Pg - Pl) / 2 < S, "im2col: reflect pad must be smaller than the image");
  hipStream_t s = (hipStream_t)stream;
  const int G = S / Pl;
  const long long n = (long long)B * G * G * ((ch * Pl * Pl + ch * Pg * Pg) / 4);
  int blocks = cdiv(n, 256);
  if (blocks > 8192) blocks = 8192;
  NVIT_REQUIRE(dt == NVIT_F32 || dt == NVIT_BF16, "im2col: bad dt %d", dt);
  const double es = dt == NVIT_F32 ? 4.0 : 2.0;
  ProfScope ps(NVIT_KID_PATCHIFY, 0.0, (double)B * ch * S * S * 4.0 + (double)n * 4.0 * es, s);
  if (dt == NVIT_F32)
    hipLaunchKernelGGL((im2col_kernel<float, float>), dim3(blocks), dim3(256), 0, s, img, (float*)A_l, (float*)A_g, B, ch, S, Pl, Pg);
  else
    hipLaunchKernelGGL((im2col_kernel<bf16, bf16>), dim3(blocks), dim3(256), 0, s, img, (bf16*)A_l, (bf16*)A_g, B, ch, S, Pl, Pg);
  NVIT_CHECK_LAUNCH("im2col");
  return NVIT_OK;
}

extern "C" int nvit_pool_ln_fwd(int dt, const float* x, const float* w, const float* b, float eps, float* pooled,
                                float* ln, void* ln_lo, float* stats, float* ws, int nchunk, int B, int T, int C,
                                void* stream) {
  NVIT_REQUIRE(C % 4 == 0 && C <= 2048 && nchunk > 0, "pool_ln_fwd: bad C=%d", C);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_MISC, 0.0, (double)B * T * C * 4.0, s);
  hipLaunchKernelGGL(pool_partial_kernel, dim3(nchunk, B), dim3(256), 0, s, x, ws, nchunk, T, C);
  NVIT_CHECK_LAUNCH("pool_partial");
  if (dt == NVIT_F32)
    hipLaunchKernelGGL(pool_ln_kernel<float>, dim3(B), dim3(256), 0, s, ws, nchunk, T, C, w, b, eps, pooled, ln, (float*)ln_lo, stats);
  else
    hipLaunchKernelGGL(pool_ln_kernel<bf16>, dim3(B), dim3(256), 0, s, ws, nchunk, T, C, w, b, eps, pooled, ln, (bf16*)ln_lo, stats);
  NVIT_CHECK_LAUNCH("pool_ln");
  return NVIT_OK;
}

extern "C" int nvit_pool_ln_bwd(const float* dln, const float* pooled, const float* w, const float* stats, float* dx,
                                float* dw, float* db, int accumulate, int B, int T, int C, void* stream) {
  NVIT_REQUIRE(C <= 2048, "pool_ln_bwd: bad C=%d", C);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_MISC, 0.0, (double)B * T * C * 4.0, s);
  int nchunk = cdiv(2048, B);
  if (nchunk > T) nchunk = T;
  hipLaunchKernelGGL(pool_ln_bwd_kernel, dim3(nchunk, B), dim3(256), 0, s, dln, pooled, w, stats, dx, nchunk, T, C);
  NVIT_CHECK_LAUNCH("pool_ln_bwd");
  hipLaunchKernelGGL(ln_param_grad_kernel, dim3(cdiv(C, 128)), dim3(128), 0, s, dln, pooled, stats, dw, db, accumulate, B, C);
  NVIT_CHECK_LAUNCH("ln_param_grad");
  return NVIT_OK;
}

extern "C" int nvit_recon_loss(const float* raw, const float* img, float* part, int nblk, float* loss, int B, int ch,
                               int S, int P, void* stream) {
  NVIT_REQUIRE(nblk > 0 && nblk <= 4096 && S % P == 0, "recon_loss: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_MISC, 0.0, (double)B * ch * S * S * 8.0, s);
  hipLaunchKernelGGL(recon_partial_kernel, dim3(nblk), dim3(256), 0, s, raw, img, part, B, ch, S, P);
  NVIT_CHECK_LAUNCH("recon_partial");
  const double n = (double)B * ch * S * S;
  hipLaunchKernelGGL(recon_final_kernel, dim3(1), dim3(64), 0, s, part, nblk, (float)(1.0 / n), loss);
  NVIT_CHECK_LAUNCH("recon_final");
  return NVIT_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Cross-entropy loss, forward and gradient in one pass (reference train.py:906
// `F.cross_entropy(logits, Y)`: mean over the batch of logsumexp(row) - row[label]).  One workgroup per sample writes
// rowloss[b] and dlogits[b,:] = (softmax(row) - onehot(label)) / B; a single-workgroup launch sums the row losses in
// a fixed order (deterministic).
namespace {

__global__ __launch_bounds__(256) void ce_rows_kernel(const float* logits, const int64_t* labels, float* rowloss,
                                                      float* dlogits, int B, int N) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* row = logits + (size_t)b * N;
  float mx = -INFINITY;
  for (int n = tid; n < N; n += 256) mx = fmaxf(mx, row[n]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int n = tid; n < N; n += 256) s += expf(row[n] - mx);
  s = wave_sum(s);
  if (lane == 0) red[wid] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  const int lab = (int)labels[b];
  const float lse = mx + logf(s);
  if (tid == 0) rowloss[b] = (lab >= 0 && lab < N) ? lse - row[lab] : 0.f;
  const float inv = 1.0f / s, invB = 1.0f / (float)B;
  float* drow = dlogits + (size_t)b * N;
  for (int n = tid; n < N; n += 256) drow[n] = (expf(row[n] - mx) * inv - (n == lab ? 1.0f : 0.0f)) * invB;
}

__global__ __launch_bounds__(256) void ce_mean_kernel(const float* rowloss, float* loss, int B) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float s = 0.f;
  for (int b = tid; b < B; b += 256) s += rowloss[b];
  s = wave_sum(s);
  if (lane == 0) red[wid] = s;
  __syncthreads();
  if (tid == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) / (float)B;
}

}  // namespace

extern "C" int nvit_ce_loss(const float* logits, const int64_t* labels, float* rowloss, float* loss, float* dlogits,
                            int B, int N, void* stream) {
  NVIT_REQUIRE(logits && labels && rowloss && loss && dlogits && B > 0 && N > 0, "ce_loss: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(B), dim3(256), 0, s, logits, labels, rowloss, dlogits, B, N);
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, s, rowloss, loss, B);
  NVIT_CHECK_LAUNCH("ce_loss");
  return NVIT_OK;
}
