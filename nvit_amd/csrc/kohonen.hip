// Kohonen (SOM) head kernels — BASELINE config C5.  Reference: /root/reference/nvit/kohonen.py:100-165 and
// nvit/model.py:419-444, 482-561.
//   som_bmu       best-matching unit = argmin_n ||x - node_n||  from the score GEMM x.node^T (exact-f32 MFMA)
//   gather/scatter rows   repr = nodes[idx] and its (deterministic) backward
//   som_update    the reference's sequential per-sample neighbourhood update, collapsed into two launches:
//                 the recurrence node <- node + s_i (v_i - node) is independent per (node, channel), so one thread
//                 carries it through the B samples in registers (reference: ~242 tiny torch kernels per sample)
//   consistency / huber / smoothness losses and their gradients, reconstruction-loss gradient
#include "common.h"

namespace {

__global__ void rownorm2_kernel(const float* a, int N, int C, float* out) {
  const int n = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 64) s += a[(size_t)n * C + c] * a[(size_t)n * C + c];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[n] = s;
}

// idx[m] = argmin_n (nn[n] - 2 * S[m][n]); first minimum wins (torch.argmin semantics on ties)
__global__ __launch_bounds__(256) void bmu_kernel(const float* S, const float* nn, int M, int N, long long* idx) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int m = blockIdx.x * 4 + wid; m < M; m += gridDim.x * 4) {
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int n = lane; n < N; n += 64) {
      const float d = nn[n] - 2.0f * S[(size_t)m * N + n];
      if (d < best) {
        best = d;
        bi = n;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob < best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    if (lane == 0) idx[m] = bi;
  }
}

__global__ void gather_rows_kernel(const float* nodes, const long long* idx, float* out, long long M, int C) {
  const int c4 = C >> 2;
  const long long total = M * c4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / c4;
    const int c = (int)(i % c4) * 4;
    *reinterpret_cast<f32x4*>(out + m * C + c) = *reinterpret_cast<const f32x4*>(nodes + idx[m] * C + c);
  }
}

// dnodes[n][c] = sum over rows m with idx[m] == n of dout[m][c], rows visited in increasing m (deterministic).
// One workgroup per node: the index array is scanned 256 entries at a time (one compare per thread, wave ballots
// through LDS), then every thread walks the set bits in order and accumulates its columns (C <= 2048).
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* dout, const long long* idx, float* dnodes,
                                                           long long M, int C) {
  __shared__ unsigned long long masks[4];
  const int n = blockIdx.x, tid = threadIdx.x, wid = tid >> 6;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  for (long long m0 = 0; m0 < M; m0 += 256) {
    const long long m = m0 + tid;
    const bool hit = m < M && idx[m] == n;
    const unsigned long long bal = __ballot(hit);
    if ((tid & 63) == 0) masks[wid] = bal;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned long long mm = masks[w];
      while (mm) {
        const int b = __builtin_ctzll(mm);
        mm &= mm - 1;
        const float* row = dout + (m0 + w * 64 + b) * C;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = tid + 256 * j;
          if (c < C) s[j] += row[c];
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = tid + 256 * j;
    if (c < C) dnodes[(size_t)n * C + c] = s[j];
  }
}

// one-hot rows: out[m][n] = (idx[m] == n); the node gradient is then onehot^T . dout, a perfectly load-balanced
// weight-gradient GEMM (exact in fp32: products are 1*x), whatever the BMU histogram looks like
__global__ void onehot_kernel(const long long* idx, float* out, long long M, int N) {
  const int n4 = N >> 2;
  const long long total = M * n4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / n4;
    const int n = (int)(i % n4) * 4;
    const int w = (int)idx[m];
    f32x4 v = {w == n ? 1.f : 0.f, w == n + 1 ? 1.f : 0.f, w == n + 2 ? 1.f : 0.f, w == n + 3 ? 1.f : 0.f};
    *reinterpret_cast<f32x4*>(out + m * N + n) = v;
  }
}

// pooled sample vectors v[i][c] = mean(flat_i[c*T .. c*T+T-1]), flat_i = x[i] viewed as T*C floats; and the
// update strengths s[i][node] = la * exp(-d2(node, bmu_i) / (2 sigma^2)), d2 = periodic grid distance^2
__global__ void som_prepare_kernel(const float* x, const long long* idx, float* v, float* strength, int B, int T, int C,
                                   int Nn, int gm, int gn, float la, float inv2s2, int periodic) {
  const int i = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float* p = x + ((size_t)i * T * C) + (size_t)c * T;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += p[t];
    v[(size_t)i * C + c] = s / (float)T;
  }
  const int w = (int)idx[i];  // BMU of FLAT token i (reference quirk)
  const int bi = w / gn, bj = w % gn;
  for (int n = threadIdx.x; n < Nn; n += blockDim.x) {
    const int ni = n / gn, nj = n % gn;
    float best = INFINITY;
    const int oi[9] = {0, -gm, gm, -gm, gm, 0, 0, -gm, gm};
    const int oj[9] = {0, -gn, gn, 0, 0, -gn, gn, gn, -gn};
    const int nk = periodic ? 9 : 1;   // non-periodic map: the un-shifted grid only (kohonen.py:95-96)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float di = (float)(ni + oi[k] - bi), dj = (float)(nj + oj[k] - bj);
      if (k < nk) best = fminf(best, di * di + dj * dj);
    }
    strength[(size_t)i * Nn + n] = la * expf(-best * inv2s2);
  }
}

__global__ void som_update_kernel(float* nodes, const float* v, const float* strength, int B, int C, int Nn) {
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float val = nodes[(size_t)n * C + c];
    for (int i = 0; i < B; ++i) {
      const float s = strength[(size_t)i * Nn + n];
      val = val + s * (v[(size_t)i * C + c] - val);
    }
    nodes[(size_t)n * C + c] = val;
  }
}

// ---- consistency loss 1 - mean cos(a_m, b_m): per-block partial sums of cos; backward per row
__global__ __launch_bounds__(256) void cos_fwd_kernel(const float* a, const float* b, float* part, float* stats, int M,
                                                      int C) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float acc = 0.f;
  for (int m = blockIdx.x * 4 + wid; m < M; m += gridDim.x * 4) {
    float saa = 0.f, sbb = 0.f, sab = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(a + (size_t)m * C + c);
      const f32x4 y = *reinterpret_cast<const f32x4*>(b + (size_t)m * C + c);
      saa += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
      sbb += y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
      sab += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
    saa = wave_sum(saa);
    sbb = wave_sum(sbb);
    sab = wave_sum(sab);
    const float ra = 1.0f / sqrtf(saa), rb = 1.0f / sqrtf(sbb);
    const float cs = sab * ra * rb;
    if (lane == 0) {
      stats[(size_t)m * 3] = ra;
      stats[(size_t)m * 3 + 1] = rb;
      stats[(size_t)m * 3 + 2] = cs;
    }
    acc += cs;
  }
  if (lane == 0) red[wid] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// da = gs * (b_hat - a_hat cos) / |a| ; db symmetric ; gs = -g / M
__global__ __launch_bounds__(256) void cos_bwd_kernel(const float* a, const float* b, const float* stats, const float* g,
                                                      float* da, float* db, int M, int C) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float gs = -g[0] / (float)M;
  for (int m = blockIdx.x * 4 + wid; m < M; m += gridDim.x * 4) {
    const float ra = stats[(size_t)m * 3], rb = stats[(size_t)m * 3 + 1], cs = stats[(size_t)m * 3 + 2];
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(a + (size_t)m * C + c) * ra;
      const f32x4 y = *reinterpret_cast<const f32x4*>(b + (size_t)m * C + c) * rb;
      *reinterpret_cast<f32x4*>(da + (size_t)m * C + c) = (y - x * cs) * (ra * gs);
      *reinterpret_cast<f32x4*>(db + (size_t)m * C + c) = (x - y * cs) * (rb * gs);
    }
  }
}

// ---- huber loss (delta = 1, mean)
__global__ __launch_bounds__(256) void huber_fwd_kernel(const float* a, const float* b, float* part, long long n4) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(a + i * 4) - *reinterpret_cast<const f32x4*>(b + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ad = fabsf(d[e]);
      s += ad < 1.0f ? 0.5f * d[e] * d[e] : ad - 0.5f;
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void huber_bwd_kernel(const float* a, const float* b, const float* g, float inv_n, float* da, float* db,
                                 long long n4) {
  const float gs = g[0] * inv_n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(a + i * 4) - *reinterpret_cast<const f32x4*>(b + i * 4);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = fminf(1.0f, fmaxf(-1.0f, d[e])) * gs;
    *reinterpret_cast<f32x4*>(da + i * 4) = r;
    *reinterpret_cast<f32x4*>(db + i * 4) = -r;
  }
}

__global__ void sum_scale_kernel(const float* part, int n, float scale, float bias, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += part[i];
    out[0] = bias + s * scale;
  }
}

// ---- map smoothness: mean over tokens and 8 neighbours of ||node[idx] - node[nb]||  (model.py:503-561)
// Histogram of the BMU indices.  Early in training most tokens pick the same node, so one atomic per token would
// serialise on a single address (measured 0.54 ms for 100k tokens): each wave first groups equal values with ballots
// (one LDS add per distinct value), each workgroup then flushes its LDS bins with one global atomic per non-empty bin.
__global__ __launch_bounds__(256) void hist_kernel(const long long* idx, long long M, int* cnt, int Nn) {
  extern __shared__ int bins[];
  for (int n = threadIdx.x; n < Nn; n += 256) bins[n] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (long long base = blockIdx.x * 256ll; base < M; base += (long long)gridDim.x * 256) {
    const long long i = base + threadIdx.x;
    int v = i < M ? (int)idx[i] : -1;
    unsigned long long todo = __ballot(v >= 0);
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const int lv = __shfl(v, leader, 64);
      const unsigned long long same = __ballot(v == lv) & todo;
      if (lane == leader) atomicAdd(&bins[lv], __builtin_popcountll(same));
      todo &= ~same;
    }
  }
  __syncthreads();
  for (int n = threadIdx.x; n < Nn; n += 256)
    if (bins[n]) atomicAdd(&cnt[n], bins[n]);
}

__device__ __forceinline__ int nb_of(int n, int k, int ms) {
  const int dr[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
  const int dc[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
  const int r = (n / ms + dr[k] + ms) % ms, c = (n % ms + dc[k] + ms) % ms;
  return r * ms + c;
}

// D[n][k] = ||node_n - node_nb(n,k)||; one wave per (n,k)
__global__ void smooth_dist_kernel(const float* nodes, int Nn, int C, int ms, float* D) {
  const int n = blockIdx.x >> 3, k = blockIdx.x & 7;
  const int nb = nb_of(n, k, ms);
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 64) {
    const float d = nodes[(size_t)n * C + c] - nodes[(size_t)nb * C + c];
    s += d * d;
  }
  s = wave_sum(s);
  if (threadIdx.x == 0) D[n * 8 + k] = sqrtf(s);
}

__global__ void smooth_loss_kernel(const float* D, const int* cnt, int Nn, float inv, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int n = 0; n < Nn; ++n) {
      float t = 0.f;
      for (int k = 0; k < 8; ++k) t += D[n * 8 + k];
      s += (float)cnt[n] * t;
    }
    out[0] = s * inv;
  }
}

// dnode_n = gs * sum_k (cnt[n] + cnt[nb_k(n)]) / D[n][k] * (node_n - node_nb)   (the 8-neighbourhood is symmetric)
__global__ void smooth_bwd_kernel(const float* nodes, const float* D, const int* cnt, const float* g, float inv, int Nn,
                                  int C, int ms, float* dnodes, int accumulate) {
  const int n = blockIdx.x;
  const float gs = g[0] * inv;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    const float vn = nodes[(size_t)n * C + c];
    for (int k = 0; k < 8; ++k) {
      const int nb = nb_of(n, k, ms);
      const float d = D[n * 8 + k];
      if (d > 0.f) s += (float)(cnt[n] + cnt[nb]) / d * (vn - nodes[(size_t)nb * C + c]);
    }
    float* o = dnodes + (size_t)n * C + c;
    *o = accumulate ? *o + gs * s : gs * s;
  }
}

// ---- reconstruction loss gradient: draw = g * 2 (tanh(raw) - tgt) (1 - tanh^2) / numel
template <typename T>
__global__ void recon_bwd_kernel(const float* raw, const float* img, const float* g, float inv_n, T* draw, int B, int ch,
                                 int S, int P) {
  const int G = S / P, Tn = G * G, K = ch * P * P;
  const long long total = (long long)B * Tn * K;
  const float gs = 2.0f * g[0] * inv_n;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % K);
    const long long m = idx / K;
    const int b = (int)(m / Tn), t = (int)(m % Tn), ty = t / G, tx = t % G;
    const int c = k / (P * P), ph = (k / P) % P, pw = k % P;
    const float tgt = img[(((size_t)b * ch + c) * S + ty * P + ph) * S + tx * P + pw];
    const float r = tanhf(raw[idx]);
    draw[idx] = (T)(gs * (r - tgt) * (1.0f - r * r));
  }
}

int grid_for(long long n, int per = 256, int cap = 4096) {
  long long b = (n + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int nvit_som_bmu(const float* scores, const float* nodes, float* nn_ws, int64_t M, int N, int C,
                            int64_t* idx, void* stream) {
  NVIT_REQUIRE(M > 0 && N > 0 && C > 0, "som_bmu: empty problem");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_MISC, 0.0, (double)M * N * 4.0, s);
  hipLaunchKernelGGL(rownorm2_kernel, dim3(N), dim3(64), 0, s, nodes, N, C, nn_ws);
  NVIT_CHECK_LAUNCH("rownorm2");
  hipLaunchKernelGGL(bmu_kernel, dim3(grid_for(M, 4, 2048)), dim3(256), 0, s, scores, nn_ws, (int)M, N, (long long*)idx);
  NVIT_CHECK_LAUNCH("bmu");
  return NVIT_OK;
}

extern "C" int nvit_gather_rows(const float* nodes, const int64_t* idx, float* out, int64_t M, int C, void* stream) {
  NVIT_REQUIRE(C % 4 == 0, "gather_rows: C must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, s, nodes, (const long long*)idx, out,
                     (long long)M, C);
  NVIT_CHECK_LAUNCH("gather_rows");
  return NVIT_OK;
}

extern "C" int nvit_scatter_rows(const float* dout, const int64_t* idx, float* dnodes, int64_t M, int N, int C,
                                 void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_MISC, 0.0, (double)M * C * 4.0, s);
  NVIT_REQUIRE(C <= 2048, "scatter_rows: C must be <= 2048");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(N), dim3(256), 0, s, dout, (const long long*)idx, dnodes, (long long)M,
                     C);
  NVIT_CHECK_LAUNCH("scatter_rows");
  return NVIT_OK;
}

extern "C" int nvit_onehot(const int64_t* idx, float* out, int64_t M, int N, void* stream) {
  NVIT_REQUIRE(N % 4 == 0 && M > 0, "onehot: N must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(onehot_kernel, dim3(grid_for(M * (N / 4))), dim3(256), 0, s, (const long long*)idx, out, (long long)M,
                     N);
  NVIT_CHECK_LAUNCH("onehot");
  return NVIT_OK;
}

extern "C" int nvit_som_update(float* nodes, const float* x, const int64_t* idx, float lr_alpha, float sigma, int gm,
                               int gn, int periodic, float* v_ws, float* s_ws, int B, int T, int C, void* stream) {
  const int Nn = gm * gn;
  NVIT_REQUIRE(B > 0 && T > 0 && C > 0 && Nn > 0 && (int64_t)B <= (int64_t)B * T, "som_update: bad shape");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_MISC, 0.0, (double)B * T * C * 4.0, s);
  hipLaunchKernelGGL(som_prepare_kernel, dim3(B), dim3(256), 0, s, x, (const long long*)idx, v_ws, s_ws, B, T, C, Nn, gm,
                     gn, lr_alpha, 1.0f / (2.0f * sigma * sigma), periodic);
  NVIT_CHECK_LAUNCH("som_prepare");
  hipLaunchKernelGGL(som_update_kernel, dim3(Nn), dim3(256), 0, s, nodes, v_ws, s_ws, B, C, Nn);
  NVIT_CHECK_LAUNCH("som_update");
  return NVIT_OK;
}

extern "C" int nvit_cos_consistency_fwd(const float* a, const float* b, float* stats, float* part, int nblk, float* loss,
                                        int64_t M, int C, void* stream) {
  NVIT_REQUIRE(C % 4 == 0 && nblk > 0 && nblk <= 4096, "cos_consistency_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(cos_fwd_kernel, dim3(nblk), dim3(256), 0, s, a, b, part, stats, (int)M, C);
  NVIT_CHECK_LAUNCH("cos_fwd");
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(64), 0, s, part, nblk, -1.0f / (float)M, 1.0f, loss);
  NVIT_CHECK_LAUNCH("cos_sum");
  return NVIT_OK;
}

extern "C" int nvit_cos_consistency_bwd(const float* a, const float* b, const float* stats, const float* g, float* da,
                                        float* db, int64_t M, int C, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(cos_bwd_kernel, dim3(grid_for(M, 4, 2048)), dim3(256), 0, s, a, b, stats, g, da, db, (int)M, C);
  NVIT_CHECK_LAUNCH("cos_bwd");
  return NVIT_OK;
}

extern "C" int nvit_huber_fwd(const float* a, const float* b, float* part, int nblk, float* loss, int64_t n,
                              void* stream) {
  NVIT_REQUIRE(n % 4 == 0 && nblk > 0 && nblk <= 4096, "huber_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(huber_fwd_kernel, dim3(nblk), dim3(256), 0, s, a, b, part, (long long)(n / 4));
  NVIT_CHECK_LAUNCH("huber_fwd");
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(64), 0, s, part, nblk, 1.0f / (float)n, 0.0f, loss);
  NVIT_CHECK_LAUNCH("huber_sum");
  return NVIT_OK;
}

extern "C" int nvit_huber_bwd(const float* a, const float* b, const float* g, float* da, float* db, int64_t n,
                              void* stream) {
  NVIT_REQUIRE(n % 4 == 0, "huber_bwd: n must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(huber_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, a, b, g, 1.0f / (float)n, da, db,
                     (long long)(n / 4));
  NVIT_CHECK_LAUNCH("huber_bwd");
  return NVIT_OK;
}

extern "C" int nvit_som_smooth_fwd(const float* nodes, const int64_t* idx, int* cnt, float* D, float* loss, int64_t M,
                                   int Nn, int C, int map_size, void* stream) {
  NVIT_REQUIRE(map_size * map_size == Nn, "som_smooth: nodes per map must be a perfect square");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(cnt, 0, sizeof(int) * Nn, s);
  if (e != hipSuccess) NVIT_FAIL((int)e, "som_smooth: memset: %s", hipGetErrorString(e));
  {
    long long hb = (M + 255) / 256;
    if (hb > 256) hb = 256;
    hipLaunchKernelGGL(hist_kernel, dim3((unsigned)hb), dim3(256), sizeof(int) * Nn, s, (const long long*)idx, (long long)M, cnt, Nn);
  }
  NVIT_CHECK_LAUNCH("hist");
  hipLaunchKernelGGL(smooth_dist_kernel, dim3(Nn * 8), dim3(64), 0, s, nodes, Nn, C, map_size, D);
  NVIT_CHECK_LAUNCH("smooth_dist");
  hipLaunchKernelGGL(smooth_loss_kernel, dim3(1), dim3(64), 0, s, D, cnt, Nn, 1.0f / (8.0f * (float)M), loss);
  NVIT_CHECK_LAUNCH("smooth_loss");
  return NVIT_OK;
}

extern "C" int nvit_som_smooth_bwd(const float* nodes, const float* D, const int* cnt, const float* g, float* dnodes,
                                   int accumulate, int64_t M, int Nn, int C, int map_size, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(smooth_bwd_kernel, dim3(Nn), dim3(256), 0, s, nodes, D, cnt, g, 1.0f / (8.0f * (float)M), Nn, C,
                     map_size, dnodes, accumulate);
  NVIT_CHECK_LAUNCH("smooth_bwd");
  return NVIT_OK;
}

extern "C" int nvit_recon_bwd(int dt, const float* raw, const float* img, const float* g, void* draw, int B, int ch,
                              int S, int P, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)B * ch * S * S;
  const float inv_n = 1.0f / (float)total;
  if (dt == NVIT_F32)
    hipLaunchKernelGGL(recon_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, raw, img, g, inv_n, (float*)draw,
                       B, ch, S, P);
  else
    hipLaunchKernelGGL(recon_bwd_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, raw, img, g, inv_n, (bf16*)draw,
                       B, ch, S, P);
  NVIT_CHECK_LAUNCH("recon_bwd");
  return NVIT_OK;
}
