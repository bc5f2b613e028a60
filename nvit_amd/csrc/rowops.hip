// Row-wise fp32 kernels on the residual stream: normalised LERP (+norm_skip) forward/backward,
// per-head cosine normalisation forward/backward, SwiGLU gate forward/backward, and the small
// column reductions that turn per-block partial sums into parameter gradients.
// HBM-bound: one wave owns one row (16-byte loads, all reductions by lane shuffles), rows are
// grid-strided, per-column partial sums stay in registers until the wave's last row.
#include "common.h"

namespace {

constexpr int ROW_WAVES = 4;  // waves (rows in flight) per 256-thread workgroup

// Each lane owns NV float4 groups of a row: columns (i*64 + lane)*4 .. +3, i < NV, while < C.
template <int NV>
struct RowVec {
  f32x4 v[NV];
};

template <int NV, typename T>
__device__ __forceinline__ void row_load(RowVec<NV>& r, const T* p, int C, int lane) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    r.v[i] = c < C ? load4<T>(p + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}
template <int NV, typename T>
__device__ __forceinline__ void row_store(const RowVec<NV>& r, T* p, int C, int lane) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) store4<T>(p + c, r.v[i]);
  }
}
template <int NV>
__device__ __forceinline__ float row_dot(const RowVec<NV>& a, const RowVec<NV>& b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += a.v[i][0] * b.v[i][0] + a.v[i][1] * b.v[i][1] + a.v[i][2] * b.v[i][2] + a.v[i][3] * b.v[i][3];
  return wave_sum(s);
}

// Streaming (non-temporal) loads of read-once fp32 rows: the residual stream and the incoming gradient are not read again
// soon, keeping them out of the caches leaves room for what the neighbouring GEMMs re-read (measured on the whole step:
// backward -0.34 ms in round 3, forward -0.35 ms in round 4; streaming STORES of the backward outputs: +-0, not used).
template <int NV>
__device__ __forceinline__ void row_load_f32_nt(RowVec<NV>& r, const float* p, int C, int lane) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    r.v[i] = c < C ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + c)) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}

// ------------------------------------------------------------------------------ LERP forward
struct LerpFwdArgs {
  const float* h;
  const void* y;
  const float* alpha;
  float c_a;
  const float* skip_x;
  const float* skip;
  float* out;
  void* out_lo;
  int M, C;
};

template <int NV, typename TY, typename TL>
__global__ __launch_bounds__(256) void lerp_fwd_kernel(LerpFwdArgs a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  RowVec<NV> lam;
  row_load<NV, float>(lam, a.alpha, a.C, lane);
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) lam.v[i][e] = fabsf(lam.v[i][e] * a.c_a);
  const float skip = a.skip_x ? a.skip[0] : 0.f;
  for (int m = blockIdx.x * ROW_WAVES + wid; m < a.M; m += gridDim.x * ROW_WAVES) {
    RowVec<NV> x, y, r, xs;
    row_load_f32_nt<NV>(x, a.h + (size_t)m * a.C, a.C, lane);   // (the residual stream is read once here: streaming load)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if constexpr (sizeof(TY) == 2) {
        uint2 raw = make_uint2(0u, 0u);
        if (c < a.C) {
          typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
          raw = __builtin_bit_cast(uint2, __builtin_nontemporal_load(reinterpret_cast<const u32x2_*>(reinterpret_cast<const TY*>(a.y) + (size_t)m * a.C + c)));
        }
        const bf16x4 b4 = __builtin_bit_cast(bf16x4, raw);
        y.v[i] = (f32x4){(float)b4[0], (float)b4[1], (float)b4[2], (float)b4[3]};
      } else {
        y.v[i] = c < a.C ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.y) + (size_t)m * a.C + c)) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    if (a.skip_x) row_load_f32_nt<NV>(xs, a.skip_x + (size_t)m * a.C, a.C, lane);   // with the others: one round trip per row
    const float rsx = 1.0f / sqrtf(row_dot<NV>(x, x));
    const float rsy = 1.0f / sqrtf(row_dot<NV>(y, y));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const f32x4 av = x.v[i] * rsx, bv = y.v[i] * rsy;
      r.v[i] = av + lam.v[i] * (bv - av);
    }
    const float rsr = 1.0f / sqrtf(row_dot<NV>(r, r));
#pragma unroll
    for (int i = 0; i < NV; ++i) r.v[i] = r.v[i] * rsr;
    if (a.skip_x) {
#pragma unroll
      for (int i = 0; i < NV; ++i) r.v[i] = r.v[i] * skip + xs.v[i];
      const float rst = 1.0f / sqrtf(row_dot<NV>(r, r));
#pragma unroll
      for (int i = 0; i < NV; ++i) r.v[i] = r.v[i] * rst;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {   // the fp32 stream is read again only far later (next block / backward): streaming store
      const int c = (i * 64 + lane) * 4;
      if (c < a.C) __builtin_nontemporal_store(r.v[i], reinterpret_cast<f32x4*>(a.out + (size_t)m * a.C + c));
    }
    if (a.out_lo) row_store<NV, TL>(r, reinterpret_cast<TL*>(a.out_lo) + (size_t)m * a.C, a.C, lane);
  }
}

// ------------------------------------------------------------------------------ LERP backward
struct LerpBwdArgs {
  const float* dout;
  const bf16* dout_add;   // optional second addend of the incoming gradient (a data-gradient GEMM's bf16 output), or NULL
  const float* h;
  const void* y;
  const float* alpha;
  float c_a;
  const float* skip_x;
  const float* skip;
  float* dh;
  int accum_dh;
  float* dy;
  void* dy_lo;
  float* dskip_x;
  float* part_dlam;
  float* part_dskip;
  int M, C;
};

// ADD: the incoming gradient has a second, bf16 addend (a.dout_add).  A template switch, not a run-time test: with the test
// inside the row loop hipcc waits for the addend right where it is loaded, before the remaining loads of the row have
// been issued - a second memory round trip per row (+137 us per Base block, measured).
//
// Register-lean form: per row only the two unit vectors a, b and the running gradient g stay in registers (plus the
// per-column step sizes and their gradient sums); the LERP output o = (a + lam*(b - a)) * rsr is recomputed where it is
// needed (three flops per element) and da / db are formed at the store.  No LDS: every wave writes its own row of
// column partials (part_dlam [4*nblk, C], part_dskip [4*nblk]).  At C = 768 this is 104 registers instead of 160
// (4 waves per SIMD instead of 3) - small enough to be co-resident with two waves of the 193-register weight-gradient
// GEMM on the same SIMD.
template <int NV>
__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}

// SKIP (norm_skip fused behind the LERP: the MLP half of a block) and ACCUM (dh += : the attention half, whose dh already
// holds the gradient of the skip path) are template switches as well: a row then keeps 5 vectors in flight instead of 6.
template <int NV, typename TY, typename TL, bool ADD, bool SKIP, bool ACCUM>
__global__ __launch_bounds__(256, 1) void lerp_bwd_kernel(LerpBwdArgs a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  RowVec<NV> lam, dlam;
  row_load<NV, float>(lam, a.alpha, a.C, lane);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    dlam.v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) lam.v[i][e] = fabsf(lam.v[i][e] * a.c_a);
  }
  const float skip = SKIP ? a.skip[0] : 0.f;
  float dskip_acc = 0.f;
  for (int m = blockIdx.x * ROW_WAVES + wid; m < a.M; m += gridDim.x * ROW_WAVES) {
    RowVec<NV> av, bv, g, t, old;   // av: h then a = nrm(h); bv: y then b = nrm(y); g: incoming gradient, then d(lerp out), then dr
    uint2 ga[NV];                    // the bf16 addend stays packed until it is added (half the registers in flight)
    const size_t ro = (size_t)m * a.C;
    // every load of the row goes out before the first reduction: one memory round trip per row, not two or three
    row_load_f32_nt<NV>(av, a.h + ro, a.C, lane);
    row_load<NV, TY>(bv, reinterpret_cast<const TY*>(a.y) + ro, a.C, lane);
    row_load_f32_nt<NV>(g, a.dout + ro, a.C, lane);
    if constexpr (ADD) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        ga[i] = c < a.C ? *reinterpret_cast<const uint2*>(a.dout_add + ro + c) : make_uint2(0u, 0u);
      }
    }
    if constexpr (SKIP) row_load_f32_nt<NV>(t, a.skip_x + ro, a.C, lane);
    if constexpr (ADD) {   // incoming gradient = dout + dout_add: the GEMM that produced dout_add need not read-modify-write dout
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const bf16x4 q = __builtin_bit_cast(bf16x4, ga[i]);
        g.v[i] += (f32x4){(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
      }
    }
    const float rsx = 1.0f / sqrtf(row_dot<NV>(av, av));
    const float rsy = 1.0f / sqrtf(row_dot<NV>(bv, bv));
    // (the old dh values are needed only at the store: requested here, they arrive under the remaining reductions)
    if constexpr (ACCUM) row_load<NV, float>(old, a.dh + ro, a.C, lane);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      av.v[i] = av.v[i] * rsx;
      bv.v[i] = bv.v[i] * rsy;
      const f32x4 ou = av.v[i] + lam.v[i] * (bv.v[i] - av.v[i]);
      ss += dot4<NV>(ou, ou);
    }
    const float rsr = 1.0f / sqrtf(wave_sum(ss));
#define NVIT_LERP_O(i_) ((av.v[i_] + lam.v[i_] * (bv.v[i_] - av.v[i_])) * rsr)   /* o = lerp output (unit norm), recomputed */
    if constexpr (SKIP) {
      // t = o*skip + xs ; out = t/|t| ; dt = (g - out<out,g>)/|t| ; do = skip*dt ; dxs = dt ; dskip += <dt,o>
      float st = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        t.v[i] = NVIT_LERP_O(i) * skip + t.v[i];
        st += dot4<NV>(t.v[i], t.v[i]);
      }
      const float rst = 1.0f / sqrtf(wave_sum(st));
      float tg = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        t.v[i] = t.v[i] * rst;
        tg += dot4<NV>(t.v[i], g.v[i]);
      }
      tg = wave_sum(tg);
      float dso = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        g.v[i] = (g.v[i] - t.v[i] * tg) * rst;   // g = dt
        dso += dot4<NV>(g.v[i], NVIT_LERP_O(i));
      }
      row_store<NV, float>(g, a.dskip_x + ro, a.C, lane);
      dskip_acc += wave_sum(dso);
#pragma unroll
      for (int i = 0; i < NV; ++i) g.v[i] = g.v[i] * skip;   // g = d(lerp output)
    }
    // dr = (g - o<o,g>) * rsr
    float og = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) og += dot4<NV>(NVIT_LERP_O(i), g.v[i]);
    og = wave_sum(og);
    float ada = 0.f, bdb = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      g.v[i] = (g.v[i] - NVIT_LERP_O(i) * og) * rsr;   // g = dr
      dlam.v[i] += g.v[i] * (bv.v[i] - av.v[i]);
      const f32x4 db = lam.v[i] * g.v[i];
      ada += dot4<NV>(av.v[i], g.v[i] - db);
      bdb += dot4<NV>(bv.v[i], db);
    }
#undef NVIT_LERP_O
    ada = wave_sum(ada);
    bdb = wave_sum(bdb);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      const f32x4 db = lam.v[i] * g.v[i];
      f32x4 dh = ((g.v[i] - db) - av.v[i] * ada) * rsx;
      const f32x4 dy = (db - bv.v[i] * bdb) * rsy;
      if constexpr (ACCUM) dh += old.v[i];
      if (c < a.C) {
        store4<float>(a.dh + ro + c, dh);
        if (a.dy) store4<float>(a.dy + ro + c, dy);
        if (a.dy_lo) store4<TL>(reinterpret_cast<TL*>(a.dy_lo) + ro + c, dy);
      }
    }
  }
  // column partials: one row per wave (fixed layout, reduced in fixed order by nvit_colsum_reduce)
  const size_t prow = (size_t)blockIdx.x * ROW_WAVES + wid;
  row_store<NV, float>(dlam, a.part_dlam + prow * a.C, a.C, lane);
  if (SKIP && lane == 0) a.part_dskip[prow] = dskip_acc;
}

// ------------------------------------------------------------------------------ standalone norm_skip
// out = nrm(source*skip + target)  (Block.norm_skip, reference model.py:84-87) for callers that use it on its own;
// ViT.forward uses the version fused into lerp_fwd/bwd.
template <int NV>
__global__ __launch_bounds__(256) void norm_skip_fwd_kernel(const float* src, const float* tgt, const float* skip,
                                                             float* out, int M, int C) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float sk = skip[0];
  for (int m = blockIdx.x * ROW_WAVES + wid; m < M; m += gridDim.x * ROW_WAVES) {
    RowVec<NV> a, b;
    row_load<NV, float>(a, src + (size_t)m * C, C, lane);
    if (tgt) {   // tgt == NULL: plain justnorm(source * skip)
      row_load<NV, float>(b, tgt + (size_t)m * C, C, lane);
#pragma unroll
      for (int i = 0; i < NV; ++i) a.v[i] = a.v[i] * sk + b.v[i];
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) a.v[i] = a.v[i] * sk;
    }
    const float rs = 1.0f / sqrtf(row_dot<NV>(a, a));
#pragma unroll
    for (int i = 0; i < NV; ++i) a.v[i] = a.v[i] * rs;
    row_store<NV, float>(a, out + (size_t)m * C, C, lane);
  }
}

template <int NV>
__global__ __launch_bounds__(256) void norm_skip_bwd_kernel(const float* dout, const float* src, const float* tgt,
                                                             const float* skip, float* dsrc, float* dtgt,
                                                             float* part_dskip, int M, int C) {
  __shared__ float red[ROW_WAVES];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float sk = skip[0];
  float acc = 0.f;
  for (int m = blockIdx.x * ROW_WAVES + wid; m < M; m += gridDim.x * ROW_WAVES) {
    RowVec<NV> a, b, g;
    row_load<NV, float>(a, src + (size_t)m * C, C, lane);
    if (tgt) {
      row_load<NV, float>(b, tgt + (size_t)m * C, C, lane);
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) b.v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    row_load<NV, float>(g, dout + (size_t)m * C, C, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) b.v[i] = a.v[i] * sk + b.v[i];
    const float rs = 1.0f / sqrtf(row_dot<NV>(b, b));
#pragma unroll
    for (int i = 0; i < NV; ++i) b.v[i] = b.v[i] * rs;
    const float og = row_dot<NV>(b, g);
#pragma unroll
    for (int i = 0; i < NV; ++i) g.v[i] = (g.v[i] - b.v[i] * og) * rs;  // d(source*skip + target)
    acc += row_dot<NV>(g, a);
    if (dtgt) row_store<NV, float>(g, dtgt + (size_t)m * C, C, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) g.v[i] = g.v[i] * sk;
    row_store<NV, float>(g, dsrc + (size_t)m * C, C, lane);
  }
  if (lane == 0) red[wid] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part_dskip[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------ q/k normalise
struct QkArgs {
  const void *q, *k, *v;
  int ldq, ldk, ldv;
  const float* sqk;
  float c_q;
  void *qh, *kh, *vh;
  float *rq, *rk;
  int B, T, H, d;
};

template <int NV, typename TI, typename T>   // TI: type of the projection outputs read, T: type of the head tensors written
__global__ __launch_bounds__(256) void qknorm_fwd_kernel(QkArgs a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int C = a.H * a.d, M = a.B * a.T, G = a.d >> 2;  // lanes per head
  RowVec<NV> s;
  row_load<NV, float>(s, a.sqk, C, lane);
#pragma unroll
  for (int i = 0; i < NV; ++i) s.v[i] = s.v[i] * a.c_q;
  for (int m = blockIdx.x * ROW_WAVES + wid; m < M; m += gridDim.x * ROW_WAVES) {
    const int b = m / a.T, t = m % a.T;
    RowVec<NV> q, k, v;
    row_load<NV, TI>(q, reinterpret_cast<const TI*>(a.q) + (size_t)m * a.ldq, C, lane);
    row_load<NV, TI>(k, reinterpret_cast<const TI*>(a.k) + (size_t)m * a.ldk, C, lane);
    row_load<NV, TI>(v, reinterpret_cast<const TI*>(a.v) + (size_t)m * a.ldv, C, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      float sq = q.v[i][0] * q.v[i][0] + q.v[i][1] * q.v[i][1] + q.v[i][2] * q.v[i][2] + q.v[i][3] * q.v[i][3];
      float sk = k.v[i][0] * k.v[i][0] + k.v[i][1] * k.v[i][1] + k.v[i][2] * k.v[i][2] + k.v[i][3] * k.v[i][3];
      sq = group_sum_dyn(sq, G);
      sk = group_sum_dyn(sk, G);
      if (c < C) {
        const float rq = 1.0f / sqrtf(sq), rk = 1.0f / sqrtf(sk);
        const int h = c / a.d, j = c % a.d;
        const size_t dst = (((size_t)b * a.H + h) * a.T + t) * a.d + j;
        store4<T>(reinterpret_cast<T*>(a.qh) + dst, q.v[i] * rq * s.v[i]);
        store4<T>(reinterpret_cast<T*>(a.kh) + dst, k.v[i] * rk * s.v[i]);
        store4<T>(reinterpret_cast<T*>(a.vh) + dst, v.v[i]);
        if (j == 0) {
          a.rq[(size_t)m * a.H + h] = rq;
          a.rk[(size_t)m * a.H + h] = rk;
        }
      }
    }
  }
}

struct QkBwdArgs {
  const void *dqh, *dkh, *dvh, *qh, *kh;
  const float *rq, *rk, *sqk;
  float c_q;
  void *dq, *dk, *dv;
  int ldq, ldk, ldv;
  float* part;
  int B, T, H, d;
};

template <int NV, typename T>
__global__ __launch_bounds__(256) void qknorm_bwd_kernel(QkBwdArgs a) {
  __shared__ float red[ROW_WAVES][NV * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int C = a.H * a.d, M = a.B * a.T, G = a.d >> 2;
  RowVec<NV> s, sinv, ds;
  row_load<NV, float>(s, a.sqk, C, lane);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    s.v[i] = s.v[i] * a.c_q;
    ds.v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) sinv.v[i][e] = s.v[i][e] != 0.f ? 1.0f / s.v[i][e] : 0.f;
  }
  for (int m = blockIdx.x * ROW_WAVES + wid; m < M; m += gridDim.x * ROW_WAVES) {
    const int b = m / a.T, t = m % a.T;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      const bool ok = c < C;
      const int h = ok ? c / a.d : 0, j = ok ? c % a.d : 0;
      const size_t src = (((size_t)b * a.H + h) * a.T + t) * a.d + j;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 gq = ok ? load4<T>(reinterpret_cast<const T*>(a.dqh) + src) : z;
      const f32x4 gk = ok ? load4<T>(reinterpret_cast<const T*>(a.dkh) + src) : z;
      const f32x4 nq = ok ? load4<T>(reinterpret_cast<const T*>(a.qh) + src) * sinv.v[i] : z;  // unit vector
      const f32x4 nk = ok ? load4<T>(reinterpret_cast<const T*>(a.kh) + src) * sinv.v[i] : z;
      ds.v[i] += gq * nq + gk * nk;
      const f32x4 sgq = gq * s.v[i], sgk = gk * s.v[i];
      float dq_ = sgq[0] * nq[0] + sgq[1] * nq[1] + sgq[2] * nq[2] + sgq[3] * nq[3];
      float dk_ = sgk[0] * nk[0] + sgk[1] * nk[1] + sgk[2] * nk[2] + sgk[3] * nk[3];
      dq_ = group_sum_dyn(dq_, G);
      dk_ = group_sum_dyn(dk_, G);
      if (ok) {
        const float rq = a.rq[(size_t)m * a.H + h], rk = a.rk[(size_t)m * a.H + h];
        store4<T>(reinterpret_cast<T*>(a.dq) + (size_t)m * a.ldq + c, (sgq - nq * dq_) * rq);
        store4<T>(reinterpret_cast<T*>(a.dk) + (size_t)m * a.ldk + c, (sgk - nk * dk_) * rk);
        store4<T>(reinterpret_cast<T*>(a.dv) + (size_t)m * a.ldv + c,
                  load4<T>(reinterpret_cast<const T*>(a.dvh) + src));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wid][(i * 64 + lane) * 4 + e] = ds.v[i][e];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < ROW_WAVES; ++w) t += red[w][c];
    a.part[(size_t)blockIdx.x * C + c] = t;
  }
}

// ------------------------------------------------------------------------------ SwiGLU
// thread owns 4 consecutive output columns j..j+3 (inside one 16-group), loops over a row range.
template <typename TI, typename T>   // TI: type of the pre-activations read, T: type of the gated output
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const TI* uv, const float* suv, float gscale, T* x, int M,
                                                          int F, int rows_per_blk) {
  const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (j >= F) return;
  const int q = j >> 4, w = j & 15;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f};
  const f32x4 gu = suv ? *reinterpret_cast<const f32x4*>(suv + j) * gscale : one;
  const f32x4 gv = suv ? *reinterpret_cast<const f32x4*>(suv + F + j) * gscale : one;
  const int r0 = blockIdx.y * rows_per_blk;
  const int r1 = min(M, r0 + rows_per_blk);
  for (int m = r0; m < r1; ++m) {
    const TI* row = uv + (size_t)m * 2 * F + q * 32 + w;
    const f32x4 u = load4<TI>(row) * gu;
    const f32x4 v = load4<TI>(row + 16) * gv;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = u[e] * (v[e] * sigmoidf_(v[e]));
    store4<T>(x + (size_t)m * F + j, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* dx, const T* uv, const float* suv, float gscale,
                                                          T* duv, float* part, int M, int F, int rows_per_blk) {
  const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (j >= F) return;
  const int q = j >> 4, w = j & 15;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f};
  const f32x4 gu = suv ? *reinterpret_cast<const f32x4*>(suv + j) * gscale : one;
  const f32x4 gv = suv ? *reinterpret_cast<const f32x4*>(suv + F + j) * gscale : one;
  f32x4 dgu = {0.f, 0.f, 0.f, 0.f}, dgv = {0.f, 0.f, 0.f, 0.f};
  const int r0 = blockIdx.y * rows_per_blk;
  const int r1 = min(M, r0 + rows_per_blk);
  for (int m = r0; m < r1; ++m) {
    const size_t off = (size_t)m * 2 * F + q * 32 + w;
    const f32x4 ur = load4<T>(uv + off), vr = load4<T>(uv + off + 16);
    const f32x4 g = load4<T>(dx + (size_t)m * F + j);
    const f32x4 u = ur * gu, v = vr * gv;
    f32x4 du, dv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float sg = sigmoidf_(v[e]);
      du[e] = g[e] * v[e] * sg;
      dv[e] = g[e] * u[e] * sg * (1.0f + v[e] * (1.0f - sg));
    }
    dgu += du * ur;
    dgv += dv * vr;
    store4<T>(duv + off, du * gu);
    store4<T>(duv + off + 16, dv * gv);
  }
  if (part) {
    float* p = part + (size_t)blockIdx.y * 2 * F;
    *reinterpret_cast<f32x4*>(p + j) = dgu * gscale;
    *reinterpret_cast<f32x4*>(p + F + j) = dgv * gscale;
  }
}

// ------------------------------------------------------------------------------ small reductions
// out[n] = f(sum_b part[b][n]); 1024 threads = 32 columns x 32 row groups, fixed summation order (the partial arrays
// are a few MB and there are only N/32 workgroups, so the kernel is latency-bound: many short rows per thread)
constexpr int CSR_RG = 32;
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* part, int nblk, int N, float* out,
                                                              int accumulate, int kind, const float* ref, float scale) {
  __shared__ float red[CSR_RG][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (n < N) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = rg;
    for (; b + 3 * CSR_RG < nblk; b += 4 * CSR_RG) {
      s0 += part[(size_t)b * N + n];
      s1 += part[(size_t)(b + CSR_RG) * N + n];
      s2 += part[(size_t)(b + 2 * CSR_RG) * N + n];
      s3 += part[(size_t)(b + 3 * CSR_RG) * N + n];
    }
    for (; b < nblk; b += CSR_RG) s0 += part[(size_t)b * N + n];
    s = (s0 + s1) + (s2 + s3);
  }
  red[rg][cl] = s;
  __syncthreads();
  if (rg != 0 || n >= N) return;
#pragma unroll
  for (int g = 1; g < CSR_RG; ++g) s += red[g][cl];
  if (kind == 1) {
    const float r = ref[n] * scale;
    s = s * (r > 0.f ? scale : (r < 0.f ? -scale : 0.f));
  } else {
    s *= scale;
  }
  int dst = n;
  if (kind == 2) {
    const int q = n >> 5, w = n & 31, F = N >> 1;
    dst = w < 16 ? q * 16 + w : F + q * 16 + (w - 16);
  }
  out[dst] = accumulate ? out[dst] + s : s;
}

// Several reductions of the kind above in ONE launch (the six parameter-gradient reductions of a block's backward were
// six ~12 us launches on the critical stream, 0.9 ms per Base step).  An item may have a second partial array whose
// reduced, scaled sum is added after the first one's (fixed order: e.g. the q and k contributions to d sqk).
constexpr int CSR_MAX_ITEMS = 8;
struct CsrItem {
  const float* part;
  const float* part_b;   // optional second partial array (same N), or NULL
  const float* ref;
  float* out;
  int nblk, nblk_b, N, accumulate, kind, block0;   // block0: first workgroup of this item in the launch
  float scale;
};
struct CsrBatch {
  CsrItem it[CSR_MAX_ITEMS];
  int n;
};

__device__ __forceinline__ float csr_column(const float* part, int nblk, int N, int n, int rg) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = rg;
  for (; b + 3 * CSR_RG < nblk; b += 4 * CSR_RG) {
    s0 += part[(size_t)b * N + n];
    s1 += part[(size_t)(b + CSR_RG) * N + n];
    s2 += part[(size_t)(b + 2 * CSR_RG) * N + n];
    s3 += part[(size_t)(b + 3 * CSR_RG) * N + n];
  }
  for (; b < nblk; b += CSR_RG) s0 += part[(size_t)b * N + n];
  return (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(1024) void colsum_reduce_multi_kernel(CsrBatch bt) {
  __shared__ float red[2][CSR_RG][33];
  int k = 0;
#pragma unroll
  for (int i = 1; i < CSR_MAX_ITEMS; ++i)
    if (i < bt.n && (int)blockIdx.x >= bt.it[i].block0) k = i;
  // (copy of the selected item through a uniform index: the struct lives in kernel-argument memory)
  const float* part = bt.it[k].part;
  const float* part_b = bt.it[k].part_b;
  const float* ref = bt.it[k].ref;
  float* out = bt.it[k].out;
  const int nblk = bt.it[k].nblk, nblk_b = bt.it[k].nblk_b, N = bt.it[k].N, accumulate = bt.it[k].accumulate,
            kind = bt.it[k].kind, blk = (int)blockIdx.x - bt.it[k].block0;
  const float scale = bt.it[k].scale;
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int n = blk * 32 + cl;
  red[0][rg][cl] = n < N ? csr_column(part, nblk, N, n, rg) : 0.f;
  red[1][rg][cl] = (n < N && part_b) ? csr_column(part_b, nblk_b, N, n, rg) : 0.f;
  __syncthreads();
  if (rg != 0 || n >= N) return;
  float s = red[0][0][cl], sb = red[1][0][cl];
#pragma unroll
  for (int g = 1; g < CSR_RG; ++g) {
    s += red[0][g][cl];
    sb += red[1][g][cl];
  }
  if (kind == 1) {
    const float r = ref[n] * scale;
    const float f = r > 0.f ? scale : (r < 0.f ? -scale : 0.f);
    s = s * f;
    sb = sb * f;
  } else {
    s *= scale;
    sb *= scale;
  }
  int dst = n;
  if (kind == 2) {
    const int q = n >> 5, w = n & 31, F = N >> 1;
    dst = w < 16 ? q * 16 + w : F + q * 16 + (w - 16);
  }
  float r = accumulate ? out[dst] + s : s;
  if (part_b) r += sb;
  out[dst] = r;
}

template <typename TA, typename TB>
__global__ void colsum_kernel(const TA* a, int lda, const TB* b, int ldb, int R, int N, int period, float* out,
                              int accumulate, float scale) {
  // one thread per output element (row class rc in [0,period), column n); loops rows r = rc, rc+period, ...
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int rc = blockIdx.y;
  if (n >= N) return;
  // four independent partial sums (fixed order): four loads in flight per thread instead of a dependent chain
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = rc;
  for (; r + 3 * period < R; r += 4 * period) {
    float v0 = ld1<TA>(a + (size_t)r * lda + n);
    float v1 = ld1<TA>(a + (size_t)(r + period) * lda + n);
    float v2 = ld1<TA>(a + (size_t)(r + 2 * period) * lda + n);
    float v3 = ld1<TA>(a + (size_t)(r + 3 * period) * lda + n);
    if (b) {
      v0 *= ld1<TB>(b + (size_t)r * ldb + n);
      v1 *= ld1<TB>(b + (size_t)(r + period) * ldb + n);
      v2 *= ld1<TB>(b + (size_t)(r + 2 * period) * ldb + n);
      v3 *= ld1<TB>(b + (size_t)(r + 3 * period) * ldb + n);
    }
    s0 += v0;
    s1 += v1;
    s2 += v2;
    s3 += v3;
  }
  for (; r < R; r += period) {
    float v = ld1<TA>(a + (size_t)r * lda + n);
    if (b) v *= ld1<TB>(b + (size_t)r * ldb + n);
    s0 += v;
  }
  float s = ((s0 + s1) + (s2 + s3)) * scale;
  float* o = out + (size_t)rc * N + n;
  *o = accumulate ? *o + s : s;
}

template <typename T>
__global__ void cast_kernel(const float* src, T* dst, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    store4<T>(dst + i * 4, *reinterpret_cast<const f32x4*>(src + i * 4));
}

template <typename T>
__global__ void scale_cols_kernel(const float* a, int lda, const float* s, float c, T* out, int ldo, int R, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (n >= N || r >= R) return;
  st1<T>(out + (size_t)r * ldo + n, a[(size_t)r * lda + n] * s[n] * c);
}

// Input pipeline, deterministic part (reference train.py:1084-1090): ToTensor (uint8 HWC -> float CHW in [0,1]) and
// kornia Normalize(mean 0.5, std 0.5) in one pass: out[b,c,y,x] = (in/255 - mean) / std.  IN = uint8 (HWC, one byte per
// channel) or float (CHW, already in [0,1]).  Four output pixels per thread, 16-byte stores.
template <typename IN>
__global__ void normalize_images_kernel(const IN* in, float* out, int B, int C, int H, int W, float mean, float inv_std) {
  const long long n4 = (long long)B * C * H * (W / 4);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int x4 = (int)(i % (W / 4)) * 4;
    long long r = i / (W / 4);
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), b = (int)(r / C);
    f32x4 v;
    if constexpr (sizeof(IN) == 1) {
      const IN* p = in + (((size_t)b * H + y) * W + x4) * C + c;   // HWC
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (float)p[(size_t)e * C] * (1.0f / 255.0f);
    } else {
      v = *reinterpret_cast<const f32x4*>(in + (((size_t)b * C + c) * H + y) * W + x4);
    }
    *reinterpret_cast<f32x4*>(out + (((size_t)b * C + c) * H + y) * W + x4) = (v - mean) * inv_std;
  }
}

// ------------------------------------------------------------------------------ RMSNorm (reference model.py:170-182)
// y = x * rsqrt(mean(x^2) + eps) * w.  Dead on the nViT path (its Block never calls it); kept as a working public module.
template <int NV>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* x, const float* w, float eps, float* out,
                                                           float* rstd, int M, int C) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  RowVec<NV> wv;
  row_load<NV, float>(wv, w, C, lane);
  for (int m = blockIdx.x * ROW_WAVES + wid; m < M; m += gridDim.x * ROW_WAVES) {
    RowVec<NV> a;
    row_load<NV, float>(a, x + (size_t)m * C, C, lane);
    const float rs = 1.0f / sqrtf(row_dot<NV>(a, a) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) a.v[i] = a.v[i] * rs * wv.v[i];
    row_store<NV, float>(a, out + (size_t)m * C, C, lane);
    if (lane == 0) rstd[m] = rs;
  }
}

// dx = rstd * (g*w - xn * mean(g*w*xn)),  xn = x * rstd;  part_dw[block][c] = sum over the block's rows of g * xn
template <int NV>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* dout, const float* x, const float* w,
                                                           const float* rstd, float* dx, float* part_dw, int M, int C) {
  __shared__ float red[ROW_WAVES][NV * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  RowVec<NV> wv, dw;
  row_load<NV, float>(wv, w, C, lane);
#pragma unroll
  for (int i = 0; i < NV; ++i) dw.v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int m = blockIdx.x * ROW_WAVES + wid; m < M; m += gridDim.x * ROW_WAVES) {
    RowVec<NV> a, g;
    row_load<NV, float>(a, x + (size_t)m * C, C, lane);
    row_load<NV, float>(g, dout + (size_t)m * C, C, lane);
    const float rs = rstd[m];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      a.v[i] = a.v[i] * rs;            // xn
      dw.v[i] += g.v[i] * a.v[i];
      g.v[i] = g.v[i] * wv.v[i];       // g*w
    }
    const float mean = row_dot<NV>(g, a) / (float)C;
#pragma unroll
    for (int i = 0; i < NV; ++i) g.v[i] = (g.v[i] - a.v[i] * mean) * rs;
    row_store<NV, float>(g, dx + (size_t)m * C, C, lane);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wid][(i * 64 + lane) * 4 + e] = dw.v[i][e];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s_ = 0.f;
#pragma unroll
    for (int wv_ = 0; wv_ < ROW_WAVES; ++wv_) s_ += red[wv_][c];
    part_dw[(size_t)blockIdx.x * C + c] = s_;
  }
}

int row_grid(int M) {
  static int cap = 0;
  if (cap == 0) {
    const char* e = getenv("NVIT_ROW_GRID");   // experiments
    cap = e ? atoi(e) : 2048;
  }
  int blocks = cdiv(M, ROW_WAVES);
  return blocks > cap ? cap : blocks;
}

}  // namespace

#define DISPATCH_NV(C, ...)                     \
  do {                                          \
    if ((C) <= 256) { constexpr int NV = 1; __VA_ARGS__; }       \
    else if ((C) <= 512) { constexpr int NV = 2; __VA_ARGS__; }  \
    else if ((C) <= 768) { constexpr int NV = 3; __VA_ARGS__; }  \
    else if ((C) <= 1024) { constexpr int NV = 4; __VA_ARGS__; } \
    else { constexpr int NV = 8; __VA_ARGS__; }                  \
  } while (0)

extern "C" int nvit_lerp_fwd(int dt, const float* h, const void* y, int y_dt, const float* alpha, float c_a,
                             const float* skip_x, const float* skip, float* out, void* out_lo, int M, int C,
                             void* stream) {
  NVIT_REQUIRE(C % 4 == 0 && C <= 2048 && M > 0, "lerp_fwd: C=%d must be a multiple of 4 and <= 2048", C);
  NVIT_REQUIRE(!skip_x || skip, "lerp_fwd: skip_x without skip");
  LerpFwdArgs a{h, y, alpha, c_a, skip_x, skip, out, out_lo, M, C};
  hipStream_t s = (hipStream_t)stream;
  const int grid = row_grid(M);
  ProfScope ps(NVIT_KID_ROWOPS, 0.0, (double)M * C * (skip_x ? 16.0 : 12.0), s);
  DISPATCH_NV(C, {
    if (y_dt == NVIT_F32 && dt == NVIT_F32)
      hipLaunchKernelGGL((lerp_fwd_kernel<NV, float, float>), dim3(grid), dim3(256), 0, s, a);
    else if (y_dt == NVIT_F32)
      hipLaunchKernelGGL((lerp_fwd_kernel<NV, float, bf16>), dim3(grid), dim3(256), 0, s, a);
    else if (dt == NVIT_F32)
      hipLaunchKernelGGL((lerp_fwd_kernel<NV, bf16, float>), dim3(grid), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((lerp_fwd_kernel<NV, bf16, bf16>), dim3(grid), dim3(256), 0, s, a);
  });
  NVIT_CHECK_LAUNCH("lerp_fwd");
  return NVIT_OK;
}

// Workgroups that are resident at once for the lerp_bwd instantiation a call with these options would launch (CUs x
// blocks per CU from the occupancy query).  The kernel walks its rows with a grid stride, so a grid of exactly this many
// blocks keeps every SIMD busy to the end; the round-2 default of 1024 blocks ran as 768 + 256 at C = 768 (3 waves per
// SIMD): a second, one-third-full round (row kernels 14.9 -> 13.1 ms per Base step with the resident count).
template <typename K>
static int resident_blocks(K kernel) {
  int dev = 0, per_cu = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) return 0;
  return per_cu * prop.multiProcessorCount;
}

extern "C" int nvit_lerp_bwd_blocks(int dt, int y_dt, int C, int has_add, int has_skip, int accum) {
  if (C % 4 != 0 || C <= 0 || C > 2048) return 0;
  int n = 0;
#define NVIT_LB_Q3(TY_, TL_, ADD_)                                                                   \
  {                                                                                                  \
    if (has_skip && accum) n = resident_blocks(lerp_bwd_kernel<NV, TY_, TL_, ADD_, true, true>);     \
    else if (has_skip) n = resident_blocks(lerp_bwd_kernel<NV, TY_, TL_, ADD_, true, false>);        \
    else if (accum) n = resident_blocks(lerp_bwd_kernel<NV, TY_, TL_, ADD_, false, true>);           \
    else n = resident_blocks(lerp_bwd_kernel<NV, TY_, TL_, ADD_, false, false>);                     \
  }
#define NVIT_LB_Q(TY_, TL_)                \
  {                                        \
    if (has_add) NVIT_LB_Q3(TY_, TL_, true) \
    else NVIT_LB_Q3(TY_, TL_, false)       \
  }
  DISPATCH_NV(C, {
    if (y_dt == NVIT_F32 && dt == NVIT_F32) NVIT_LB_Q(float, float)
    else if (y_dt == NVIT_F32) NVIT_LB_Q(float, bf16)
    else if (dt == NVIT_F32) NVIT_LB_Q(bf16, float)
    else NVIT_LB_Q(bf16, bf16)
  });
#undef NVIT_LB_Q
#undef NVIT_LB_Q3
  return n > 4096 ? 4096 : n;
}

extern "C" int nvit_lerp_bwd(int dt, const float* dout, const void* dout_add, const float* h, const void* y, int y_dt, const float* alpha,
                             float c_a, const float* skip_x, const float* skip, float* dh, int accum_dh, float* dy,
                             void* dy_lo, float* dskip_x, float* part_dlam, float* part_dskip, int nblk, int M,
                             int C, void* stream) {
  NVIT_REQUIRE(C % 4 == 0 && C <= 2048 && M > 0, "lerp_bwd: C=%d must be a multiple of 4 and <= 2048", C);
  NVIT_REQUIRE(nblk > 0 && nblk <= 4096, "lerp_bwd: nblk out of range");
  NVIT_REQUIRE(!skip_x || (skip && dskip_x && part_dskip), "lerp_bwd: skip buffers missing");
  LerpBwdArgs a{dout, (const bf16*)dout_add, h, y, alpha, c_a, skip_x, skip, dh, accum_dh, dy, dy_lo, dskip_x, part_dlam, part_dskip, M, C};
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: dout, h read, y read (2 or 4 B), dh written, dy_lo written (+ skip_x read, dskip_x written; + dout_add)
  const double lb_bytes = (double)M * C * (4.0 + 4.0 + (y_dt == NVIT_F32 ? 4.0 : 2.0) + 4.0 + (dy ? 4.0 : 0.0) + (dy_lo ? (dt == NVIT_F32 ? 4.0 : 2.0) : 0.0) +
                                           (skip_x ? 8.0 : 0.0) + (accum_dh ? 4.0 : 0.0) + (dout_add ? 2.0 : 0.0));
  ProfScope ps(NVIT_KID_ROWOPS, 0.0, lb_bytes, s);
#define NVIT_LERP_BWD_LAUNCH3(TY_, TL_, ADD_)                                                                          \
  {                                                                                                                    \
    if (skip_x && accum_dh)                                                                                            \
      hipLaunchKernelGGL((lerp_bwd_kernel<NV, TY_, TL_, ADD_, true, true>), dim3(nblk), dim3(256), 0, s, a);           \
    else if (skip_x)                                                                                                   \
      hipLaunchKernelGGL((lerp_bwd_kernel<NV, TY_, TL_, ADD_, true, false>), dim3(nblk), dim3(256), 0, s, a);          \
    else if (accum_dh)                                                                                                 \
      hipLaunchKernelGGL((lerp_bwd_kernel<NV, TY_, TL_, ADD_, false, true>), dim3(nblk), dim3(256), 0, s, a);          \
    else                                                                                                               \
      hipLaunchKernelGGL((lerp_bwd_kernel<NV, TY_, TL_, ADD_, false, false>), dim3(nblk), dim3(256), 0, s, a);         \
  }
#define NVIT_LERP_BWD_LAUNCH(TY_, TL_)          \
  {                                             \
    if (dout_add)                               \
      NVIT_LERP_BWD_LAUNCH3(TY_, TL_, true)     \
    else                                        \
      NVIT_LERP_BWD_LAUNCH3(TY_, TL_, false)    \
  }
  DISPATCH_NV(C, {
    if (y_dt == NVIT_F32 && dt == NVIT_F32)
      NVIT_LERP_BWD_LAUNCH(float, float)
    else if (y_dt == NVIT_F32)
      NVIT_LERP_BWD_LAUNCH(float, bf16)
    else if (dt == NVIT_F32)
      NVIT_LERP_BWD_LAUNCH(bf16, float)
    else
      NVIT_LERP_BWD_LAUNCH(bf16, bf16)
  });
#undef NVIT_LERP_BWD_LAUNCH
#undef NVIT_LERP_BWD_LAUNCH3
  NVIT_CHECK_LAUNCH("lerp_bwd");
  return NVIT_OK;
}

extern "C" int nvit_norm_skip_fwd(const float* src, const float* tgt, const float* skip, float* out, int M, int C,
                                  void* stream) {
  NVIT_REQUIRE(C % 4 == 0 && C <= 2048 && M > 0, "norm_skip_fwd: C=%d must be a multiple of 4 and <= 2048", C);
  hipStream_t s = (hipStream_t)stream;
  const int grid = row_grid(M);
  DISPATCH_NV(C, { hipLaunchKernelGGL((norm_skip_fwd_kernel<NV>), dim3(grid), dim3(256), 0, s, src, tgt, skip, out, M, C); });
  NVIT_CHECK_LAUNCH("norm_skip_fwd");
  return NVIT_OK;
}

extern "C" int nvit_norm_skip_bwd(const float* dout, const float* src, const float* tgt, const float* skip, float* dsrc,
                                  float* dtgt, float* part_dskip, int nblk, int M, int C, void* stream) {
  NVIT_REQUIRE(C % 4 == 0 && C <= 2048 && M > 0 && nblk > 0 && nblk <= 4096, "norm_skip_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_NV(C, {
    hipLaunchKernelGGL((norm_skip_bwd_kernel<NV>), dim3(nblk), dim3(256), 0, s, dout, src, tgt, skip, dsrc, dtgt,
                       part_dskip, M, C);
  });
  NVIT_CHECK_LAUNCH("norm_skip_bwd");
  return NVIT_OK;
}

extern "C" int nvit_rmsnorm_fwd(const float* x, const float* w, float eps, float* out, float* rstd, int M, int C,
                                void* stream) {
  NVIT_REQUIRE(C % 4 == 0 && C <= 2048 && M > 0, "rmsnorm_fwd: C=%d must be a multiple of 4 and <= 2048", C);
  hipStream_t s = (hipStream_t)stream;
  const int grid = row_grid(M);
  DISPATCH_NV(C, { hipLaunchKernelGGL((rmsnorm_fwd_kernel<NV>), dim3(grid), dim3(256), 0, s, x, w, eps, out, rstd, M, C); });
  NVIT_CHECK_LAUNCH("rmsnorm_fwd");
  return NVIT_OK;
}

extern "C" int nvit_rmsnorm_bwd(const float* dout, const float* x, const float* w, const float* rstd, float* dx,
                                float* part_dw, int nblk, int M, int C, void* stream) {
  NVIT_REQUIRE(C % 4 == 0 && C <= 2048 && M > 0 && nblk > 0 && nblk <= 4096, "rmsnorm_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_NV(C, {
    hipLaunchKernelGGL((rmsnorm_bwd_kernel<NV>), dim3(nblk), dim3(256), 0, s, dout, x, w, rstd, dx, part_dw, M, C);
  });
  NVIT_CHECK_LAUNCH("rmsnorm_bwd");
  return NVIT_OK;
}

extern "C" int nvit_qknorm_fwd(int dt, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                               const float* sqk, float c_q, void* qh, void* kh, void* vh, float* rq, float* rk,
                               int B, int T, int H, int d, void* stream) {
  const int C = H * d;
  NVIT_REQUIRE(d == 16 || d == 32 || d == 64 || d == 128, "qknorm_fwd: head dim %d unsupported", d);
  NVIT_REQUIRE(C <= 2048 && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0, "qknorm_fwd: bad C/ld");
  QkArgs a{q, k, v, ldq, ldk, ldv, sqk, c_q, qh, kh, vh, rq, rk, B, T, H, d};
  hipStream_t s = (hipStream_t)stream;
  const int grid = row_grid(B * T);
  NVIT_REQUIRE(dt == NVIT_F32 || dt == NVIT_BF16 || dt == NVIT_BF16_F32IN, "qknorm_fwd: bad dt %d", dt);
  ProfScope ps(NVIT_KID_ROWOPS, 0.0, (double)B * T * C * (dt == NVIT_F32 ? 24.0 : dt == NVIT_BF16 ? 12.0 : 18.0), s);
  DISPATCH_NV(C, {
    if (dt == NVIT_F32)
      hipLaunchKernelGGL((qknorm_fwd_kernel<NV, float, float>), dim3(grid), dim3(256), 0, s, a);
    else if (dt == NVIT_BF16)
      hipLaunchKernelGGL((qknorm_fwd_kernel<NV, bf16, bf16>), dim3(grid), dim3(256), 0, s, a);
    else   // fp32 projection outputs -> bf16 head tensors: normalised from the unrounded values
      hipLaunchKernelGGL((qknorm_fwd_kernel<NV, float, bf16>), dim3(grid), dim3(256), 0, s, a);
  });
  NVIT_CHECK_LAUNCH("qknorm_fwd");
  return NVIT_OK;
}

extern "C" int nvit_qknorm_bwd(int dt, const void* dqh, const void* dkh, const void* dvh, const void* qh,
                               const void* kh, const float* rq, const float* rk, const float* sqk, float c_q,
                               void* dq, int ldq, void* dk, int ldk, void* dv, int ldv, float* part_dsqk, int nblk,
                               int B, int T, int H, int d, void* stream) {
  const int C = H * d;
  NVIT_REQUIRE(d == 16 || d == 32 || d == 64 || d == 128, "qknorm_bwd: head dim %d unsupported", d);
  NVIT_REQUIRE(C <= 2048 && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0, "qknorm_bwd: bad C/ld");
  NVIT_REQUIRE(nblk > 0 && nblk <= 4096, "qknorm_bwd: nblk out of range");
  QkBwdArgs a{dqh, dkh, dvh, qh, kh, rq, rk, sqk, c_q, dq, dk, dv, ldq, ldk, ldv, part_dsqk, B, T, H, d};
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_ROWOPS, 0.0, (double)B * T * C * (dt == NVIT_F32 ? 32.0 : 16.0), s);
  DISPATCH_NV(C, {
    if (dt == NVIT_F32)
      hipLaunchKernelGGL((qknorm_bwd_kernel<NV, float>), dim3(nblk), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((qknorm_bwd_kernel<NV, bf16>), dim3(nblk), dim3(256), 0, s, a);
  });
  NVIT_CHECK_LAUNCH("qknorm_bwd");
  return NVIT_OK;
}

static int swiglu_rows_per_blk(int M, int F) {
  const int colblocks = cdiv(F, 1024);
  int rowblocks = cdiv(2048, colblocks);
  int rpb = cdiv(M, rowblocks);
  return rpb < 1 ? 1 : rpb;
}

extern "C" int nvit_swiglu_fwd(int dt, const void* uv, const float* suv, float gscale, void* x, int M, int F,
                               void* stream) {
  NVIT_REQUIRE(F % 16 == 0 && M > 0, "swiglu_fwd: F=%d must be a multiple of 16", F);
  hipStream_t s = (hipStream_t)stream;
  const int rpb = swiglu_rows_per_blk(M, F);
  dim3 grid(cdiv(F, 1024), cdiv(M, rpb));
  NVIT_REQUIRE(dt == NVIT_F32 || dt == NVIT_BF16 || dt == NVIT_BF16_F32IN, "swiglu_fwd: bad dt %d", dt);
  ProfScope ps(NVIT_KID_ROWOPS, 0.0, (double)M * F * (dt == NVIT_F32 ? 12.0 : dt == NVIT_BF16 ? 6.0 : 10.0), s);
  if (dt == NVIT_F32)
    hipLaunchKernelGGL((swiglu_fwd_kernel<float, float>), grid, dim3(256), 0, s, (const float*)uv, suv, gscale, (float*)x, M, F, rpb);
  else if (dt == NVIT_BF16)
    hipLaunchKernelGGL((swiglu_fwd_kernel<bf16, bf16>), grid, dim3(256), 0, s, (const bf16*)uv, suv, gscale, (bf16*)x, M, F, rpb);
  else   // fp32 pre-activations -> bf16 gated output: gated from the unrounded values (like the fused GEMM epilogue)
    hipLaunchKernelGGL((swiglu_fwd_kernel<float, bf16>), grid, dim3(256), 0, s, (const float*)uv, suv, gscale, (bf16*)x, M, F, rpb);
  NVIT_CHECK_LAUNCH("swiglu_fwd");
  return NVIT_OK;
}

extern "C" int nvit_swiglu_bwd(int dt, const void* dx, const void* uv, const float* suv, float gscale, void* duv,
                               float* part_dsuv, int rows_per_blk, int M, int F, void* stream) {
  NVIT_REQUIRE(F % 16 == 0 && M > 0 && rows_per_blk > 0, "swiglu_bwd: bad shape");
  NVIT_REQUIRE(!suv || part_dsuv, "swiglu_bwd: part_dsuv missing");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(F, 1024), cdiv(M, rows_per_blk));
  ProfScope ps(NVIT_KID_ROWOPS, 0.0, (double)M * F * 5.0 * (dt == NVIT_F32 ? 4 : 2), s);
  if (dt == NVIT_F32)
    hipLaunchKernelGGL(swiglu_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)dx, (const float*)uv, suv, gscale,
                       (float*)duv, suv ? part_dsuv : nullptr, M, F, rows_per_blk);
  else
    hipLaunchKernelGGL(swiglu_bwd_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)dx, (const bf16*)uv, suv, gscale,
                       (bf16*)duv, suv ? part_dsuv : nullptr, M, F, rows_per_blk);
  NVIT_CHECK_LAUNCH("swiglu_bwd");
  return NVIT_OK;
}

extern "C" int nvit_colsum_reduce(const float* part, int nblk, int N, float* out, int accumulate, int kind,
                                  const float* ref, float scale, void* stream) {
  NVIT_REQUIRE(kind == 0 || (kind == 1 && ref) || (kind == 2 && N % 32 == 0), "colsum_reduce: bad kind");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(N, 32)), dim3(1024), 0, s, part, nblk, N, out, accumulate, kind,
                     ref, scale);
  NVIT_CHECK_LAUNCH("colsum_reduce");
  return NVIT_OK;
}

// n <= 8 reductions in one launch.  Host arrays of n entries each; part_b[i] may be 0 (no second partial array).
extern "C" int nvit_colsum_reduce_multi(const int64_t* part, const int* nblk, const int64_t* part_b, const int* nblk_b,
                                        const int* N, const int64_t* out, const int* accumulate, const int* kind,
                                        const int64_t* ref, const float* scale, int n, void* stream) {
  NVIT_REQUIRE(n >= 1 && n <= CSR_MAX_ITEMS && part && nblk && part_b && nblk_b && N && out && accumulate && kind && ref && scale,
               "colsum_reduce_multi: 1..%d items", CSR_MAX_ITEMS);
  CsrBatch bt{};
  bt.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    NVIT_REQUIRE(part[i] && out[i] && nblk[i] > 0 && N[i] > 0, "colsum_reduce_multi: item %d is empty", i);
    NVIT_REQUIRE(kind[i] == 0 || (kind[i] == 1 && ref[i]) || (kind[i] == 2 && N[i] % 32 == 0), "colsum_reduce_multi: bad kind");
    NVIT_REQUIRE(!part_b[i] || nblk_b[i] > 0, "colsum_reduce_multi: item %d: second partial array without rows", i);
    CsrItem& it = bt.it[i];
    it.part = reinterpret_cast<const float*>(part[i]);
    it.part_b = reinterpret_cast<const float*>(part_b[i]);
    it.ref = reinterpret_cast<const float*>(ref[i]);
    it.out = reinterpret_cast<float*>(out[i]);
    it.nblk = nblk[i];
    it.nblk_b = nblk_b[i];
    it.N = N[i];
    it.accumulate = accumulate[i];
    it.kind = kind[i];
    it.scale = scale[i];
    it.block0 = blocks;
    blocks += cdiv(N[i], 32);
  }
  hipLaunchKernelGGL(colsum_reduce_multi_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, bt);
  NVIT_CHECK_LAUNCH("colsum_reduce_multi");
  return NVIT_OK;
}

extern "C" int nvit_colsum(const void* a, int a_dt, int lda, const void* b, int b_dt, int ldb, int R, int N,
                           int period, float* out, int accumulate, float scale, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int per = period > 0 ? period : 1;
  dim3 grid(cdiv(N, 128), per);
#define CS(TA, TB) \
  hipLaunchKernelGGL((colsum_kernel<TA, TB>), grid, dim3(128), 0, s, (const TA*)a, lda, (const TB*)b, ldb, R, N, per, out, accumulate, scale)
  if (a_dt == NVIT_F32 && (b_dt == NVIT_F32 || !b)) CS(float, float);
  else if (a_dt == NVIT_F32) CS(float, bf16);
  else if (b_dt == NVIT_F32 || !b) CS(bf16, float);
  else CS(bf16, bf16);
#undef CS
  NVIT_CHECK_LAUNCH("colsum");
  return NVIT_OK;
}

extern "C" int nvit_cast(const float* src, void* dst, int dt, int64_t n, void* stream) {
  NVIT_REQUIRE(n % 4 == 0, "cast: n must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  const long long n4 = n / 4;
  int blocks = cdiv(n4, 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) return NVIT_OK;
  if (dt == NVIT_F32)
    hipLaunchKernelGGL(cast_kernel<float>, dim3(blocks), dim3(256), 0, s, src, (float*)dst, n4);
  else
    hipLaunchKernelGGL(cast_kernel<bf16>, dim3(blocks), dim3(256), 0, s, src, (bf16*)dst, n4);
  NVIT_CHECK_LAUNCH("cast");
  return NVIT_OK;
}

extern "C" int nvit_normalize_images(const void* in, int in_is_u8_hwc, float* out, int B, int C, int H, int W, float mean,
                                     float std_, void* stream) {
  NVIT_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0 && W % 4 == 0 && std_ != 0.f,
               "normalize_images: bad arguments (W must be a multiple of 4)");
  NVIT_REQUIRE(((uintptr_t)out & 15) == 0 && (in_is_u8_hwc || ((uintptr_t)in & 15) == 0),
               "normalize_images: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const long long n4 = (long long)B * C * H * (W / 4);
  int blocks = cdiv(n4, 256);
  if (blocks > 8192) blocks = 8192;
  ProfScope ps(NVIT_KID_PATCHIFY, 0.0, (double)n4 * 4.0 * (in_is_u8_hwc ? 5.0 : 8.0), s);
  if (in_is_u8_hwc)
    hipLaunchKernelGGL(normalize_images_kernel<unsigned char>, dim3(blocks), dim3(256), 0, s, (const unsigned char*)in, out,
                       B, C, H, W, mean, 1.0f / std_);
  else
    hipLaunchKernelGGL(normalize_images_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)in, out, B, C, H, W,
                       mean, 1.0f / std_);
  NVIT_CHECK_LAUNCH("normalize_images");
  return NVIT_OK;
}

extern "C" int nvit_scale_cols(const float* a, int lda, const float* sc, float c, void* out, int out_dt, int ldo,
                               int R, int N, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(N, 128), R);
  if (out_dt == NVIT_F32)
    hipLaunchKernelGGL(scale_cols_kernel<float>, grid, dim3(128), 0, s, a, lda, sc, c, (float*)out, ldo, R, N);
  else
    hipLaunchKernelGGL(scale_cols_kernel<bf16>, grid, dim3(128), 0, s, a, lda, sc, c, (bf16*)out, ldo, R, N);
  NVIT_CHECK_LAUNCH("scale_cols");
  return NVIT_OK;
}
