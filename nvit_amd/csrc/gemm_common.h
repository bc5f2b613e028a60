// Pieces shared by the NT GEMM kernels (gemm.hip: 128x128 tile; gemm_p.hip: persistent 256x128 tile).
#pragma once
#include "common.h"

constexpr int ROWB = 128;  // bytes of K per LDS row per stage (64 bf16 / 32 fp32)

struct NtArgs {
  const char* A;
  const char* B;
  void* C;
  int M, N, K;
  int lda, ldb, ldc;  // elements
  const float* bias;
  const float* colscale;
  const float* rowadd;
  int rowadd_period;
  int accumulate;
  int out_dt;
  int tiles_n;
  // --- fused SwiGLU epilogue (EPI 3): C = raw uv (interleaved u16|v16 columns), xm = (gu*u)*silu(gv*v)
  void* xm;
  int ld_xm;
  const float* gs;  // suv in the interleaved column order of C (or NULL = ones)
  float gscale;
  // --- fused q/k-normalise epilogue (EPI 4): columns are nparts stacked [C]-wide projections
  void *qh, *kh, *vh;  // [B,H,T,64] bf16
  float *rq, *rk;      // [M,H] 1/||.||
  const float* sqk;
  float c_q;
  int part0, Cemb, Ttok, H;
};

template <typename T>
struct Mma;
template <>
struct Mma<bf16> {
  // one 16B chunk = 8 bf16 of K -> one 16x16x32 MFMA
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                  acc, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  // one 16B chunk = 4 fp32 of K -> four 16x16x4 MFMAs (k slot = lane>>4, any consistent
  // assignment of k to slots is valid because A and B use the same one)
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& acc) {
    f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t], fb[t], acc, 0, 0, 0);
  }
};

// LDS-DMA of 16 bytes per lane: LDS destination = lds_off (wave-uniform, in M0) + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_off);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(m)
      : "memory");
}


// Epilogue of one wave's (16*FMR)x64 sub-tile whose top-left element is C[m_base][n_base]:
// acc[i][j][r] = C[m_base + 16i + l15][n_base + 16j + 4*lg + r]; +bias, *colscale, +rowadd, +old C; store.
template <int FMR>
__device__ __forceinline__ void nt_store_tile(const NtArgs& g, f32x4 (&acc)[FMR][4], int m_base, int n_base, int l15,
                                              int lg) {
  const bool vec_ok = ((g.N & 3) == 0) && ((g.ldc & 3) == 0);
#pragma unroll
  for (int i = 0; i < FMR; ++i) {
    const int m = m_base + i * 16 + l15;
    if (m >= g.M) continue;
    const float* radd = g.rowadd ? g.rowadd + (size_t)(m % g.rowadd_period) * g.N : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nb = n_base + j * 16 + 4 * lg;
      if (nb >= g.N) continue;
      f32x4 v = acc[i][j];
      if (vec_ok) {
        if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + nb);
        if (g.colscale) v *= *reinterpret_cast<const f32x4*>(g.colscale + nb);
        if (radd) v += *reinterpret_cast<const f32x4*>(radd + nb);
        if (g.out_dt == NVIT_F32) {
          float* cp = reinterpret_cast<float*>(g.C) + (size_t)m * g.ldc + nb;
          if (g.accumulate) v += *reinterpret_cast<const f32x4*>(cp);
          *reinterpret_cast<f32x4*>(cp) = v;
        } else {
          bf16* cp = reinterpret_cast<bf16*>(g.C) + (size_t)m * g.ldc + nb;
          if (g.accumulate) v += load4<bf16>(cp);
          store4<bf16>(cp, v);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = nb + r;
          if (n >= g.N) break;
          float x = v[r];
          if (g.bias) x += g.bias[n];
          if (g.colscale) x *= g.colscale[n];
          if (radd) x += radd[n];
          if (g.out_dt == NVIT_F32) {
            float* cp = reinterpret_cast<float*>(g.C) + (size_t)m * g.ldc + n;
            if (g.accumulate) x += *cp;
            *cp = x;
          } else {
            bf16* cp = reinterpret_cast<bf16*>(g.C) + (size_t)m * g.ldc + n;
            if (g.accumulate) x += (float)*cp;
            *cp = (bf16)x;
          }
        }
      }
    }
  }
}

// LDS-staged epilogue (persistent kernels): the wave re-shapes each 16-row slab of its sub-tile
// through a private 2 KiB LDS scratch so that every global store instruction writes whole 128-byte
// row segments (16 B per lane, 8 lanes per row) instead of 16 scattered 32-byte pieces.
// One pass = 16 rows x 128 B of OUTPUT (64 bf16 or 32 fp32 columns); 16-byte chunks are XOR-swizzled
// by (row & 7) inside the scratch.  Requires N and ldc to be multiples of 16 B / sizeof(out).
template <int FMR, typename TO>
__device__ __forceinline__ void nt_store_tile_staged(const NtArgs& g, f32x4 (&acc)[FMR][4], int m_base, int n_base,
                                                     int lane, char* scratch) {
  constexpr int EO = sizeof(TO);          // output element bytes
  constexpr int CPP = 128 / EO;           // columns per pass: 64 (bf16) / 32 (fp32)
  constexpr int JPP = CPP / 16;           // accumulator column blocks per pass: 4 / 2
  constexpr int PASSES = 4 / JPP;         // 1 / 2
  constexpr int EPC_O = 16 / EO;          // output elements per 16-byte chunk
  const int l15 = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < FMR; ++i) {
    const int mfrag = m_base + i * 16 + l15;
    const float* radd =
        g.rowadd ? g.rowadd + (size_t)((mfrag < g.M ? mfrag : g.M - 1) % g.rowadd_period) * g.N : nullptr;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      // fragment layout -> scratch[row = l15][cols 16*jj + 4*lg .. +3]
#pragma unroll
      for (int jj = 0; jj < JPP; ++jj) {
        const int j = p * JPP + jj;
        const int nb = n_base + j * 16 + 4 * lg;
        f32x4 v = acc[i][j];
        if (nb < g.N) {
          if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + nb);
          if (g.colscale) v *= *reinterpret_cast<const f32x4*>(g.colscale + nb);
          if (radd) v += *reinterpret_cast<const f32x4*>(radd + nb);
        }
        const int bytecol = (jj * 16 + 4 * lg) * EO;
        const int chunk = bytecol >> 4;
        char* dst = scratch + l15 * 128 + ((chunk ^ (l15 & 7)) << 4) + (bytecol & 15);
        store4<TO>(reinterpret_cast<TO*>(dst), v);
      }
      // scratch -> global: 128 chunks of 16 B, two per lane; row = idx >> 3, chunk = idx & 7
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int idx = lane + 64 * t;
        const int row = idx >> 3, chunk = idx & 7;
        const uint4 raw = *reinterpret_cast<const uint4*>(scratch + row * 128 + ((chunk ^ (row & 7)) << 4));
        const int m = m_base + i * 16 + row;
        const int n = n_base + p * CPP + chunk * EPC_O;
        if (m < g.M && n < g.N) {
          TO* cp = reinterpret_cast<TO*>(g.C) + (size_t)m * g.ldc + n;
          if (g.accumulate) {
            if constexpr (EO == 4) {
              f32x4 v = __builtin_bit_cast(f32x4, raw);
              v += *reinterpret_cast<const f32x4*>(cp);
              *reinterpret_cast<f32x4*>(cp) = v;
            } else {
              const bf16x8 nv = __builtin_bit_cast(bf16x8, raw);
              const bf16x8 ov = *reinterpret_cast<const bf16x8*>(cp);
              bf16x8 r;
#pragma unroll
              for (int e = 0; e < 8; ++e) r[e] = (bf16)((float)nv[e] + (float)ov[e]);
              *reinterpret_cast<bf16x8*>(cp) = r;
            }
          } else {
            *reinterpret_cast<uint4*>(cp) = raw;
          }
        }
      }
    }
  }
}

// ---- EPI 3: SwiGLU fused into the c_fc / proj GEMM (reference model.py:148-154, 259-261) -----------
// The weight shadow interleaves u/v partners 16 columns apart (perm=1), so acc[i][2jj] (u) and
// acc[i][2jj+1] (v) of one lane are gate partners.  Writes the raw pre-activation tile (saved for
// backward) and the gated activation, both bf16, both as whole-row 16-byte stores via the LDS scratch.
template <int FMR>
__device__ __forceinline__ void nt_store_tile_swiglu(const NtArgs& g, f32x4 (&acc)[FMR][4], int m_base, int n_base,
                                                     int lane, char* scratch) {
  const int l15 = lane & 15, lg = lane >> 4;
  f32x4 gu[2], gv[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int nb = n_base + jj * 32 + 4 * lg;
    if (g.gs) {
      gu[jj] = *reinterpret_cast<const f32x4*>(g.gs + nb) * g.gscale;
      gv[jj] = *reinterpret_cast<const f32x4*>(g.gs + nb + 16) * g.gscale;
    } else {
      gu[jj] = (f32x4){1.f, 1.f, 1.f, 1.f};
      gv[jj] = gu[jj];
    }
  }
#pragma unroll
  for (int i = 0; i < FMR; ++i) {
    // pass A: raw uv, 64 columns
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int bytecol = (j * 16 + 4 * lg) * 2;
      char* dst = scratch + l15 * 128 + (((bytecol >> 4) ^ (l15 & 7)) << 4) + (bytecol & 15);
      store4<bf16>(reinterpret_cast<bf16*>(dst), acc[i][j]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int idx = lane + 64 * t;
      const int row = idx >> 3, chunk = idx & 7;
      const uint4 raw = *reinterpret_cast<const uint4*>(scratch + row * 128 + ((chunk ^ (row & 7)) << 4));
      const int m = m_base + i * 16 + row;
      if (m < g.M)
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(g.C) + (size_t)m * g.ldc + n_base + chunk * 8) = raw;
    }
    // pass B: gated activation, 32 columns
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const f32x4 u = acc[i][2 * jj] * gu[jj], v = acc[i][2 * jj + 1] * gv[jj];
      f32x4 x;
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = u[e] * (v[e] / (1.0f + __expf(-v[e])));
      const int bytecol = (jj * 16 + 4 * lg) * 2;
      char* dst = scratch + l15 * 128 + (((bytecol >> 4) ^ (l15 & 7)) << 4) + (bytecol & 15);
      store4<bf16>(reinterpret_cast<bf16*>(dst), x);
    }
    {
      const int row = lane >> 2, chunk = lane & 3;
      const uint4 raw = *reinterpret_cast<const uint4*>(scratch + row * 128 + ((chunk ^ (row & 7)) << 4));
      const int m = m_base + i * 16 + row;
      if (m < g.M)
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(g.xm) + (size_t)m * g.ld_xm + (n_base >> 1) + chunk * 8) = raw;
    }
  }
}

// ---- EPI 4: per-head cosine normalise + learned scale + head split fused into the q/k/v GEMM --------
// (reference model.py:104-119, 231-247).  A wave's 64 columns are exactly one head (d = 64).
template <int FMR>
__device__ __forceinline__ void nt_store_tile_qknorm(const NtArgs& g, f32x4 (&acc)[FMR][4], int m_base, int n_base,
                                                     int lane, char* scratch) {
  const int l15 = lane & 15, lg = lane >> 4;
  const int part = g.part0 + n_base / g.Cemb;  // 0 = q, 1 = k, 2 = v
  const int c0 = n_base % g.Cemb, h = c0 >> 6;
  f32x4 sc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) sc[j] = *reinterpret_cast<const f32x4*>(g.sqk + c0 + j * 16 + 4 * lg) * g.c_q;
  bf16* outp = reinterpret_cast<bf16*>(part == 0 ? g.qh : (part == 1 ? g.kh : g.vh));
  float* rn_out = part == 0 ? g.rq : g.rk;
#pragma unroll
  for (int i = 0; i < FMR; ++i) {
    float rn = 1.0f;
    if (part < 2) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        ss += acc[i][j][0] * acc[i][j][0] + acc[i][j][1] * acc[i][j][1] + acc[i][j][2] * acc[i][j][2] +
              acc[i][j][3] * acc[i][j][3];
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      rn = 1.0f / sqrtf(ss);
      const int m = m_base + i * 16 + l15;
      if (lg == 0 && m < g.M) rn_out[(size_t)m * g.H + h] = rn;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v = acc[i][j];
      if (part < 2) v = v * rn * sc[j];
      const int bytecol = (j * 16 + 4 * lg) * 2;
      char* dst = scratch + l15 * 128 + (((bytecol >> 4) ^ (l15 & 7)) << 4) + (bytecol & 15);
      store4<bf16>(reinterpret_cast<bf16*>(dst), v);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int idx = lane + 64 * t;
      const int row = idx >> 3, chunk = idx & 7;
      const uint4 raw = *reinterpret_cast<const uint4*>(scratch + row * 128 + ((chunk ^ (row & 7)) << 4));
      const int m = m_base + i * 16 + row;
      if (m < g.M) {
        const int b = m / g.Ttok, tt = m - b * g.Ttok;
        *reinterpret_cast<uint4*>(outp + (((size_t)b * g.H + h) * g.Ttok + tt) * 64 + chunk * 8) = raw;
      }
    }
  }
}
