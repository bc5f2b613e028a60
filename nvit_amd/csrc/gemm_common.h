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
