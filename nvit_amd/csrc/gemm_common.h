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
  float q_prescale;   // extra factor folded into the q part's scale (attention exponent pre-scale), 1 = none
  int part0, Cemb, Ttok, H;
  // --- fused SwiGLU-backward epilogue (EPI 5): acc = dx [M,Fh]; C = duv (interleaved, ldc = 2*Fh)
  const void* uv_in;  // raw pre-activations saved by EPI 3 (interleaved u16|v16 columns), bf16
  int ld_uv;
  int Fh;             // hidden width F; gs = suv in natural order [u(F) | v(F)] (or NULL = ones)
  float* part;        // [2*tiles_m, 2*Fh] column partials of d(suv) per 128-row wave tile (or NULL)
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

// The same with the source split into a wave-uniform 64-bit base (scalar registers) and a 32-bit per-lane byte offset: the
// per-stage part of the address (which k-slab, which tile) is scalar arithmetic, not two vector adds per piece, and a
// lane's address costs one register instead of two.
__device__ __forceinline__ void glds16s(unsigned long long sbase, unsigned voff, unsigned lds_off) {
  unsigned keep;
  const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_off);
  const unsigned long long sb = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sbase) |
                                ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sbase >> 32)) << 32);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sb), "s"(m)
      : "memory");
}

// N (2 or 4) such pieces of one operand in ONE statement - LDS destinations lds_off + i * 8 KiB, lane offsets voff[i] from
// the same scalar base: M0 is saved and restored once per operand instead of once per piece, and the destinations are
// formed by scalar adds into M0 itself (the per-piece form costs ~8 scalar instructions per piece; an in-order wave
// pays for every one of them beside its MFMAs).
template <int N>
__device__ __forceinline__ void glds16s_n(unsigned long long sbase, const unsigned (&voff)[N], unsigned lds_off) {
  static_assert(N == 2 || N == 4, "2 or 4 pieces");
  unsigned keep;
  const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_off);
  const unsigned long long sb = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sbase) |
                                ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sbase >> 32)) << 32);
  if constexpr (N == 4) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
        "s_add_u32 m0, %6, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
        "s_add_u32 m0, %6, 0x4000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
        "s_add_u32 m0, %6, 0x6000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(sb), "s"(m)
        : "memory", "scc");
  } else {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
        "s_add_u32 m0, %4, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff[0]), "v"(voff[1]), "s"(sb), "s"(m)
        : "memory", "scc");
  }
}

// 16-byte non-temporal global store.  GEMM outputs are written once and read by a LATER kernel; storing them with
// the nt policy keeps them from evicting the A/B operand tiles that the next tiles of THIS kernel re-read from L2
// (measured on 256x256 tiles: -7 % (bf16 out) / -10 % (fp32 out) per tile at K = 1536, -1..3 % at K = 768).
__device__ __forceinline__ uint4 ld16_nt(const void* p) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_*>(p)));
}
__device__ __forceinline__ void st16_nt(void* p, const uint4& v) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(__builtin_bit_cast(u32x4_, v), reinterpret_cast<u32x4_*>(p));
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
  constexpr int NP = FMR * PASSES;        // passes of the wave's sub-tile (16 rows x 128 B of output each)
  constexpr int PD = EO == 4 ? 4 : 1;     // fp32 accumulate: passes whose old C values are in flight ahead of their use
  const int l15 = lane & 15, lg = lane >> 4;
  // fp32 "+=" (the data-gradient GEMMs that accumulate into the residual-stream gradient): the old C values are read
  // through a register queue PD passes ahead.  Loaded at their point of use, each of the 16 passes of a wave waited a
  // full memory round trip: 22 us of epilogue per tile instead of 9 (tools/probes/gemm_parts.hip with accumulate on).
  // (the main loop's fragment registers are dead here, so the queue is free)
  uint4 oldq[PD][2];
  auto old_ptr = [&](int pass, int t) -> const TO* {
    const int i = pass / PASSES, p = pass % PASSES;
    const int idx = lane + 64 * t;
    const int row = idx >> 3, chunk = idx & 7;
    int m = m_base + i * 16 + row, n = n_base + p * CPP + chunk * EPC_O;
    m = m < g.M ? m : g.M - 1;
    n = n < g.N ? n : 0;
    return reinterpret_cast<const TO*>(g.C) + (size_t)m * g.ldc + n;
  };
  const bool prefetch = EO == 4 && g.accumulate;
  if constexpr (EO == 4) {
    if (prefetch) {
#pragma unroll
      for (int q = 0; q < PD; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) oldq[q][t] = ld16_nt(old_ptr(q, t));
    }
  }
#pragma unroll
  for (int i = 0; i < FMR; ++i) {
    const int mfrag = m_base + i * 16 + l15;
    // (the row-periodic addend is a rare option - the exact-f32 patch embedding: the branch and the empty statement keep
    //  its per-lane modulo out of the persistent kernel's stage loop, where the compiler otherwise computes it
    //  speculatively every stage, ~17 VALU beside the MFMAs)
    const float* radd = nullptr;
    if (g.rowadd) {
      int rrow = (mfrag < g.M ? mfrag : g.M - 1) % g.rowadd_period;
      asm volatile("" : "+v"(rrow));
      radd = g.rowadd + (size_t)rrow * g.N;
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int pass = i * PASSES + p;
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
        uint4 oldv = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (EO == 4) {
          if (prefetch) {
            oldv = oldq[pass % PD][t];
            if (pass + PD < NP) oldq[pass % PD][t] = ld16_nt(old_ptr(pass + PD, t));
          }
        }
        if (m < g.M && n < g.N) {
          TO* cp = reinterpret_cast<TO*>(g.C) + (size_t)m * g.ldc + n;
          if (g.accumulate) {
            if constexpr (EO == 4) {
              f32x4 v = __builtin_bit_cast(f32x4, raw);
              v += __builtin_bit_cast(f32x4, oldv);
              st16_nt(cp, __builtin_bit_cast(uint4, v));
            } else {
              const bf16x8 nv = __builtin_bit_cast(bf16x8, raw);
              const bf16x8 ov = __builtin_bit_cast(bf16x8, ld16_nt(cp));
              bf16x8 r;
#pragma unroll
              for (int e = 0; e < 8; ++e) r[e] = (bf16)((float)nv[e] + (float)ov[e]);
              st16_nt(cp, __builtin_bit_cast(uint4, r));
            }
          } else {
            st16_nt(cp, raw);
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
        st16_nt(reinterpret_cast<bf16*>(g.C) + (size_t)m * g.ldc + n_base + chunk * 8, raw);
    }
    // pass B: gated activation, 32 columns
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const f32x4 u = acc[i][2 * jj] * gu[jj], v = acc[i][2 * jj + 1] * gv[jj];
      f32x4 x;
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = u[e] * (v[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[e])));   // v_rcp_f32 (1 ulp; the result is rounded to bf16): the IEEE division is 10 VALU instructions per element in a tile-blocking epilogue
      const int bytecol = (jj * 16 + 4 * lg) * 2;
      char* dst = scratch + l15 * 128 + (((bytecol >> 4) ^ (l15 & 7)) << 4) + (bytecol & 15);
      store4<bf16>(reinterpret_cast<bf16*>(dst), x);
    }
    {
      const int row = lane >> 2, chunk = lane & 3;
      const uint4 raw = *reinterpret_cast<const uint4*>(scratch + row * 128 + ((chunk ^ (row & 7)) << 4));
      const int m = m_base + i * 16 + row;
      if (m < g.M)
        st16_nt(reinterpret_cast<bf16*>(g.xm) + (size_t)m * g.ld_xm + (n_base >> 1) + chunk * 8, raw);
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
  const float cq = part == 0 ? g.c_q * g.q_prescale : g.c_q;
#pragma unroll
  for (int j = 0; j < 4; ++j) sc[j] = *reinterpret_cast<const f32x4*>(g.sqk + c0 + j * 16 + 4 * lg) * cq;
  bf16* outp = reinterpret_cast<bf16*>(part == 0 ? g.qh : (part == 1 ? g.kh : g.vh));
  float* rn_out = part == 0 ? g.rq : g.rk;
  // (batch, token) of the wave tile's first row: ONE integer division per call - a division by the run-time token count
  // per stored row (16 per lane per tile, ~30 VALU instructions each) was as much VALU work as the rest of this epilogue
  // (round 4: -2 % on the q/k/v GEMM, interleaved A/B)
  const int b_base = m_base / g.Ttok, t_base = m_base - b_base * g.Ttok;
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
        int b = b_base, tt = t_base + i * 16 + row;
        while (tt >= g.Ttok) {   // at most once when Ttok >= the tile's 128 rows
          tt -= g.Ttok;
          ++b;
        }
        st16_nt(outp + (((size_t)b * g.H + h) * g.Ttok + tt) * 64 + chunk * 8, raw);
      }
    }
  }
}

// ---- EPI 5: SwiGLU backward fused into the data-gradient GEMM of mlp_c_proj / out_proj ---------------
// (autograd of reference model.py:148-154, 259-261).  The accumulators hold dx = dL/d(u*silu(v)) for a
// 128x64 block of hidden columns; they are first packed to bf16 (the precision the unfused path stores dx
// in), which frees half the accumulator registers for the rest of the epilogue.  Per 16-row x 32-column
// piece, dx goes through the wave's LDS scratch into a row layout in which a lane owns 8 consecutive hidden
// columns of one row; the matching raw u and v chunks (16 B each, saved by EPI 3) are streamed straight from
// global memory through a register prefetch queue and d(uv) leaves as 16-byte stores, so uv/duv never touch
// LDS.  Column sums of d(suv) are reduced over the wave's 128 rows and written once per wave tile (no atomics).
template <int FMR>
__device__ __forceinline__ void nt_store_tile_swiglu_bwd(const NtArgs& g, f32x4 (&acc)[FMR][4], int m_base,
                                                         int n_base, int lane, char* scratch) {
  const int l15 = lane & 15, lg = lane >> 4;
  const int r = lane >> 2, qd = lane & 3;  // row-layout role: row r of the piece, hidden columns [8qd, 8qd+8)
  uint2 pk[FMR][4];
#pragma unroll
  for (int i = 0; i < FMR; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = (bf16)acc[i][j][e];
      pk[i][j] = __builtin_bit_cast(uint2, t);
      asm volatile("" : "+v"(pk[i][j].x), "+v"(pk[i][j].y));  // materialise here (the compiler otherwise sinks the pack)
    }
  __builtin_amdgcn_sched_barrier(0);  // pack first: everything below runs with the fp32 accumulators dead
  const int ucol = 32 * (qd >> 1) + 8 * (qd & 1);  // interleaved column of this lane's u chunk inside the piece
  const bf16* uvp = reinterpret_cast<const bf16*>(g.uv_in) + 2 * n_base + ucol;
  bf16* dp = reinterpret_cast<bf16*>(g.C) + 2 * n_base + ucol;
  constexpr int NH = 2 * FMR;  // pieces; hh = h * FMR + i, column half h outermost (its scales/sums stay live)
  constexpr int PD = 6;        // prefetch depth (pieces)
  uint4 pre[PD][2];
  // scratch piece: 16 rows x 64 B (32 bf16 columns), 16-byte chunks XOR-swizzled by (row >> 1) & 3
  char* wfrag = scratch + l15 * 64 + 8 * (lg & 1);
  const int wsw = (l15 >> 1) & 3;
  const char* rrow = scratch + r * 64 + ((qd ^ ((r >> 1) & 3)) << 4);
#define NVIT_SWB_ISSUE(hh_, dst_)                                                        \
  {                                                                                      \
    int m_ = m_base + ((hh_) % FMR) * 16 + r;                                            \
    m_ = m_ < g.M ? m_ : g.M - 1;                                                        \
    const bf16* p_ = uvp + (size_t)m_ * g.ld_uv + 64 * ((hh_) / FMR);                    \
    dst_[0] = *reinterpret_cast<const uint4*>(p_);                                       \
    dst_[1] = *reinterpret_cast<const uint4*>(p_ + 16);                                  \
  }
#pragma unroll
  for (int p = 0; p < PD; ++p) NVIT_SWB_ISSUE(p, pre[p]);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float gu[8], gv[8], su[8], sv[8];
    {
      f32x4 a0 = {1.f, 1.f, 1.f, 1.f}, a1 = a0, b0 = a0, b1 = a0;
      if (g.gs) {
        const float* gp = g.gs + n_base + 32 * h + 8 * qd;
        a0 = *reinterpret_cast<const f32x4*>(gp) * g.gscale;
        a1 = *reinterpret_cast<const f32x4*>(gp + 4) * g.gscale;
        b0 = *reinterpret_cast<const f32x4*>(gp + g.Fh) * g.gscale;
        b1 = *reinterpret_cast<const f32x4*>(gp + g.Fh + 4) * g.gscale;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gu[e] = a0[e];
        gu[e + 4] = a1[e];
        gv[e] = b0[e];
        gv[e + 4] = b1[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        su[e] = 0.f;
        sv[e] = 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < FMR; ++i) {
      const int hh = h * FMR + i;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        *reinterpret_cast<uint2*>(wfrag + (((2 * jj + (lg >> 1)) ^ wsw) << 4)) = pk[i][2 * h + jj];
      const bf16x8 db = *reinterpret_cast<const bf16x8*>(rrow);
      const bf16x8 ub = __builtin_bit_cast(bf16x8, pre[hh % PD][0]);
      const bf16x8 vb = __builtin_bit_cast(bf16x8, pre[hh % PD][1]);
      if (hh + PD < NH) NVIT_SWB_ISSUE(hh + PD, pre[hh % PD]);
      const int m = m_base + i * 16 + r;
      const float live = m < g.M ? 1.0f : 0.0f;
      uint4 dub, dvb;  // d(uv) of this lane's 8 columns, packed bf16
#pragma unroll
      for (int e4 = 0; e4 < 8; e4 += 4) {
        bf16x4 pu, pv;
#pragma unroll
        for (int e = e4; e < e4 + 4; ++e) {
          const float ur = (float)ub[e], vr = (float)vb[e];
          const float u = ur * gu[e], v = vr * gv[e];
          const float gg = (float)db[e];
          const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
          const float du = gg * v * sg;
          const float dv = gg * u * sg * (1.0f + v * (1.0f - sg));
          su[e] += du * ur * live;
          sv[e] += dv * vr * live;
          asm volatile("" : "+v"(su[e]), "+v"(sv[e]));  // accumulate here (else the whole chain is sunk into `if (part)`)
          pu[e - e4] = (bf16)(du * gu[e]);
          pv[e - e4] = (bf16)(dv * gv[e]);
        }
        const uint2 qu = __builtin_bit_cast(uint2, pu), qv = __builtin_bit_cast(uint2, pv);
        if (e4 == 0) {
          dub.x = qu.x, dub.y = qu.y, dvb.x = qv.x, dvb.y = qv.y;
          asm volatile("" : "+v"(dub.x), "+v"(dub.y), "+v"(dvb.x), "+v"(dvb.y));  // pack now, not at the store
        } else {
          dub.z = qu.x, dub.w = qu.y, dvb.z = qv.x, dvb.w = qv.y;
          asm volatile("" : "+v"(dub.z), "+v"(dub.w), "+v"(dvb.z), "+v"(dvb.w));
        }
        __builtin_amdgcn_sched_barrier(0);  // two groups of four columns: bounds the live temporaries
      }
      if (m < g.M) {
        bf16* o_ = dp + (size_t)m * g.ldc + 64 * h;
        st16_nt(o_, dub);
        st16_nt(o_ + 16, dvb);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch queue PD deep (no hoisting of later pieces' loads)
    }
    if (g.part) {
      // sum over the 16 row lanes that share qd (fixed order); lanes r == 0 write 8 consecutive columns
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = su[e], b = sv[e];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          b += __shfl_xor(b, o, 64);
        }
        su[e] = a * g.gscale;
        sv[e] = b * g.gscale;
      }
      if (r == 0) {
        float* pp = g.part + (size_t)(m_base >> 7) * (2 * g.Fh) + n_base + 32 * h + 8 * qd;
        *reinterpret_cast<f32x4*>(pp) = (f32x4){su[0], su[1], su[2], su[3]};
        *reinterpret_cast<f32x4*>(pp + 4) = (f32x4){su[4], su[5], su[6], su[7]};
        *reinterpret_cast<f32x4*>(pp + g.Fh) = (f32x4){sv[0], sv[1], sv[2], sv[3]};
        *reinterpret_cast<f32x4*>(pp + g.Fh + 4) = (f32x4){sv[4], sv[5], sv[6], sv[7]};
      }
    }
  }
#undef NVIT_SWB_ISSUE
}
