// impl 1: MFMA flash attention for gfx950 (bf16 operands, fp32 softmax/accumulate), head dim 64.
// Non-causal, unmasked, arbitrary (ragged) sequence lengths.
//
// Orientation (everything is computed "key-major" so that per-query softmax statistics are
// per-LANE scalars and the probability tile never leaves registers):
//   S^T[key][q] = K Q^T      MFMA-A = K rows (ds_read_b128 from the LDS K tile),
//                            MFMA-B = Q^T     (the lane's own Q row chunks, kept in registers)
//   O^T[d][q]  += V^T P^T    MFMA-B = P^T taken straight from the S^T accumulators (the k index
//                            inside one 32-deep MFMA step is permuted consistently on both
//                            operands), MFMA-A = V^T via ds_read_b64_tr_b16 (hardware transpose)
// One workgroup = 4 waves x 32 queries; K/V tiles of 64 keys arrive by LDS-DMA (global_load_lds, inline asm) into a
// 3-slot ring two tiles ahead, retired by a counted vmcnt, one barrier per tile.  LDS rows are 128 B (64 bf16)
// with the 16-byte chunk index XOR-swizzled by (row & 7): conflict-free for both the row reads and the
// transposed reads.  At head dim 64 these kernels are VALU-ISSUE bound, not MFMA bound (rocprofv3 PMC: VALU busy
// ~70 % + MFMA issue ~18 % of the SIMD cycles): per score the exp / scale / pack work costs about as many issue
// cycles as its share of the two MFMAs.  What the code below does about it: the bounded-score forward path (no
// running maximum), row constants folded into initial accumulators, scales applied once to the accumulators, tile
// addresses in scalar registers, and no compiler-visible load left pending across the tile loop (settle()).
//
// Backward (recompute, two kernels, no atomics, deterministic):
//   dq kernel : same structure as forward; per KV tile S^T, dP^T = V dO^T, dS^T = P^T(dP^T - delta),
//               dQ^T[d][q] += K^T dS^T   (K tile read by rows for S^T and transposed for dQ^T)
//   dkv kernel: one workgroup = 4 waves x 32 keys; loops over 64-query tiles (Q, dO, lse, delta in
//               LDS); S[q][key] = Q K^T, dP = dO V^T (query-major, key on the lane),
//               dV^T[d][key] += dO^T P, dK^T[d][key] += Q^T dS  (P/dS from accumulators, Q/dO
//               tiles read by rows and transposed).
#include <type_traits>

#include "common.h"

namespace {

constexpr int D = 64;         // head dim
constexpr int ROWB = D * 2;   // bytes per LDS row
constexpr int TKV = 64;       // rows per staged tile
constexpr int TILE_BYTES = TKV * ROWB;  // 8 KiB
constexpr float LOG2E = 1.4426950408889634f;

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// raw v_exp_f32 (no denormal range fix-up: arguments here are <= 0 after max subtraction, or bounded
// by the saved log-sum-exp in the backward kernels; results below 2^-126 flush to 0, which is the
// correct limit for a softmax weight)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * ROWB + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// transposed fragment: element j<4 = tile[row0 + 4*lg + j][col0 + l15], j>=4 = tile[row0 + 16 + 4*lg + (j-4)][..]
// (the k-slot order matching accumulators of two adjacent 16-row MFMA tiles used as the other operand)
__device__ __forceinline__ uint4 tr_frag(const char* tile, int row0, int col0, int l15, int lg) {
  const int q = l15 >> 2, p = l15 & 3;
  const int r0 = row0 + 4 * lg + q, r1 = r0 + 16;
  const int ch = (col0 >> 3) + (p >> 1);
  const int o0 = swz_off(r0, ch) + ((p & 1) << 3), o1 = swz_off(r1, ch) + ((p & 1) << 3);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + o0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + o1));
  uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// row fragment: tile[row0 + l15][32*ks + 8*lg .. +7]
__device__ __forceinline__ uint4 row_frag(const char* tile, int row0, int ks, int l15, int lg) {
  return *reinterpret_cast<const uint4*>(tile + swz_off(row0 + l15, ks * 4 + lg));
}

__device__ __forceinline__ uint4 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 v = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
  return __builtin_bit_cast(uint4, v);
}

__device__ __forceinline__ uint2 pack4(const f32x4& a) {
  bf16x4 v = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3]};
  return __builtin_bit_cast(uint2, v);
}

// Store a wave's [32 rows x 64] result held in accumulator layout (g[df][f][r] = out[row 16f + l15][col 16df + 4lg + r])
// as bf16 rows of 128 bytes: through a wave-private 4 KiB LDS scratch (XOR-swizzled like the K/V tiles) so that every
// global store instruction writes whole 128-byte rows, 16 bytes per lane (4 instructions per wave), instead of eight
// instructions of 8-byte pieces that each touch a quarter of 16 different rows.  dst -> element (row 0, col 0) of the
// tile, ld = row stride in elements, rows >= nvalid are not written.
__device__ __forceinline__ void store_tile32x64(const f32x4 (&g)[4][2], char* scr, bf16* dst, size_t ld, int nvalid,
                                                int lane) {
  const int l15 = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int df = 0; df < 4; ++df)
      *reinterpret_cast<uint2*>(scr + swz_off(16 * f + l15, 2 * df + (lg >> 1)) + 8 * (lg & 1)) = pack4(g[df][f]);
  __builtin_amdgcn_wave_barrier();   // DS instructions of one wave execute in order; this only pins the compiler
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int row = pass * 8 + (lane >> 3), chunk = lane & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(scr + swz_off(row, chunk));
    if (row < nvalid) *reinterpret_cast<uint4*>(dst + (size_t)row * ld + chunk * 8) = v;
  }
}

// LDS-DMA staging (global_load_lds_dwordx4 from inline asm, see gemm_common.h for why): lane L lands at
// lds_off + 16*L.  One wave-instruction = 8 rows of a 128-byte-row tile; the XOR swizzle is realised by
// fetching chunk (L&7) ^ (row&7).
// The source address is split into a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset:
// the per-tile part of the address (which tile) then lives in scalar registers and costs no VALU issue slots - these
// kernels are VALU-issue bound (PMC: VALU + MFMA issue ~ 87 % of the SIMD cycles), so address arithmetic is not free.
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_off) {
  unsigned keep;
  const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_off);
  // (the base is wave-uniform by construction; readfirstlane folds away when the compiler can see that, and keeps the
  //  "s" constraint satisfiable when it cannot - e.g. a pointer captured by reference in a lambda it did not dissolve)
  const unsigned long long ub = (unsigned long long)(uintptr_t)sbase;
  const unsigned long long sb = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ub) |
                                ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ub >> 32)) << 32);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sb), "s"(m)
      : "memory");
}
// The two pieces a wave contributes to a tile (LDS destinations 4 KiB apart) in one statement: one M0 save / restore.
__device__ __forceinline__ void glds16s_pair(const void* sbase, unsigned voff0, unsigned voff1, unsigned lds_off) {
  unsigned keep;
  const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_off);
  const unsigned long long ub = (unsigned long long)(uintptr_t)sbase;
  const unsigned long long sb = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ub) |
                                ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ub >> 32)) << 32);
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
      "s_add_u32 m0, %4, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff0), "v"(voff1), "s"(sb), "s"(m)
      : "memory", "scc");
}
// DMA one 64-row tile of `src` (row stride ld_bytes, rows clamped to nrows-1) to LDS byte offset tile_off;
// 4 waves x 2 wave-instructions.  TILE_DMA = instructions per wave per tile.  `src`, row_base and nrows are wave
// uniform; voff[i] = tile_voff(i, ...) are the lane's offsets inside a full tile, computed once per kernel.
constexpr int TILE_DMA = 2;
__device__ __forceinline__ unsigned tile_voff(int i, unsigned ld_bytes, int lane, int wid) {
  const int r8 = lane >> 3, chunk = (lane & 7) ^ r8;
  return (unsigned)((i * 4 + wid) * 8 + r8) * ld_bytes + (unsigned)chunk * 16u;
}
__device__ __forceinline__ void tile_dma(const bf16* src, unsigned ld_bytes, int row_base, int nrows, unsigned tile_off,
                                         int lane, int wid, const unsigned (&voff)[TILE_DMA]) {
  const char* sb = reinterpret_cast<const char*>(src) + (size_t)row_base * ld_bytes;
  if (row_base + TKV <= nrows) {
    static_assert(TILE_DMA == 2, "glds16s_pair issues the tile's two pieces");
    glds16s_pair(sb, voff[0], voff[1], tile_off + wid * 1024);
  } else {   // ragged last tile: rows past the end re-read the last valid row (finite values, masked by the consumer)
    const int r8 = lane >> 3, chunk = (lane & 7) ^ r8;
#pragma unroll
    for (int i = 0; i < TILE_DMA; ++i) {
      const int grp = i * 4 + wid;
      int row = grp * 8 + r8;
      row = row_base + row < nrows ? row : nrows - 1 - row_base;
      glds16s(sb, (unsigned)row * ld_bytes + (unsigned)chunk * 16u, tile_off + grp * 1024);
    }
  }
}
// Make the compiler retire its own pending global loads of `v` HERE (it inserts the s_waitcnt in front of this empty
// statement).  Without it the wait lands at the first use INSIDE the tile loop, where - hipcc does not see the
// LDS-DMA issued by inline asm - its vmcnt(0) also drains the K/V tiles that were just requested two tiles ahead,
// i.e. every tile pays a full memory round trip (cdna_hip_programming.md, "mixing load KINDS in one k-loop").
__device__ __forceinline__ void settle(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void settle(float& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)p);
}

// Work id -> (bh, tile) with XCD affinity: workgroups are dealt round-robin to the 8 XCDs, so block ids are re-dealt
// (bijectively) such that each XCD owns a contiguous range of work ids; with the tile index fastest, all tiles of one
// (batch, head) then run on ONE XCD at about the same time and its K/V (or Q/dO) stream is served by that XCD's L2
// instead of being fetched once per tile from HBM / Infinity Cache (measured 2.2 GB -> see profiles/).
__device__ __forceinline__ int work_index() {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
__device__ __forceinline__ void work_of(int ntile, int& bh, int& tile) {
  const int w = work_index();
  bh = w / ntile;
  tile = w - bh * ntile;
}

// ------------------------------------------------------------------------------------------ forward
// `sqk` != NULL (nViT call sites): q and k are s * unit vectors with s = sqk*c_q per channel, so every score obeys
// |q.k| <= smax^2 (smax = max_d |s_d| of the head).  When that bound is small enough that exp2 cannot underflow
// (BOUND_MAX), the kernel takes the FAST path: probabilities are taken relative to the bound instead of a running
// maximum - no per-tile max, no correction factor, no rescale of the output accumulators (about half of the
// VALU work of a tile at head dim 64, where this kernel is VALU-issue bound, not MFMA bound) - and the row sums come
// out of the MFMA pipe as one extra "ones" row of the V^T operand.  Softmax is shift invariant, so the result is the
// same function; lse = ln(sum) + bound.  Otherwise (generic q/k, or a learned scale beyond the limit) the online
// softmax below runs.
constexpr float BOUND_MAX = 60.0f;   // in log2 units: exp2(-2*60) is still a normal fp32 / bf16 number

__global__ __launch_bounds__(256, 3) void attn_fwd_mfma_kernel(const bf16* __restrict__ qh, const bf16* __restrict__ kh,
                                                             const bf16* __restrict__ vh, float scale, float qpre,
                                                             const float* __restrict__ sqk, float c_q,
                                                             bf16* __restrict__ o, float* __restrict__ lse, int H,
                                                             int Tq, int Tk) {
  __shared__ __attribute__((aligned(16))) char lds[3][2][TILE_BYTES];  // ring [slot][K|V]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  int bh, tile_;
  work_of((Tq + 127) / 128, bh, tile_);
  const int b = bh / H, h = bh % H;
  const int q0 = tile_ * 128 + wid * 32;
  const bf16* kbase = kh + (size_t)bh * Tk * D;
  const bf16* vbase = vh + (size_t)bh * Tk * D;
  // qh holds qpre * q_hat (the producer folds the factor into the learned scale, one rounding): the MFMA result times
  // c2 is the score in log2 units.  qpre = scale * log2(e) makes c2 exactly 1 (UNIT): the fast path then needs no
  // multiply at all - the accumulator starts at -bound and goes straight into v_exp_f32.
  const float c2t = scale * LOG2E;
  const float c2 = c2t / qpre;
  const bool unit = __builtin_amdgcn_readfirstlane(fabsf(c2 - 1.0f) < 1e-6f ? 1 : 0) != 0;

  uint4 qf[2][2];
  f32x4 oacc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int f = 0; f < 2; ++f) oacc[i][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_[2] = {-INFINITY, -INFINITY}, l_[2] = {0.f, 0.f};
  // score bound of this head in log2 units (fast path) - wave-uniform
  float tb = INFINITY;
  if (sqk) {
    float sm = fabsf(sqk[h * D + lane] * c_q);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sm = fmaxf(sm, __shfl_xor(sm, off, 64));
    tb = c2t * sm * sm;
  }
  const bool fast = __builtin_amdgcn_readfirstlane(tb <= BOUND_MAX ? 1 : 0) != 0;
  f32x4 lacc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};   // fast path: row sums from the MFMA pipe
  const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);   // 8 x bf16 1.0

  const int nt = (Tk + TKV - 1) / TKV;
  // K/V ring: LDS-DMA two tiles ahead, counted vmcnt (4 younger DMA instructions may stay in flight)
  const unsigned ring = lds_addr(&lds[0][0][0]);
  const unsigned voff[TILE_DMA] = {tile_voff(0, ROWB, lane, wid), tile_voff(1, ROWB, lane, wid)};
  tile_dma(kbase, ROWB, 0, Tk, ring, lane, wid, voff);
  tile_dma(vbase, ROWB, 0, Tk, ring + TILE_BYTES, lane, wid, voff);
  if (nt > 1) {
    tile_dma(kbase, ROWB, TKV, Tk, ring + 2 * TILE_BYTES, lane, wid, voff);
    tile_dma(vbase, ROWB, TKV, Tk, ring + 3 * TILE_BYTES, lane, wid, voff);
  }
  // the wave's own Q rows: requested AFTER the DMA (one memory round trip for everything) and settled before the loop
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    int q = q0 + 16 * f + l15;
    q = q < Tq ? q : Tq - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[f][ks] = *reinterpret_cast<const uint4*>(qh + ((size_t)bh * Tq + q) * D + ks * 32 + lg * 8);
  }
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) settle(qf[f][ks]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the compiler's own wait above already drained the queue)
  __syncthreads();
  int cur = 0;
  // Ragged sequence lengths (T = 784 = 6.125 x 128 = 12.25 x 64): a wave whose 32 queries all lie past Tq only feeds the
  // K/V ring and the barriers (wave_active), and the ragged last KV tile - peeled, so the steady-state tiles spend no
  // VALU on masks - multiplies and exponentiates only its valid 16-key fragments (nkf) / 32-key MFMA steps (ns2).
  const bool wave_active = q0 < Tq;
  f32x4 ntb = {-tb, -tb, -tb, -tb};   // UNIT: initial accumulator of the score product
  asm volatile("" : "+v"(ntb));
  auto tile_body = [&](const int t, auto masked_, auto fast_, auto unit_) {
    constexpr bool MASKED = decltype(masked_)::value;
    constexpr bool FAST = decltype(fast_)::value;
    constexpr bool UNIT = decltype(unit_)::value;
    if (t + 2 < nt) {
      const int sl = cur == 0 ? 2 : cur - 1;  // (t + 2) % 3
      tile_dma(kbase, ROWB, (t + 2) * TKV, Tk, ring + (2 * sl) * TILE_BYTES, lane, wid, voff);
      tile_dma(vbase, ROWB, (t + 2) * TKV, Tk, ring + (2 * sl + 1) * TILE_BYTES, lane, wid, voff);
    }
    if (wave_active && FAST) {
      const char* kt = &lds[cur][0][0];
      const char* vt = &lds[cur][1][0];
      const int nvalid = MASKED ? Tk - t * TKV : TKV;
      const int nkf = MASKED ? (nvalid + 15) >> 4 : 4, ns2 = MASKED ? (nvalid + 31) >> 5 : 2;
      uint4 pf[2][2];  // [s2][f]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f32x4 p_[2][2];   // [kk][f], key fragment kf = 2*s2 + kk
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int kf = 2 * s2 + kk;
          if (!MASKED || kf < nkf) {
            const uint4 a0 = row_frag(kt, kf * 16, 0, l15, lg), a1 = row_frag(kt, kf * 16, 1, l15, lg);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              f32x4 z = UNIT ? ntb : (f32x4){0.f, 0.f, 0.f, 0.f};
              z = mfma16(a0, qf[f][0], z);
              z = mfma16(a1, qf[f][1], z);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float p = UNIT ? fast_exp2(z[r]) : fast_exp2(z[r] * c2 - tb);
                if (MASKED && kf * 16 + lg * 4 + r >= nvalid) p = 0.f;
                p_[kk][f][r] = p;
              }
            }
          } else {
            p_[kk][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            p_[kk][1] = p_[kk][0];
          }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) pf[s2][f] = pack8(p_[0][f], p_[1][f]);
      }
      // O^T[df][f] += V^T P^T, and the row sums l[f] += 1^T P^T
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        if (!MASKED || s2 < ns2) {
#pragma unroll
          for (int f = 0; f < 2; ++f) lacc[f] = mfma16(ones, pf[s2][f], lacc[f]);
#pragma unroll
          for (int df = 0; df < 4; ++df) {
            const uint4 va = tr_frag(vt, s2 * 32, df * 16, l15, lg);
#pragma unroll
            for (int f = 0; f < 2; ++f) oacc[df][f] = mfma16(va, pf[s2][f], oacc[df][f]);
          }
        }
    }
    if (wave_active && !FAST) {
      const char* kt = &lds[cur][0][0];
      const char* vt = &lds[cur][1][0];
      const int kbase_i = t * TKV;
      const int nvalid = MASKED ? Tk - kbase_i : TKV;
      const int nkf = MASKED ? (nvalid + 15) >> 4 : 4, ns2 = MASKED ? (nvalid + 31) >> 5 : 2;
      // S^T[kf][f] : rows = key 16kf + 4lg + r, col = query 16f + l15
      f32x4 s[4][2];
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) {
        if (!MASKED || kf < nkf) {
          const uint4 a0 = row_frag(kt, kf * 16, 0, l15, lg), a1 = row_frag(kt, kf * 16, 1, l15, lg);
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            z = mfma16(a0, qf[f][0], z);
            s[kf][f] = mfma16(a1, qf[f][1], z);
          }
          if constexpr (MASKED) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kf * 16 + lg * 4 + r >= nvalid) {
                s[kf][0][r] = -INFINITY;
                s[kf][1][r] = -INFINITY;
              }
          }
        } else {
          s[kf][0] = (f32x4){0.f, 0.f, 0.f, 0.f};   // fragment past the end: probability 0, no work
          s[kf][1] = s[kf][0];
        }
      }
      uint4 pf[2][2];  // [s2][f]
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        float mt = -INFINITY;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
          if (!MASKED || kf < nkf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) mt = fmaxf(mt, s[kf][f][r]);
          }
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mn = fmaxf(m_[f], mt);
        const float corr = fast_exp2((m_[f] - mn) * c2);
        m_[f] = mn;
        const float mc = mn * c2;
        float rs = 0.f;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
          if (!MASKED || kf < nkf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = fast_exp2(s[kf][f][r] * c2 - mc);
              s[kf][f][r] = p;
              rs += p;
            }
          }
        l_[f] = l_[f] * corr + rs;
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc[i][f] = oacc[i][f] * corr;
        pf[0][f] = pack8(s[0][f], s[1][f]);
        pf[1][f] = pack8(s[2][f], s[3][f]);
      }
      // O^T[df][f] += V^T P^T
#pragma unroll
      for (int df = 0; df < 4; ++df)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          if (!MASKED || s2 < ns2) {
            const uint4 va = tr_frag(vt, s2 * 32, df * 16, l15, lg);
#pragma unroll
            for (int f = 0; f < 2; ++f) oacc[df][f] = mfma16(va, pf[s2][f], oacc[df][f]);
          }
    }
    if (t + 2 < nt)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur = cur == 2 ? 0 : cur + 1;
  };
  // (spelled out: wrapping the three variants in a second generic lambda keeps hipcc from dissolving the captures, and
  //  the accumulators end up in scratch memory)
#define NVIT_RUN_TILES(FAST_, UNIT_)                                                            \
  {                                                                                             \
    for (int t = 0; t + 1 < nt; ++t) tile_body(t, std::false_type{}, FAST_{}, UNIT_{});         \
    if (Tk % TKV)                                                                               \
      tile_body(nt - 1, std::true_type{}, FAST_{}, UNIT_{});                                    \
    else                                                                                        \
      tile_body(nt - 1, std::false_type{}, FAST_{}, UNIT_{});                                   \
  }
  if (fast && unit)
    NVIT_RUN_TILES(std::true_type, std::true_type)
  else if (fast)
    NVIT_RUN_TILES(std::true_type, std::false_type)
  else
    NVIT_RUN_TILES(std::false_type, std::false_type)
#undef NVIT_RUN_TILES
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    float l, lse_v;
    if (fast) {
      l = lacc[f][0];   // every row of the ones-operand product is the full sum over the keys
      lse_v = (tb + log2f(l)) * (1.0f / LOG2E);
    } else {
      l = l_[f];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      lse_v = m_[f] * (c2 * (1.0f / LOG2E)) + logf(l);   // m_ is in units of the (pre-scaled) MFMA result
    }
    const int q = q0 + 16 * f + l15;
    const float inv = 1.0f / l;
#pragma unroll
    for (int df = 0; df < 4; ++df) oacc[df][f] = oacc[df][f] * inv;
    if (q < Tq && lg == 0) lse[(size_t)bh * Tq + q] = lse_v;
  }
  if (wave_active)   // (the tile loop ended with a barrier: the ring is free, each wave takes 4 KiB of it as scratch)
    store_tile32x64(oacc, &lds[0][0][0] + wid * 4096, o + ((size_t)b * Tq + q0) * (H * D) + h * D, (size_t)H * D, Tq - q0,
                    lane);
}

// Fused backward of  x_hat = (sqk*c_q) * x/||x||  (reference model.py:108-112) in the epilogue of the attention
// backward kernels: g = dL/dx_hat in the accumulator layout (row = row0 + 16f + l15, d = 16df + 4lg + r).
// Writes dL/dx (bf16, token-major) and this workgroup's partial sums of dL/d(sqk*c_q) for its head.
struct QkFuse {
  // (pointers first, the three 4-byte scalars together: hipcc widens the load of a scalar that is splat into a vector to 16
  //  bytes; next to a pointer that slice of the struct cannot be kept in registers and is parked in LDS for the whole
  //  kernel - 16 bytes per lane, flat stores through the reloaded pointer)
  const float* rn;    // [B*T, H] 1/||x|| saved by the forward
  const float* sqk;   // [C]
  bf16* out;          // token-major gradient of the projection output: out[(b*T + row)*ld + h*64 + d]
  bf16* out_v;        // (dk/dv kernel only) same for the value projection
  float* part;        // [B * gridDim.x, C]
  float c_q;
  float xs;           // the saved x_hat rows carry an extra factor 1/xs (pre-scaled q): multiply by xs before use
  int ld;
};

// What the epilogue reads from global memory (saved unit-direction rows, 1/norm, per-channel scale): requested as early as
// the caller can - behind the tile loop, before the accumulators are even in place - so that the fetch latency runs under the
// value-gradient store instead of in front of the arithmetic.
struct QkEpiLoads {
  uint2 xr[2][4];
  float rn[2];
  f32x4 s[4];
};
__device__ __forceinline__ void qk_bwd_epilogue_loads(QkEpiLoads& L, const bf16* xh_bh, const QkFuse& fu, int row0, int T, int H,
                                                      int b, int h, int lane) {
  const int l15 = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int row = row0 + 16 * f + l15;
    const int rc = row < T ? row : T - 1;
#pragma unroll
    for (int df = 0; df < 4; ++df)
      L.xr[f][df] = *reinterpret_cast<const uint2*>(xh_bh + (size_t)rc * 64 + df * 16 + 4 * lg);
    L.rn[f] = fu.rn[((size_t)b * T + rc) * H + h];
  }
#pragma unroll
  for (int df = 0; df < 4; ++df) L.s[df] = *reinterpret_cast<const f32x4*>(fu.sqk + h * 64 + df * 16 + 4 * lg);
}

// v + (v of the lane the DPP control CTRL names): a row-internal exchange on the VALU, no LDS round trip
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
  return v + __builtin_bit_cast(float, o);
}

__device__ __forceinline__ void qk_bwd_epilogue(f32x4 (&g)[4][2], const QkEpiLoads& L, const QkFuse& fu, int row0, int T,
                                                int H, int b, int h, int lane, int wid, char* lds0, int tile,
                                                int ntile, int t256, bool item_valid = true) {
  // t256: index within the 256 threads that share the work item (= threadIdx.x, or its low 8 bits in the two-item kernel)
  // lds0: the (now idle) tile ring; bytes [4096*wid, +4096) = this wave's store scratch, [16384, +1024) = column sums
  const int l15 = lane & 15, lg = lane >> 4;
  float* red = reinterpret_cast<float*>(lds0 + 16384);
  f32x4 s[4], sinv[4], ds[4];
#pragma unroll
  for (int df = 0; df < 4; ++df) {
    s[df] = L.s[df] * fu.c_q;
    ds[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) sinv[df][e] = s[df][e] != 0.f ? __builtin_amdgcn_rcpf(s[df][e]) * fu.xs : 0.f;   // 1 ulp
  }
  f32x4 outv[4][2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const bool valid = row0 + 16 * f + l15 < T;
    f32x4 n[4], sg[4];
    float dot = 0.f;
#pragma unroll
    for (int df = 0; df < 4; ++df) {
      const bf16x4 xb = __builtin_bit_cast(bf16x4, L.xr[f][df]);
      n[df] = (f32x4){(float)xb[0], (float)xb[1], (float)xb[2], (float)xb[3]} * sinv[df];
      if (valid) ds[df] += g[df][f] * n[df];
      sg[df] = g[df][f] * s[df];
      dot += sg[df][0] * n[df][0] + sg[df][1] * n[df][1] + sg[df][2] * n[df][2] + sg[df][3] * n[df][3];
    }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
#pragma unroll
    for (int df = 0; df < 4; ++df) outv[df][f] = (sg[df] - n[df] * dot) * L.rn[f];
  }
  if (row0 < T)
    store_tile32x64(outv, lds0 + wid * 4096, fu.out + ((size_t)b * T + row0) * fu.ld + h * 64, (size_t)fu.ld, T - row0, lane);
  // column sums over this workgroup's 128 rows: 16 lanes (l15) -> 4 waves -> one partial row.  The 16-lane step is the
  // butterfly (lane ^ 1, ^ 2, ^ 4, ^ 8) as DPP row exchanges - quad swaps, then the half-row and the row mirrored, which
  // pair the same partial sums as the xor pattern does (the bits are those of the butterfly: a + b == b + a) - so it costs
  // 64 VALU adds and no LDS round trips (as 64 dependent ds_bpermute, each waited for, it was most of this epilogue's
  // time); every lane ends up with all 16 totals of its row group and stores the one its index names.
  float pick = 0.f;
#pragma unroll
  for (int df = 0; df < 4; ++df)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = ds[df][e];
      v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
      v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
      v = dpp_add<0x141>(v);   // row_half_mirror
      v = dpp_add<0x140>(v);   // row_mirror
      pick = l15 == df * 4 + e ? v : pick;
    }
  red[wid * 64 + (l15 >> 2) * 16 + 4 * lg + (l15 & 3)] = pick;
  __syncthreads();
  if (t256 < 64 && item_valid) {
    const float t = red[t256] + red[64 + t256] + red[128 + t256] + red[192 + t256];
    fu.part[((size_t)b * ntile + tile) * (H * 64) + h * 64 + t256] = t;
  }
}

// ------------------------------------------------------------------------------------------ dQ
template <bool FUSE>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_mfma_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ qh,
                                                                const bf16* __restrict__ kh, const bf16* __restrict__ vh,
                                                                const float* __restrict__ lse,
                                                                const bf16* __restrict__ og, float* __restrict__ delta,
                                                                float scale, float qpre, bf16* __restrict__ dqh, int H,
                                                                int Tq, int Tk, QkFuse fu) {
  // This kernel also produces delta[bh][q] = <dO_q, O_q> (the softmax-backward row term) from the attention output `og`
  // (token-major like dout) and stores it, with lse in log2 units, in the [2,B,H,Tq] side buffer for the dk/dv kernel,
  // which runs after it.  (og == NULL: delta is an input and no side buffer is written - not used by the launchers.)
  __shared__ __attribute__((aligned(16))) char lds[3][2][TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  int bh, tile_;
  work_of((Tq + 127) / 128, bh, tile_);
  const int b = bh / H, h = bh % H;
  const int q0 = tile_ * 128 + wid * 32;
  const bf16* kbase = kh + (size_t)bh * Tk * D;
  const bf16* vbase = vh + (size_t)bh * Tk * D;
  const float c2 = scale * LOG2E / qpre;   // see the forward kernel: 1 when q was pre-scaled by scale * log2(e)
  const bool unit = __builtin_amdgcn_readfirstlane(fabsf(c2 - 1.0f) < 1e-6f ? 1 : 0) != 0;

  uint4 qf[2][2], gf[2][2];
  float lse2[2], dl[2];
  const int nt = (Tk + TKV - 1) / TKV;
  // K/V ring: LDS-DMA two tiles ahead, counted vmcnt (4 younger DMA instructions may stay in flight); the prologue
  // DMA goes out first, the wave's own rows after it, and everything is settled before the tile loop (see settle())
  const unsigned ring = lds_addr(&lds[0][0][0]);
  const unsigned voff[TILE_DMA] = {tile_voff(0, ROWB, lane, wid), tile_voff(1, ROWB, lane, wid)};
  tile_dma(kbase, ROWB, 0, Tk, ring, lane, wid, voff);
  tile_dma(vbase, ROWB, 0, Tk, ring + TILE_BYTES, lane, wid, voff);
  if (nt > 1) {
    tile_dma(kbase, ROWB, TKV, Tk, ring + 2 * TILE_BYTES, lane, wid, voff);
    tile_dma(vbase, ROWB, TKV, Tk, ring + 3 * TILE_BYTES, lane, wid, voff);
  }
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int qu = q0 + 16 * f + l15;
    const int q = qu < Tq ? qu : Tq - 1;
    lse2[f] = lse[(size_t)bh * Tq + q] * LOG2E;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[f][ks] = *reinterpret_cast<const uint4*>(qh + ((size_t)bh * Tq + q) * D + ks * 32 + lg * 8);
      gf[f][ks] = *reinterpret_cast<const uint4*>(dout + ((size_t)b * Tq + q) * (H * D) + h * D + ks * 32 + lg * 8);
    }
    if (og) {
      float part = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const uint4 of = *reinterpret_cast<const uint4*>(og + ((size_t)b * Tq + q) * (H * D) + h * D + ks * 32 + lg * 8);
        const bf16x8 a = __builtin_bit_cast(bf16x8, gf[f][ks]), c = __builtin_bit_cast(bf16x8, of);
#pragma unroll
        for (int e = 0; e < 8; ++e) part += (float)a[e] * (float)c[e];
      }
      part += __shfl_xor(part, 16, 64);   // the row's 64 columns live on the 4 lanes l15 + 16*lg
      part += __shfl_xor(part, 32, 64);
      dl[f] = part;
      if (lg == 0 && qu < Tq) {
        // side buffer for the dk/dv kernel (runs after this one): -delta and -lse in log2 units (accumulator seeds)
        delta[(size_t)bh * Tq + qu] = -part;
        delta[((size_t)(gridDim.x / ((Tq + 127) / 128)) + bh) * Tq + qu] = -lse2[f];
      }
    } else {
      dl[f] = -delta[(size_t)bh * Tq + q];
    }
  }
  f32x4 dq[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int f = 0; f < 2; ++f) dq[i][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    settle(lse2[f]);
    settle(dl[f]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      settle(qf[f][ks]);
      settle(gf[f][ks]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // -delta as the initial accumulator of the dP product: loop-invariant register quads (passed as the MFMA's C operand)
  f32x4 ndl[2], nls[2];   // nls (UNIT): -lse as the initial accumulator of the score product
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    ndl[f] = (f32x4){-dl[f], -dl[f], -dl[f], -dl[f]};
    nls[f] = (f32x4){-lse2[f], -lse2[f], -lse2[f], -lse2[f]};
    asm volatile("" : "+v"(ndl[f]));
    asm volatile("" : "+v"(nls[f]));
  }
  int cur = 0;
  const bool wave_active = q0 < Tq;   // see the forward kernel
  auto tile_body = [&](const int t, auto masked_, auto unit_) {
    constexpr bool MASKED = decltype(masked_)::value;
    constexpr bool UNIT = decltype(unit_)::value;
    if (t + 2 < nt) {
      const int sl = cur == 0 ? 2 : cur - 1;  // (t + 2) % 3
      tile_dma(kbase, ROWB, (t + 2) * TKV, Tk, ring + (2 * sl) * TILE_BYTES, lane, wid, voff);
      tile_dma(vbase, ROWB, (t + 2) * TKV, Tk, ring + (2 * sl + 1) * TILE_BYTES, lane, wid, voff);
    }
    if (wave_active) {
      const char* kt = &lds[cur][0][0];
      const char* vt = &lds[cur][1][0];
      const int kbase_i = t * TKV;
      const int nvalid = MASKED ? Tk - kbase_i : TKV;
      const int nkf = MASKED ? (nvalid + 15) >> 4 : 4, ns2 = MASKED ? (nvalid + 31) >> 5 : 2;
      uint4 dsf[2][2];  // [s2][f]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (!MASKED || s2 < ns2) {
          f32x4 ds_[2][2];  // [kk][f] for key frags kf = 2*s2 + kk
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int kf = 2 * s2 + kk;
            if (!MASKED || kf < nkf) {
              const uint4 a0 = row_frag(kt, kf * 16, 0, l15, lg), a1 = row_frag(kt, kf * 16, 1, l15, lg);
              const uint4 v0 = row_frag(vt, kf * 16, 0, l15, lg), v1 = row_frag(vt, kf * 16, 1, l15, lg);
#pragma unroll
              for (int f = 0; f < 2; ++f) {
                f32x4 z = UNIT ? nls[f] : (f32x4){0.f, 0.f, 0.f, 0.f};
                z = mfma16(a0, qf[f][0], z);
                z = mfma16(a1, qf[f][1], z);  // S^T (UNIT: already minus lse, in log2 units)
                f32x4 w = mfma16(v0, gf[f][0], ndl[f]);        // row constant -delta as the initial accumulator
                w = mfma16(v1, gf[f][1], w);  // dP^T - delta
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const bool valid = !MASKED || (kf * 16 + lg * 4 + r) < nvalid;
                  const float p = valid ? (UNIT ? fast_exp2(z[r]) : fast_exp2(z[r] * c2 - lse2[f])) : 0.f;
                  ds_[kk][f][r] = p * w[r];   // the softmax scale is applied once, to the dQ accumulators
                }
              }
            } else {
              ds_[kk][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
              ds_[kk][1] = ds_[kk][0];
            }
          }
#pragma unroll
          for (int f = 0; f < 2; ++f) dsf[s2][f] = pack8(ds_[0][f], ds_[1][f]);
        }
      }
      // dQ^T[df][f] += K^T dS^T
#pragma unroll
      for (int df = 0; df < 4; ++df)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          if (!MASKED || s2 < ns2) {
            const uint4 ka = tr_frag(kt, s2 * 32, df * 16, l15, lg);
#pragma unroll
            for (int f = 0; f < 2; ++f) dq[df][f] = mfma16(ka, dsf[s2][f], dq[df][f]);
          }
    }
    if (t + 2 < nt)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur = cur == 2 ? 0 : cur + 1;
  };
#define NVIT_RUN_TILES(UNIT_)                                                          \
  {                                                                                    \
    for (int t = 0; t + 1 < nt; ++t) tile_body(t, std::false_type{}, UNIT_{});         \
    if (Tk % TKV)                                                                      \
      tile_body(nt - 1, std::true_type{}, UNIT_{});                                    \
    else                                                                               \
      tile_body(nt - 1, std::false_type{}, UNIT_{});                                   \
  }
  if (unit)
    NVIT_RUN_TILES(std::true_type)
  else
    NVIT_RUN_TILES(std::false_type)
#undef NVIT_RUN_TILES
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int f = 0; f < 2; ++f) dq[i][f] = dq[i][f] * scale;   // d/d(q_hat): K is not pre-scaled
  if constexpr (FUSE) {
    QkEpiLoads el;
    qk_bwd_epilogue_loads(el, qh + (size_t)bh * Tq * D, fu, q0, Tq, H, b, h, lane);
    qk_bwd_epilogue(dq, el, fu, q0, Tq, H, b, h, lane, wid, &lds[0][0][0], tile_, (Tq + 127) / 128, (int)threadIdx.x);
  } else {
    if (wave_active)
      store_tile32x64(dq, &lds[0][0][0] + wid * 4096, dqh + ((size_t)bh * Tq + q0) * D, (size_t)D, Tq - q0, lane);
  }
}

// ------------------------------------------------------------------------------------------ dK, dV
// 4-byte-per-lane LDS-DMA (one wave-instruction = 64 consecutive floats): the per-query lse / delta of a tile
__device__ __forceinline__ void glds4a(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_off);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(m)
      : "memory");
}
// One workgroup = 4 waves x 32 keys; the 64-query tiles of Q and dO (+ their lse / delta rows) arrive by LDS-DMA into a
// 3-slot ring two tiles ahead (counted vmcnt, one barrier per tile) - the same machine as the forward / dq kernels,
// instead of the register-staged double buffer this kernel used before (which spilled at 3 waves per SIMD).
// -delta enters as the initial accumulator of the dP product, the softmax scale is applied once to the dK accumulators,
// and P / dS are packed to bf16 as soon as a 16-query fragment is done, so only packed halves stay live.
constexpr int DKV_SLOT = 2 * TILE_BYTES + 512;     // Q tile | dO tile | lse[64] | delta[64]
constexpr int DKV_DMA = 2 * TILE_DMA + 2;          // DMA wave-instructions per wave per tile 
constexpr int DKV_WAVES = 2;   // waves per SIMD the register budget is sized for (222 VGPRs; at 3 the kernel spills 118)

template <bool FUSE>
__global__ __launch_bounds__(256, DKV_WAVES) void attn_bwd_dkv_mfma_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ qh,
                                                                 const bf16* __restrict__ kh, const bf16* __restrict__ vh,
                                                                 const float* __restrict__ lse,
                                                                 const float* __restrict__ delta, float scale,
                                                                 float qpre, bf16* __restrict__ dkh,
                                                                 bf16* __restrict__ dvh, int H, int Tq, int Tk,
                                                                 QkFuse fu) {
  __shared__ __attribute__((aligned(16))) char lds[3 * DKV_SLOT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  int bh, tile_;
  work_of((Tk + 127) / 128, bh, tile_);
  const int b = bh / H, h = bh % H;
  const int k0 = tile_ * 128 + wid * 32;
  const bf16* qbase = qh + (size_t)bh * Tq * D;
  const bf16* gbase = dout + (size_t)b * Tq * (H * D) + h * D;
  const float* dbase = delta + (size_t)bh * Tq;                                   // -delta, written by the dq kernel
  const float* lbase = delta + ((size_t)(gridDim.x / ((Tk + 127) / 128)) + bh) * Tq;   // -lse * log2(e), written by the dq kernel
  const float c2 = scale * LOG2E / qpre;   // see the forward kernel
  const bool unit = __builtin_amdgcn_readfirstlane(fabsf(c2 - 1.0f) < 1e-6f ? 1 : 0) != 0;
  const int nt = (Tq + TKV - 1) / TKV;
  const unsigned ring = lds_addr(&lds[0]);

  const unsigned ldg_bytes = (unsigned)(H * D * 2);
  const unsigned voff_q[TILE_DMA] = {tile_voff(0, ROWB, lane, wid), tile_voff(1, ROWB, lane, wid)};
  const unsigned voff_g[TILE_DMA] = {tile_voff(0, ldg_bytes, lane, wid), tile_voff(1, ldg_bytes, lane, wid)};
  auto tile_issue = [&](int t, int slot) {
    const unsigned so = ring + (unsigned)slot * DKV_SLOT;
    tile_dma(qbase, ROWB, t * TKV, Tq, so, lane, wid, voff_q);
    tile_dma(gbase, ldg_bytes, t * TKV, Tq, so + TILE_BYTES, lane, wid, voff_g);
    int q = t * TKV + lane;
    q = q < Tq ? q : Tq - 1;
    glds4a(lbase + q, so + 2 * TILE_BYTES);        // every wave writes the same 256 bytes (keeps vmcnt uniform)
    glds4a(dbase + q, so + 2 * TILE_BYTES + 256);
  };
  tile_issue(0, 0);
  if (nt > 1) tile_issue(1, 1);

  uint4 kf_[2][2], vf_[2][2];  // [key frag][ks]: K / V rows of this wave's 32 keys (MFMA-B operands)
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    int k = k0 + 16 * f + l15;
    k = k < Tk ? k : Tk - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf_[f][ks] = *reinterpret_cast<const uint4*>(kh + ((size_t)bh * Tk + k) * D + ks * 32 + lg * 8);
      vf_[f][ks] = *reinterpret_cast<const uint4*>(vh + ((size_t)bh * Tk + k) * D + ks * 32 + lg * 8);
    }
  }
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      settle(kf_[f][ks]);
      settle(vf_[f][ks]);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 dk[4][2], dv[4][2];  // [df][key frag]: rows d = 16df + 4lg + r, col key = l15
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      dk[i][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dv[i][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  const bool wave_active = k0 < Tk;   // a wave whose 32 keys all lie past Tk only feeds the ring and the barriers
  int cur = 0;
  auto tile_body = [&](const int t, auto masked_, auto unit_) {
    constexpr bool MASKED = decltype(masked_)::value;
    constexpr bool UNIT = decltype(unit_)::value;
    if (t + 2 < nt) tile_issue(t + 2, cur == 0 ? 2 : cur - 1);
    if (wave_active) {
      const char* qt = &lds[cur * DKV_SLOT];
      const char* gt = qt + TILE_BYTES;
      const float* st = reinterpret_cast<const float*>(qt + 2 * TILE_BYTES);
      const int nvalid = MASKED ? Tq - t * TKV : TKV;   // queries of this tile that exist
      const int nqf = MASKED ? (nvalid + 15) >> 4 : 4, ns2 = MASKED ? (nvalid + 31) >> 5 : 2;
      if constexpr (!MASKED) {
        // Full tiles: every LDS read of a 32-query half (row fragments of Q and dO, -lse, -delta, transposed fragments) is
        // issued before its first MFMA, the four accumulator chains of a query fragment are interleaved, and the region
        // is fenced - hipcc otherwise issues each fragment group just in time and waits for the LDS eight times per tile
        // (245 instead of 286 instructions per tile, 6 instead of 28 lgkmcnt waits).
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          uint4 a[2][2], gg[2][2], ga[4], qa[4];
          f32x4 nl[2], nd[2];
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const int qfi = 2 * s2 + qq;
            a[qq][0] = row_frag(qt, qfi * 16, 0, l15, lg), a[qq][1] = row_frag(qt, qfi * 16, 1, l15, lg);
            gg[qq][0] = row_frag(gt, qfi * 16, 0, l15, lg), gg[qq][1] = row_frag(gt, qfi * 16, 1, l15, lg);
            nl[qq] = *reinterpret_cast<const f32x4*>(st + qfi * 16 + 4 * lg);        // -lse (log2 units) of the 4 queries
            nd[qq] = *reinterpret_cast<const f32x4*>(st + 64 + qfi * 16 + 4 * lg);   // -delta
          }
#pragma unroll
          for (int df = 0; df < 4; ++df) {
            ga[df] = tr_frag(gt, s2 * 32, df * 16, l15, lg);   // dO^T[d][q slots]
            qa[df] = tr_frag(qt, s2 * 32, df * 16, l15, lg);   // Q^T[d][q slots]
          }
          __builtin_amdgcn_sched_barrier(0);
          uint2 ph[2][2], sh[2][2];   // [qq][key frag] packed bf16 P / dS of query frag 2*s2 + qq
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            f32x4 z[2], w[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              z[f] = mfma16(a[qq][0], kf_[f][0], UNIT ? nl[qq] : (f32x4){0.f, 0.f, 0.f, 0.f});
              w[f] = mfma16(gg[qq][0], vf_[f][0], nd[qq]);   // row constants -delta as the initial accumulator
            }
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              z[f] = mfma16(a[qq][1], kf_[f][1], z[f]);    // S[q][key] (UNIT: already minus lse, in log2 units)
              w[f] = mfma16(gg[qq][1], vf_[f][1], w[f]);   // dP[q][key] - delta[q]
            }
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              f32x4 p, dsv;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                p[r] = UNIT ? fast_exp2(z[f][r]) : fast_exp2(z[f][r] * c2 + nl[qq][r]);
                dsv[r] = p[r] * w[f][r];
              }
              ph[qq][f] = pack4(p);
              sh[qq][f] = pack4(dsv);
            }
          }
          uint4 pb[2], sb[2];
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            pb[f] = make_uint4(ph[0][f].x, ph[0][f].y, ph[1][f].x, ph[1][f].y);
            sb[f] = make_uint4(sh[0][f].x, sh[0][f].y, sh[1][f].x, sh[1][f].y);
          }
#pragma unroll
          for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              dv[df][f] = mfma16(ga[df], pb[f], dv[df][f]);
              dk[df][f] = mfma16(qa[df], sb[f], dk[df][f]);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (s2 >= ns2) continue;
        uint2 ph[2][2], sh[2][2];  // [qq][key frag] packed bf16 P / dS of query frag qfi = 2*s2 + qq (rows 16qfi + 4lg + r)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int qfi = 2 * s2 + qq;
          if (!MASKED || qfi < nqf) {
            const uint4 a0 = row_frag(qt, qfi * 16, 0, l15, lg), a1 = row_frag(qt, qfi * 16, 1, l15, lg);
            const uint4 g0 = row_frag(gt, qfi * 16, 0, l15, lg), g1 = row_frag(gt, qfi * 16, 1, l15, lg);
            f32x4 nl4 = *reinterpret_cast<const f32x4*>(st + qfi * 16 + 4 * lg);   // -lse (log2 units) of the 4 queries
            f32x4 nd4 = *reinterpret_cast<const f32x4*>(st + 64 + qfi * 16 + 4 * lg);   // -delta
            asm volatile("" : "+v"(nd4));   // one register quad, read as the C operand of both key fragments' chains
            asm volatile("" : "+v"(nl4));
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              f32x4 z = UNIT ? nl4 : (f32x4){0.f, 0.f, 0.f, 0.f};
              z = mfma16(a0, kf_[f][0], z);
              z = mfma16(a1, kf_[f][1], z);  // S[q][key] (UNIT: already minus lse, in log2 units)
              f32x4 w = mfma16(g0, vf_[f][0], nd4);   // row constants -delta as the initial accumulator
              w = mfma16(g1, vf_[f][1], w);           // dP[q][key] - delta[q]
              f32x4 p, dsv;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float pr = UNIT ? fast_exp2(z[r]) : fast_exp2(z[r] * c2 + nl4[r]);
                if (MASKED && qfi * 16 + lg * 4 + r >= nvalid) pr = 0.f;
                p[r] = pr;
                dsv[r] = pr * w[r];
              }
              ph[qq][f] = pack4(p);
              sh[qq][f] = pack4(dsv);
            }
          } else {
            ph[qq][0] = ph[qq][1] = sh[qq][0] = sh[qq][1] = make_uint2(0u, 0u);
          }
        }
        uint4 pb[2], sb[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          pb[f] = make_uint4(ph[0][f].x, ph[0][f].y, ph[1][f].x, ph[1][f].y);
          sb[f] = make_uint4(sh[0][f].x, sh[0][f].y, sh[1][f].x, sh[1][f].y);
        }
#pragma unroll
        for (int df = 0; df < 4; ++df) {
          const uint4 ga = tr_frag(gt, s2 * 32, df * 16, l15, lg);  // dO^T[d][q slots]
          const uint4 qa = tr_frag(qt, s2 * 32, df * 16, l15, lg);  // Q^T[d][q slots]
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            dv[df][f] = mfma16(ga, pb[f], dv[df][f]);
            dk[df][f] = mfma16(qa, sb[f], dk[df][f]);
          }
        }
      }
    }
    if (t + 2 < nt)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DKV_DMA) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur = cur == 2 ? 0 : cur + 1;
  };
#define NVIT_RUN_TILES(UNIT_)                                                          \
  {                                                                                    \
    for (int t = 0; t + 1 < nt; ++t) tile_body(t, std::false_type{}, UNIT_{});         \
    if (Tq % TKV)                                                                      \
      tile_body(nt - 1, std::true_type{}, UNIT_{});                                    \
    else                                                                               \
      tile_body(nt - 1, std::false_type{}, UNIT_{});                                   \
  }
  if (unit)
    NVIT_RUN_TILES(std::true_type)
  else
    NVIT_RUN_TILES(std::false_type)
#undef NVIT_RUN_TILES
  const float dks = scale / qpre;   // d/d(k_hat) = scale * dS^T q_hat, and the Q tiles hold qpre * q_hat
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int f = 0; f < 2; ++f) dk[i][f] = dk[i][f] * dks;
  char* scr = &lds[0] + wid * 4096;   // (the tile loop ended with a barrier: the ring is free)
  if constexpr (FUSE) {
    if (wave_active)
      store_tile32x64(dv, scr, fu.out_v + ((size_t)b * Tk + k0) * fu.ld + h * 64, (size_t)fu.ld, Tk - k0, lane);
    __builtin_amdgcn_wave_barrier();
    QkEpiLoads el;
    qk_bwd_epilogue_loads(el, kh + (size_t)bh * Tk * D, fu, k0, Tk, H, b, h, lane);
    qk_bwd_epilogue(dk, el, fu, k0, Tk, H, b, h, lane, wid, &lds[0], tile_, (Tk + 127) / 128, (int)threadIdx.x);
  } else {
    if (wave_active) {
      store_tile32x64(dk, scr, dkh + ((size_t)bh * Tk + k0) * D, (size_t)D, Tk - k0, lane);
      __builtin_amdgcn_wave_barrier();
      store_tile32x64(dv, scr, dvh + ((size_t)bh * Tk + k0) * D, (size_t)D, Tk - k0, lane);
    }
  }
}


// ------------------------------------------------------------------------------------------ dK, dV: hand-placed main loop
// Same arithmetic, operand layouts and accumulation order as attn_bwd_dkv_mfma_kernel above (bit-exact against it), for
// the pre-scaled-q case (c2 = 1), with the tile loop written as ONE generated inline-asm statement
// (gen/gen_attn_dkv32_asm.py -> attn_dkv32_asm.inc; structure and numbers: DESIGN.md section 5, round 4).  The
// compiler-built code around it only prepares operands and stores the result: the accumulators come back through LDS
// (ds_write from the accumulation registers), never through compiler-visible registers.
__device__ __forceinline__ unsigned uni32(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ unsigned long long uni64(const void* p) {
  const unsigned long long u = (unsigned long long)(uintptr_t)p;
  return (unsigned long long)uni32((unsigned)u) | ((unsigned long long)uni32((unsigned)(u >> 32)) << 32);
}

// The geometry of the compiler-built kernel (4 waves x 32 keys, 256 registers per wave, two workgroups per CU: one
// workgroup's prologue / epilogue runs under the other's tile loop), the tile loop software-pipelined by the generator.
// (A one-wave-per-SIMD form with 64 keys per wave was built as well - tools/probes/gen_attn_dkv64_asm.py, bit-exact too:
// its tile loop is 1.57x faster per key, but with one wave per SIMD nothing hides a workgroup's prologue and epilogue,
// 14 of its 31 us, and the kernel ends up 7-13 % slower than the compiler-built one.)
#include "attn_dkv32_asm.inc"

template <bool FUSE>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_asm32_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ qh,
                                                                  const bf16* __restrict__ kh, const bf16* __restrict__ vh,
                                                                  const float* __restrict__ delta, float scale, float qpre,
                                                                  bf16* __restrict__ dkh, bf16* __restrict__ dvh, int H,
                                                                  int Tq, int Tk, const float* fu_rn, const float* fu_sqk,
                                                                  float fu_cq, bf16* fu_out, bf16* fu_outv, int fu_ld,
                                                                  float* fu_part, float fu_xs) {
  // (the q/k-normalise operands arrive as scalars and the QkFuse is put together BEHIND the loop: taken by value as one
  //  struct, part of it is parked in LDS from the first instruction on and the address of that slot lives across the loop
  //  statement - one register too many for two waves per SIMD)
  __shared__ __attribute__((aligned(16))) char lds[4 * DKV_SLOT];   // tile ring; then (its first 4 x 16 KiB) the accumulators
  static_assert(4 * DKV_SLOT >= 4 * 16384, "the hand-over area must fit the ring");
  const int tid = threadIdx.x;
  int lane = tid & 63;
  const int wid = (int)uni32((unsigned)(tid >> 6));
  const int l15 = lane & 15, lg = lane >> 4;
  int bh, tile_;
  const int ntile = (Tk + 127) / 128;
  work_of(ntile, bh, tile_);
  const int b = bh / H, h = bh % H;
  const int k0 = tile_ * 128 + wid * 32;
  const int BH = gridDim.x / ntile;
  const int nt = (Tq + TKV - 1) / TKV;
  const int nvalid_last = Tq - (nt - 1) * TKV;
  const unsigned ldg = (unsigned)(H * D * 2);
  const int r8 = lane >> 3, chunk = (lane & 7) ^ r8;
  const unsigned voff_q0 = tile_voff(0, ROWB, lane, wid), voff_g0 = tile_voff(0, ldg, lane, wid);
  unsigned rows_last = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row = (i * 4 + wid) * 8 + r8;
    row = row < nvalid_last ? row : nvalid_last - 1;
    rows_last |= (unsigned)row << (8 * i);
  }
  unsigned kvoff[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    int k = k0 + 16 * f + l15;
    k = k < Tk ? k : Tk - 1;
    kvoff[f] = (unsigned)(k * D + lg * 8) * 2u;
  }
  const unsigned a0 = (unsigned)swz_off(l15, lg), a1 = (unsigned)swz_off(l15, 4 + lg);
  unsigned tro[4];
#pragma unroll
  for (int df = 0; df < 4; ++df)
    tro[df] = (unsigned)(swz_off(4 * lg + (l15 >> 2), 2 * df + ((l15 & 3) >> 1)) + ((l15 & 1) << 3));
  const unsigned ring = lds_addr(&lds[0]);
  const unsigned dump = ring + (unsigned)wid * 16384u + (unsigned)lane * 16u;
  const unsigned long long s_q = uni64(qh + (size_t)bh * Tq * D);
  const unsigned long long s_g = uni64(dout + (size_t)b * Tq * (H * D) + h * D);
  const unsigned long long s_l = uni64(delta + ((size_t)BH + bh) * Tq);
  const unsigned long long s_d = uni64(delta + (size_t)bh * Tq);
  const unsigned long long s_k = uni64(kh + (size_t)bh * Tk * D);
  const unsigned long long s_v = uni64(vh + (size_t)bh * Tk * D);
  const unsigned s_nt = uni32((unsigned)nt), s_ldg = uni32(ldg), s_ring = uni32(ring), s_nvl = uni32((unsigned)nvalid_last);
  const unsigned s_act = uni32(k0 < Tk ? 1u : 0u), s_wofs = uni32((unsigned)wid * 1024u);
  asm volatile(NVIT_ATTN_DKV32_ASM_BODY
               :
               : "s"(s_q), "s"(s_g), "s"(s_l), "s"(s_d), "s"(s_k), "s"(s_v), "s"(s_nt), "s"(s_ldg), "s"(s_ring), "s"(s_nvl),
                 "s"(s_act), "s"(s_wofs), "v"(voff_q0), "v"(voff_g0), "v"(rows_last), "v"((unsigned)chunk * 16u),
                 "v"((unsigned)lane * 4u), "v"(kvoff[0]), "v"(kvoff[1]), "v"(a0), "v"(a1), "v"((unsigned)(lg * 16)), "v"(tro[0]),
                 "v"(tro[1]), "v"(tro[2]), "v"(tro[3]),
                 "v"(dump)
               : NVIT_ATTN_DKV32_ASM_CLOBBERS);
  // lane-dependent values are formed afresh behind the loop: nothing per-lane has to live across it (16 registers are all the
  // compiler has there, and what does not fit would be parked in the accumulation half, i.e. cost a wave per SIMD)
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  lane = tid2 & 63;
  const QkFuse fu{fu_rn, fu_sqk, fu_out, fu_outv, fu_part, fu_cq, fu_xs, fu_ld};
  const bool wave_active = k0 < Tk;
  QkEpiLoads el;
  if constexpr (FUSE) qk_bwd_epilogue_loads(el, kh + (size_t)bh * Tk * D, fu, k0, Tk, H, b, h, lane);
  f32x4 dk[4][2], dv[4][2];
  if (wave_active) {
    const char* mine = &lds[0] + wid * 16384 + lane * 16;
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        dk[df][f] = *reinterpret_cast<const f32x4*>(mine + (df * 2 + f) * 1024);
        dv[df][f] = *reinterpret_cast<const f32x4*>(mine + (8 + df * 2 + f) * 1024);
      }
  }
  __syncthreads();   // every wave holds its accumulators: the LDS becomes store scratch (as in the compiler-built kernel)
  const float dks = scale / qpre;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int f = 0; f < 2; ++f) dk[i][f] = dk[i][f] * dks;
  char* scr = &lds[0] + wid * 4096;
  if constexpr (FUSE) {
    if (wave_active)
      store_tile32x64(dv, scr, fu.out_v + ((size_t)b * Tk + k0) * fu.ld + h * 64, (size_t)fu.ld, Tk - k0, lane);
    __builtin_amdgcn_wave_barrier();
    qk_bwd_epilogue(dk, el, fu, k0, Tk, H, b, h, lane, wid, &lds[0], tile_, ntile, (int)threadIdx.x);
  } else {
    if (wave_active) {
      store_tile32x64(dk, scr, dkh + ((size_t)bh * Tk + k0) * D, (size_t)D, Tk - k0, lane);
      __builtin_amdgcn_wave_barrier();
      store_tile32x64(dv, scr, dvh + ((size_t)bh * Tk + k0) * D, (size_t)D, Tk - k0, lane);
    }
  }
}

// -1: read NVIT_ATTN_DKV_ASM on first use.  0: compiler-built kernel; 1 (default): hand-placed loop
int g_attn_dkv_asm = -1;
int dkv_asm_mode(float scale, float qpre) {
  if (g_attn_dkv_asm < 0) {
    const char* e = getenv("NVIT_ATTN_DKV_ASM");
    g_attn_dkv_asm = (e && e[0] == '0') ? 0 : 1;
  }
  const float c2 = scale * LOG2E / qpre;
  return fabsf(c2 - 1.0f) < 1e-6f ? g_attn_dkv_asm : 0;   // the hand-placed loops assume the pre-scaled q (c2 = 1)
}

}  // namespace

int nvit_attn_fwd_mfma(const void* qh, const void* kh, const void* vh, float scale, float qpre, const float* sqk,
                       float c_q, void* o, float* lse, int B, int H, int Tq, int Tk, int d, hipStream_t s) {
  NVIT_REQUIRE(d == 64, "attn_fwd: the MFMA kernel supports head dim 64 only (got %d)", d);
  NVIT_REQUIRE(qpre > 0.f, "attn_fwd: the q pre-scale must be positive (got %g)", (double)qpre);
  dim3 grid((unsigned)(cdiv(Tq, 128) * B * H));
  hipLaunchKernelGGL(attn_fwd_mfma_kernel, grid, dim3(256), 0, s, (const bf16*)qh, (const bf16*)kh, (const bf16*)vh,
                     scale, qpre, sqk, c_q, (bf16*)o, lse, H, Tq, Tk);
  NVIT_CHECK_LAUNCH("attn_fwd_mfma");
  return NVIT_OK;
}

int nvit_attn_bwd_mfma(const void* dout, const void* qh, const void* kh, const void* vh, const void* o, const float* lse,
                       float* delta, float scale, void* dqh, void* dkh, void* dvh, int B, int H, int Tq, int Tk,
                       int d, hipStream_t s) {
  NVIT_REQUIRE(d == 64, "attn_bwd: the MFMA kernel supports head dim 64 only (got %d)", d);
  NVIT_REQUIRE(o != nullptr && delta != nullptr, "attn_bwd: the attention output and the [2,B,H,Tq] side buffer are required");
  dim3 gq((unsigned)(cdiv(Tq, 128) * B * H)), gk((unsigned)(cdiv(Tk, 128) * B * H));
  QkFuse none{};
  none.xs = 1.0f;
  hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<false>, gq, dim3(256), 0, s, (const bf16*)dout, (const bf16*)qh,
                     (const bf16*)kh, (const bf16*)vh, lse, (const bf16*)o, delta, scale, 1.0f, (bf16*)dqh, H, Tq, Tk,
                     none);
  NVIT_CHECK_LAUNCH("attn_bwd_dq_mfma");
  hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<false>, gk, dim3(256), 0, s, (const bf16*)dout, (const bf16*)qh,
                     (const bf16*)kh, (const bf16*)vh, lse, delta, scale, 1.0f, (bf16*)dkh, (bf16*)dvh, H, Tq, Tk, none);
  NVIT_CHECK_LAUNCH("attn_bwd_dkv_mfma");
  return NVIT_OK;
}

// (experiments / tests) 1: hand-placed dK/dV main loop where it applies (default), 0: the compiler-built kernel
extern "C" int nvit_set_attn_dkv_asm(int mode) {
  g_attn_dkv_asm = mode > 0 ? 1 : 0;
  return NVIT_OK;
}

// attention backward with the q/k-normalise backward fused into the epilogues: writes token-major dq/dk/dv
// (row stride ld) and the partial sums part_q [B*ceil(Tq/128), C], part_k [B*ceil(Tk/128), C].
int nvit_attn_bwd_mfma_fused(const void* dout, const void* qh, const void* kh, const void* vh, const void* o,
                             const float* lse, float* delta, float scale, const float* rq, const float* rk, const float* sqk,
                             float c_q, float qpre, void* dq, int ldq, void* dk, void* dv, int ldkv, float* part_q,
                             float* part_k, int B, int H, int Tq, int Tk, int d, hipStream_t s) {
  NVIT_REQUIRE(d == 64, "attn_bwd: the MFMA kernel supports head dim 64 only (got %d)", d);
  NVIT_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0, "attn_bwd: leading dims must be multiples of 4");
  NVIT_REQUIRE(o != nullptr && delta != nullptr, "attn_bwd: the attention output and the [2,B,H,Tq] side buffer are required");
  dim3 gq((unsigned)(cdiv(Tq, 128) * B * H)), gk((unsigned)(cdiv(Tk, 128) * B * H));
  NVIT_REQUIRE(qpre > 0.f, "attn_bwd: the q pre-scale must be positive (got %g)", (double)qpre);
  QkFuse fq{rq, sqk, (bf16*)dq, nullptr, part_q, c_q, 1.0f / qpre, ldq};
  QkFuse fk{rk, sqk, (bf16*)dk, (bf16*)dv, part_k, c_q, 1.0f, ldkv};
  hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<true>, gq, dim3(256), 0, s, (const bf16*)dout, (const bf16*)qh,
                     (const bf16*)kh, (const bf16*)vh, lse, (const bf16*)o, delta, scale, qpre, (bf16*)nullptr, H, Tq, Tk,
                     fq);
  NVIT_CHECK_LAUNCH("attn_bwd_dq_mfma_fused");
  const int mode = dkv_asm_mode(scale, qpre);
  if (mode == 1)
    hipLaunchKernelGGL(attn_bwd_dkv_asm32_kernel<true>, gk, dim3(256), 0, s, (const bf16*)dout, (const bf16*)qh,
                       (const bf16*)kh, (const bf16*)vh, delta, scale, qpre, (bf16*)nullptr, (bf16*)nullptr, H, Tq, Tk, fk.rn,
                       fk.sqk, fk.c_q, fk.out, fk.out_v, fk.ld, fk.part, fk.xs);
  else
    hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<true>, gk, dim3(256), 0, s, (const bf16*)dout, (const bf16*)qh,
                       (const bf16*)kh, (const bf16*)vh, lse, delta, scale, qpre, (bf16*)nullptr, (bf16*)nullptr, H, Tq,
                       Tk, fk);
  NVIT_CHECK_LAUNCH("attn_bwd_dkv_mfma_fused");
  return NVIT_OK;
}
