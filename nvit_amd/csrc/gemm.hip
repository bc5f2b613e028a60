// MFMA GEMMs for gfx950.
//
//   gemm_nt : C[M,N] = A[M,K] * B[N,K]^T   (forward linears and, with the transposed weight
//             shadow as B, every data-gradient GEMM)
//   gemm_tn : G[N,K] = sum_m A[m,N] * B[m,K] (weight gradients; both operands have the
//             reduction index as their row index, so fragments come from transposed LDS reads)
//
// Tile 128x128, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles of 16x16.
// bf16: v_mfma_f32_16x16x32_bf16; fp32: v_mfma_f32_16x16x4_f32 (exact f32, used for the
// 1e-5 parity mode).  Both element types use the same 128-byte LDS row geometry
// (64 bf16 / 32 fp32 of K per stage), XOR-swizzled in 16-byte chunks so that every
// ds_read_b128 fragment read is bank-conflict free.  Global->LDS staging goes through
// registers: the loads for stage t+1 are issued before the MFMAs of stage t and written to
// the other LDS buffer after them (one barrier per stage).
//
// Operand orientation: the MFMA "A" operand is fed from the B matrix (rows = n) and the "B"
// operand from the A matrix (cols = m), so each lane ends up with 4 CONSECUTIVE n of one
// row m: epilogue vectors (bias, column scales) are float4 loads and stores are 8/16 bytes.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;  // bytes of K per LDS row per stage

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

template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_kernel(NtArgs g) {
  __shared__ __attribute__((aligned(16))) char lds[2][2][BM * ROWB];  // [buf][A|B] = 64 KiB
  constexpr int EPC = 16 / sizeof(T);   // elements per 16-byte chunk
  constexpr int BK = ROWB / sizeof(T);  // K elements per stage
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int tm = blockIdx.x / g.tiles_n, tn = blockIdx.x % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // staging map: thread -> rows (tid>>3)+32*i, global chunk tid&7, LDS slot gc ^ (row&7)
  const int srow = tid >> 3, gc = tid & 7;
  const int sslot = gc ^ (srow & 7);
  const char* ap[4];
  const char* bp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ra = m0 + srow + 32 * i;
    ra = ra < g.M ? ra : g.M - 1;
    int rb = n0 + srow + 32 * i;
    rb = rb < g.N ? rb : g.N - 1;
    ap[i] = g.A + ((size_t)ra * g.lda + (size_t)gc * EPC) * sizeof(T);
    bp[i] = g.B + ((size_t)rb * g.ldb + (size_t)gc * EPC) * sizeof(T);
  }
  const int soff = srow * ROWB + sslot * 16;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  const int nt = g.K / BK;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *reinterpret_cast<const uint4*>(ap[i]);
    rb[i] = *reinterpret_cast<const uint4*>(bp[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<uint4*>(&lds[0][0][soff + i * 32 * ROWB]) = ra[i];
    *reinterpret_cast<uint4*>(&lds[0][1][soff + i * 32 * ROWB]) = rb[i];
  }
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    {
      // prefetch stage t+1 into registers (the last iteration re-loads its own stage: no branch,
      // so the staging registers stay in VGPRs and the loads stay in flight under the MFMAs)
      const int tnext = (t + 1 < nt) ? (t + 1) : t;
      const size_t ko = (size_t)tnext * BK * sizeof(T);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const uint4*>(ap[i] + ko);
        rb[i] = *reinterpret_cast<const uint4*>(bp[i] + ko);
      }
    }
    const char* la = &lds[cur][0][0];
    const char* lb = &lds[cur][1][0];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = kk * 4 + lg;
      uint4 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wr * 64 + i * 16 + l15;
        fa[i] = *reinterpret_cast<const uint4*>(la + r * ROWB + ((c ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wc * 64 + j * 16 + l15;
        fb[j] = *reinterpret_cast<const uint4*>(lb + r * ROWB + ((c ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<T>::run(fb[j], fa[i], acc[i][j]);
    }
    {
      char* wa = &lds[cur ^ 1][0][0];
      char* wb = &lds[cur ^ 1][1][0];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<uint4*>(wa + soff + i * 32 * ROWB) = ra[i];
        *reinterpret_cast<uint4*>(wb + soff + i * 32 * ROWB) = rb[i];
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: acc[i][j][r] = C[m0 + wr*64 + 16i + l15][n0 + wc*64 + 16j + 4*lg + r]
  const bool vec_ok = ((g.N & 3) == 0) && ((g.ldc & 3) == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr * 64 + i * 16 + l15;
    if (m >= g.M) continue;
    const float* radd = g.rowadd ? g.rowadd + (size_t)(m % g.rowadd_period) * g.N : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nb = n0 + wc * 64 + j * 16 + 4 * lg;
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

// ------------------------------------------------------------------------------------------
// TN (weight gradient).  Output tile: 128 (n) x 128 (k'), reduction over rows m in stages of
// RB rows (64 for bf16, 32 for fp32).  LDS tiles are stored [m][col] exactly as they sit in
// memory (coalesced 16-byte loads); bf16 fragments are fetched with ds_read_b64_tr_b16
// (hardware 4x16 transpose), fp32 fragments with ds_read_b32.
// Result orientation: MFMA rows = k' (4 consecutive k' per lane), cols = n.
struct TnArgs {
  const char* A;  // [Mred, N]
  const char* B;  // [Mred, K]
  float* ws;      // [splits, N, K]
  int Mred, N, K;
  int lda, ldb;
  int rows_per_split;
  int tiles_k;
};

template <typename T>
struct TnGeom;
template <>
struct TnGeom<bf16> {
  static constexpr int RB = 64;          // reduction rows per stage
  static constexpr int ROW_BYTES = 256;  // 128 cols * 2 B
  static constexpr int CHUNKS = 16;      // 16-byte chunks per row
};
template <>
struct TnGeom<float> {
  static constexpr int RB = 32;
  static constexpr int ROW_BYTES = 512;
  static constexpr int CHUNKS = 32;
};

// swizzle of the 16-byte chunk index by the reduction row (bf16 tiles): rows that one
// transposed read touches together (m = 8g+q, g in {0,1} per 32-lane half, q in 0..3) land on
// 8 distinct chunk pairs of the 256-byte bank row.
__device__ __forceinline__ int tn_swz(int m) { return (((m & 3) | (((m >> 3) & 1) << 2)) << 1); }

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TnArgs g) {
  using G = TnGeom<T>;
  constexpr int RB = G::RB, ROW_BYTES = G::ROW_BYTES, CHUNKS = G::CHUNKS;
  constexpr int EPC = 16 / sizeof(T);
  constexpr int TILE_BYTES = RB * ROW_BYTES;                           // 16 KiB
  constexpr int LPT = TILE_BYTES / 16 / 256;                           // 16-byte loads per thread per tile = 4
  __shared__ __attribute__((aligned(16))) char lds[2][2][TILE_BYTES];  // 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;  // wr: k' half (MFMA rows), wc: n half (MFMA cols)
  const int l15 = lane & 15, lg = lane >> 4;
  const int tn = blockIdx.x / g.tiles_k, tk = blockIdx.x % g.tiles_k;
  const int n0 = tn * BN, k0 = tk * BM;
  const int split = blockIdx.y;
  const int mbeg = split * g.rows_per_split;
  int mend = mbeg + g.rows_per_split;
  if (mend > g.Mred) mend = g.Mred;
  const int nt = (mend - mbeg + RB - 1) / RB;

  // staging: chunk q = tid + 256*i : row = q / CHUNKS, chunk = q % CHUNKS.  Out-of-range rows /
  // columns are loaded from a clamped (valid) address and zeroed by a select, so there is no
  // branch around any load.
  int soffs[LPT];
  size_t aoff[LPT], boff[LPT];
  int srow[LPT];
  bool a_ok[LPT], b_ok[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int q = tid + 256 * i;
    srow[i] = q / CHUNKS;
    const int ch = q % CHUNKS;
    a_ok[i] = (n0 + ch * EPC) < g.N;  // N, K are multiples of EPC (checked on host)
    b_ok[i] = (k0 + ch * EPC) < g.K;
    aoff[i] = (size_t)(a_ok[i] ? n0 + ch * EPC : 0) * sizeof(T);
    boff[i] = (size_t)(b_ok[i] ? k0 + ch * EPC : 0) * sizeof(T);
    if constexpr (sizeof(T) == 2)
      soffs[i] = srow[i] * ROW_BYTES + ((ch ^ tn_swz(srow[i])) << 4);
    else
      soffs[i] = srow[i] * ROW_BYTES + (ch << 4);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 ra[LPT], rb[LPT];
#define TN_GLOAD(t_)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < LPT; ++i) {                                                    \
    const int m = mbeg + (t_) * RB + srow[i];                                                          \
    const bool mok = m < mend;                                                                         \
    const size_t mr = (size_t)(mok ? m : (g.Mred - 1));                                                \
    uint4 va = *reinterpret_cast<const uint4*>(g.A + mr * g.lda * sizeof(T) + aoff[i]);                \
    uint4 vb = *reinterpret_cast<const uint4*>(g.B + mr * g.ldb * sizeof(T) + boff[i]);                \
    const unsigned ma = (mok && a_ok[i]) ? 0xFFFFFFFFu : 0u, mb = (mok && b_ok[i]) ? 0xFFFFFFFFu : 0u; \
    va.x &= ma; va.y &= ma; va.z &= ma; va.w &= ma;                                                    \
    vb.x &= mb; vb.y &= mb; vb.z &= mb; vb.w &= mb;                                                    \
    ra[i] = va;                                                                                        \
    rb[i] = vb;                                                                                        \
  }
#define TN_LSTORE(buf_)                                                                                \
  _Pragma("unroll") for (int i = 0; i < LPT; ++i) {                                                    \
    *reinterpret_cast<uint4*>(&lds[buf_][0][soffs[i]]) = ra[i];                                        \
    *reinterpret_cast<uint4*>(&lds[buf_][1][soffs[i]]) = rb[i];                                        \
  }

  TN_GLOAD(0)
  TN_LSTORE(0)
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    {
      const int tnext = (t + 1 < nt) ? (t + 1) : t;
      TN_GLOAD(tnext)
    }
    const char* la = &lds[cur][0][0];  // A tile: [m][n]   -> MFMA B operand (cols = n)
    const char* lb = &lds[cur][1][0];  // B tile: [m][k']  -> MFMA A operand (rows = k')
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < RB / 32; ++ks) {
        // lane (group lg, i = l15 = 4q+p): address of row m = 32ks + 8lg + q (+4), cols c0 + 4p..4p+3
        const int q = l15 >> 2, p = l15 & 3;
        const int mrow = ks * 32 + lg * 8 + q;
        uint4 fa[4], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c0 = wc * 64 + j * 16;  // n offset inside tile
          const int ch = (c0 >> 3) + (p >> 1);
          const int o0 = mrow * ROW_BYTES + ((ch ^ tn_swz(mrow)) << 4) + ((p & 1) << 3);
          const int o1 = (mrow + 4) * ROW_BYTES + ((ch ^ tn_swz(mrow + 4)) << 4) + ((p & 1) << 3);
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(la + o0));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(la + o1));
          uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          fa[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c0 = wr * 64 + i * 16;  // k' offset inside tile
          const int ch = (c0 >> 3) + (p >> 1);
          const int o0 = mrow * ROW_BYTES + ((ch ^ tn_swz(mrow)) << 4) + ((p & 1) << 3);
          const int o1 = (mrow + 4) * ROW_BYTES + ((ch ^ tn_swz(mrow + 4)) << 4) + ((p & 1) << 3);
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(lb + o0));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(lb + o1));
          uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          fb[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) Mma<bf16>::run(fb[i], fa[j], acc[i][j]);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < RB / 4; ++ks) {
        const int mrow = ks * 4 + lg;
        float fa[4], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          fa[j] = *reinterpret_cast<const float*>(la + mrow * ROW_BYTES + (wc * 64 + j * 16 + l15) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          fb[i] = *reinterpret_cast<const float*>(lb + mrow * ROW_BYTES + (wr * 64 + i * 16 + l15) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[i], fa[j], acc[i][j], 0, 0, 0);
      }
    }
    TN_LSTORE(cur ^ 1)
    __syncthreads();
    cur ^= 1;
  }
#undef TN_GLOAD
#undef TN_LSTORE

  // acc[i][j][r] = G[n = n0 + wc*64 + 16j + l15][k' = k0 + wr*64 + 16i + 4lg + r]
  float* out = g.ws + (size_t)split * g.N * g.K;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wc * 64 + j * 16 + l15;
    if (n >= g.N) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kb = k0 + wr * 64 + i * 16 + 4 * lg;
      if (kb >= g.K) continue;  // K % 4 == 0
      *reinterpret_cast<f32x4*>(out + (size_t)n * g.K + kb) = acc[i][j];
    }
  }
}

__device__ __forceinline__ int perm_row(int perm, int s, int F) {
  if (perm == 0) return s;
  const int q = s >> 5, w = s & 31;
  return w < 16 ? q * 16 + w : F + q * 16 + (w - 16);
}

__global__ void slab_reduce_kernel(const float* ws, int splits, int N, int K, float* G, int ldg, int perm,
                                   int accumulate) {
  const int kq = K >> 2;
  const long long total = (long long)N * kq;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx / kq), k = (int)(idx % kq) * 4;
    f32x4 s = *reinterpret_cast<const f32x4*>(ws + (size_t)n * K + k);
    for (int p = 1; p < splits; ++p) s += *reinterpret_cast<const f32x4*>(ws + ((size_t)p * N + n) * K + k);
    float* dst = G + (size_t)perm_row(perm, n, N / 2) * ldg + k;
    if (accumulate) s += *reinterpret_cast<const f32x4*>(dst);
    *reinterpret_cast<f32x4*>(dst) = s;
  }
}

}  // namespace

extern "C" int nvit_gemm_nt(int dt, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int out_dt,
                            int M, int N, int K, const float* bias, const float* colscale, const float* rowadd,
                            int rowadd_period, int accumulate, void* stream) {
  NVIT_REQUIRE(dt == NVIT_F32 || dt == NVIT_BF16, "gemm_nt: bad dt %d", dt);
  NVIT_REQUIRE(out_dt == NVIT_F32 || out_dt == NVIT_BF16, "gemm_nt: bad out_dt %d", out_dt);
  const int es = dt == NVIT_F32 ? 4 : 2;
  const int bk = ROWB / es;
  NVIT_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem %dx%dx%d", M, N, K);
  NVIT_REQUIRE(K % bk == 0, "gemm_nt: K=%d must be a multiple of %d", K, bk);
  NVIT_REQUIRE((lda * es) % 16 == 0 && (ldb * es) % 16 == 0, "gemm_nt: lda/ldb must be 16-byte multiples");
  NVIT_REQUIRE(lda >= K && ldb >= K && ldc >= N, "gemm_nt: leading dims too small");
  NVIT_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0,
               "gemm_nt: pointers must be 16-byte aligned");
  NVIT_REQUIRE(!rowadd || rowadd_period > 0, "gemm_nt: rowadd needs a period");
  NtArgs g;
  g.A = (const char*)A;
  g.B = (const char*)B;
  g.C = C;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = ldc;
  g.bias = bias;
  g.colscale = colscale;
  g.rowadd = rowadd;
  g.rowadd_period = rowadd_period;
  g.accumulate = accumulate;
  g.out_dt = out_dt;
  g.tiles_n = cdiv(N, BN);
  const long long blocks = (long long)cdiv(M, BM) * g.tiles_n;
  NVIT_REQUIRE(blocks < (1ll << 31), "gemm_nt: grid too large");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_GEMM_NT, 2.0 * M * N * K, 0.0, s);
  if (dt == NVIT_BF16)
    hipLaunchKernelGGL(gemm_nt_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL(gemm_nt_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, g);
  NVIT_CHECK_LAUNCH("gemm_nt");
  return NVIT_OK;
}

extern "C" int nvit_gemm_tn(int dt, const void* A, int lda, const void* B, int ldb, float* G, int ldg, int Mred,
                            int N, int K, int splits, float* ws, int64_t ws_bytes, int perm, int accumulate,
                            void* stream) {
  NVIT_REQUIRE(dt == NVIT_F32 || dt == NVIT_BF16, "gemm_tn: bad dt %d", dt);
  const int es = dt == NVIT_F32 ? 4 : 2;
  const int epc = 16 / es;
  const int rb = dt == NVIT_F32 ? 32 : 64;
  NVIT_REQUIRE(Mred > 0 && N > 0 && K > 0 && splits > 0, "gemm_tn: empty problem");
  NVIT_REQUIRE(N % epc == 0 && K % epc == 0, "gemm_tn: N=%d and K=%d must be multiples of %d", N, K, epc);
  NVIT_REQUIRE((lda * es) % 16 == 0 && (ldb * es) % 16 == 0 && ldg % 4 == 0, "gemm_tn: leading dims alignment");
  NVIT_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)G & 15) == 0 &&
                   ((uintptr_t)ws & 15) == 0,
               "gemm_tn: pointers must be 16-byte aligned");
  NVIT_REQUIRE(ws_bytes >= (int64_t)splits * N * K * 4, "gemm_tn: workspace too small");
  NVIT_REQUIRE(perm == 0 || (perm == 1 && N % 32 == 0), "gemm_tn: perm=1 needs N %% 32 == 0");
  TnArgs g;
  g.A = (const char*)A;
  g.B = (const char*)B;
  g.ws = ws;
  g.Mred = Mred;
  g.N = N;
  g.K = K;
  g.lda = lda;
  g.ldb = ldb;
  int rps = cdiv(Mred, splits);
  rps = cdiv(rps, rb) * rb;
  g.rows_per_split = rps;
  g.tiles_k = cdiv(K, BM);
  dim3 grid((unsigned)(cdiv(N, BN) * g.tiles_k), (unsigned)splits);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(NVIT_KID_GEMM_TN, 2.0 * Mred * (double)N * K, 0.0, s);
  if (dt == NVIT_BF16)
    hipLaunchKernelGGL(gemm_tn_kernel<bf16>, grid, dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, dim3(256), 0, s, g);
  NVIT_CHECK_LAUNCH("gemm_tn");
  const long long total = (long long)N * (K / 4);
  int rblocks = cdiv(total, 256);
  if (rblocks > 4096) rblocks = 4096;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(rblocks), dim3(256), 0, s, ws, splits, N, K, G, ldg, perm,
                     accumulate);
  NVIT_CHECK_LAUNCH("slab_reduce");
  return NVIT_OK;
}
