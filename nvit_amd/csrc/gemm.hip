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
// ds_read_b128 fragment read is bank-conflict free.  gemm_nt stages global->LDS with LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write), one stage ahead, one barrier per
// stage (issued from inline asm so hipcc does not serialise it), in both kernels.
//
// Operand orientation: the MFMA "A" operand is fed from the B matrix (rows = n) and the "B"
// operand from the A matrix (cols = m), so each lane ends up with 4 CONSECUTIVE n of one
// row m: epilogue vectors (bias, column scales) are float4 loads and stores are 8/16 bytes.
#include <stdlib.h>

#include "gemm_common.h"

int nvit_gemm_nt_persistent_launch(int dt, const NtArgs& g, int tile_n, hipStream_t s);
int nvit_gemm_nt_fused_launch(const NtArgs& g, int epi, hipStream_t s);

int nvit_gemm_tn_persistent_launch(int dt, const void* A, int lda, const void* B, int ldb, float* ws,
                                   const float* zeros, int Mred, int N, int K, int splits, hipStream_t s);

namespace {

constexpr int BM = 128, BN = 128;

template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_kernel(NtArgs g) {
  __shared__ __attribute__((aligned(16))) char lds[2][2][BM * ROWB];  // [buf][A|B] = 64 KiB
  constexpr int EPC = 16 / sizeof(T);   // elements per 16-byte chunk
  constexpr int BK = ROWB / sizeof(T);  // K elements per stage
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  // Tile order.  Workgroups are dealt round-robin to the 8 XCDs (private 4 MiB L2 each), so block
  // ids are first re-dealt such that every XCD owns ONE contiguous range of the tile sequence
  // (bijective for any grid size), and the sequence itself walks 8(m) x tiles_n super-columns in
  // m-fastest order: the 64 workgroups resident on an XCD then share 8 A panels and 8 B panels
  // through L2 instead of re-fetching them from HBM/Infinity Cache.
  int tm, tn;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    constexpr int GM = 8;
    const int per_group = GM * g.tiles_n;
    const int group = pid / per_group, first_m = group * GM;
    const int tiles_m = nwg / g.tiles_n;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = pid - group * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // staging: LDS-DMA (global_load_lds_dwordx4).  One wave-instruction fills 1 KiB = 8 LDS rows:
  // lane L lands at row 8*grp + (L>>3), 16-byte slot L&7, so to realise the XOR swizzle the lane
  // FETCHES global chunk (L&7) ^ (row&7) of its row (the 128-byte line stays fully coalesced).
  // Wave w issues groups grp = 4*i + w, i = 0..3, for each operand.
  const int srow = lane >> 3;
  const int gc = (lane & 7) ^ srow;
  const char* ap[4];
  const char* bp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ra = m0 + (i * 4 + wid) * 8 + srow;
    ra = ra < g.M ? ra : g.M - 1;
    int rb = n0 + (i * 4 + wid) * 8 + srow;
    rb = rb < g.N ? rb : g.N - 1;
    ap[i] = g.A + ((size_t)ra * g.lda + (size_t)gc * EPC) * sizeof(T);
    bp[i] = g.B + ((size_t)rb * g.ldb + (size_t)gc * EPC) * sizeof(T);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = g.K / BK;
  // The DMA is issued from inline asm so that hipcc does not serialise it: with the builtin the
  // compiler waits vmcnt(0) before the very next ds_read (it cannot see that the DMA targets the
  // OTHER buffer).  We wait ourselves, once per stage, right before the barrier.
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)&lds[0][0][0]);
  const unsigned wave_off = (unsigned)__builtin_amdgcn_readfirstlane(wid * 8 * ROWB);
#define NT_STAGE(buf_, t_)                                                                              \
  {                                                                                                     \
    const size_t ko = (size_t)(t_) * BK * sizeof(T);                                                    \
    const unsigned bo = lds_base + wave_off + (unsigned)(buf_) * (2 * BM * ROWB);                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
      glds16(ap[i] + ko, bo + i * 32 * ROWB);                                                           \
      glds16(bp[i] + ko, bo + BM * ROWB + i * 32 * ROWB);                                               \
    }                                                                                                   \
  }
  NT_STAGE(0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    // stage t+1 streams into the other buffer while stage t is consumed (the last iteration
    // re-fetches its own stage into the idle buffer: branch-free, never read)
    NT_STAGE(cur ^ 1, (t + 1 < nt) ? (t + 1) : t)
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

#undef NT_STAGE
  nt_store_tile(g, acc, m0 + wr * 64, n0 + wc * 64, l15, lg);
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
  const float* zeros;  // >= 16 bytes of zeros (tail rows / columns are fetched from here)
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

  // staging by LDS-DMA: one wave-instruction = 1 KiB = RPI rows of the [m][col] tile; lane L lands at
  // row L / CHUNKS, slot L % CHUNKS and therefore fetches global chunk slot ^ swz(row) (bf16).
  // Rows past the split's end and columns past N / K are fetched from a 16-byte block of zeros.
  constexpr int RPI = 1024 / ROW_BYTES;  // rows per wave-instruction: 4 (bf16) / 2 (fp32)
  const int lrow = lane / CHUNKS, lslot = lane % CHUNKS;
  int srow[LPT];
  const char* abase[LPT];
  const char* bbase[LPT];
  bool a_ok[LPT], b_ok[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    srow[i] = (i * 4 + wid) * RPI + lrow;
    int ch;
    if constexpr (sizeof(T) == 2)
      ch = lslot ^ tn_swz(srow[i]);
    else
      ch = lslot;
    a_ok[i] = (n0 + ch * EPC) < g.N;  // N, K are multiples of EPC (checked on host)
    b_ok[i] = (k0 + ch * EPC) < g.K;
    abase[i] = g.A + ((size_t)(mbeg + srow[i]) * g.lda + (a_ok[i] ? n0 + ch * EPC : 0)) * sizeof(T);
    bbase[i] = g.B + ((size_t)(mbeg + srow[i]) * g.ldb + (b_ok[i] ? k0 + ch * EPC : 0)) * sizeof(T);
  }
  const size_t astep = (size_t)RB * g.lda * sizeof(T), bstep = (size_t)RB * g.ldb * sizeof(T);
  const char* zsrc = reinterpret_cast<const char*>(g.zeros);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)&lds[0][0][0]);
  const unsigned wave_off = (unsigned)__builtin_amdgcn_readfirstlane(wid * 1024);
#define TN_STAGE(buf_, t_)                                                                              \
  {                                                                                                     \
    const unsigned bo = lds_base + wave_off + (unsigned)(buf_) * (2 * TILE_BYTES);                      \
    _Pragma("unroll") for (int i = 0; i < LPT; ++i) {                                                   \
      const bool mok = (mbeg + (t_) * RB + srow[i]) < mend;                                             \
      const char* pa = (mok && a_ok[i]) ? abase[i] + (size_t)(t_) * astep : zsrc;                       \
      const char* pb = (mok && b_ok[i]) ? bbase[i] + (size_t)(t_) * bstep : zsrc;                       \
      glds16(pa, bo + i * 4096);                                                                        \
      glds16(pb, bo + TILE_BYTES + i * 4096);                                                           \
    }                                                                                                   \
  }

  TN_STAGE(0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    TN_STAGE(cur ^ 1, (t + 1 < nt) ? (t + 1) : t)
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
#undef TN_STAGE

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

// Kernel selection (tests and experiments; -1 = read NVIT_GEMM_NT_IMPL / NVIT_GEMM_TN_IMPL on first use).
// nt: 0 = always the 128x128 kernel, 1 = persistent kernels for large problems (default), 2 = persistent kernels
// whatever the tile count.  tn: 0 = always the 128x128 kernel, 1 = persistent kernel for eligible shapes (default).
static int g_nt_impl = -1, g_tn_impl = -1;
extern "C" int nvit_set_gemm_impl(int nt_impl, int tn_impl) {
  NVIT_REQUIRE(nt_impl >= -1 && nt_impl <= 2 && tn_impl >= -1 && tn_impl <= 1, "set_gemm_impl: nt in -1..2, tn in -1..1");
  g_nt_impl = nt_impl;
  g_tn_impl = tn_impl;
  return NVIT_OK;
}

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
  NtArgs g{};
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
  // algorithmic bytes: A and B read once, C written once (read as well when accumulating)
  const double nt_bytes = (double)es * ((double)M * K + (double)N * K) +
                          (double)(out_dt == NVIT_F32 ? 4 : 2) * M * N * (accumulate ? 2.0 : 1.0);
  ProfScope ps(dt == NVIT_F32 ? NVIT_KID_GEMM_F32 : NVIT_KID_GEMM_NT, 2.0 * M * N * K, nt_bytes, s);
  {
    // large problems: persistent kernels (gemm_p.hip), 256x256 tiles when N allows, else 256x128.
    // NVIT_GEMM_NT_IMPL=0 forces the 128x128 kernel, NVIT_GEMM_NT_TILE=128|256 forces a tile width.
    static int force_tile = -1;
    if (g_nt_impl < 0) {
      const char* e = getenv("NVIT_GEMM_NT_IMPL");
      g_nt_impl = e ? atoi(e) : 1;
    }
    if (force_tile < 0) {
      const char* t = getenv("NVIT_GEMM_NT_TILE");
      force_tile = t ? atoi(t) : 0;
    }
    const int impl = g_nt_impl;
    if (impl >= 1) {  // 2: persistent kernel whatever the tile count (experiments)
      const long long t256 = (long long)cdiv(M, 256) * cdiv(N, 256), t128 = (long long)cdiv(M, 256) * cdiv(N, 128);
      int tile = 0;
      if (N % 256 == 0 && (t256 >= 512 || impl == 2)) tile = 256;
      else if (t128 >= 512 || impl == 2) tile = 128;
      if (force_tile && tile) tile = force_tile;
      if (tile) return nvit_gemm_nt_persistent_launch(dt, g, tile, s);
    }
  }
  if (dt == NVIT_BF16)
    hipLaunchKernelGGL(gemm_nt_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL(gemm_nt_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, g);
  NVIT_CHECK_LAUNCH("gemm_nt");
  return NVIT_OK;
}

// Tail rows / columns of the TN kernels are fetched from a block of zeros.  It is a static device array owned by the
// library (zero-initialised at module load, never written), not a memset of caller memory per call: a memset node
// inside a captured hipGraph did not reliably re-zero the block on replay, and 55 tiny memsets per step were 0.26 ms.
__device__ __attribute__((aligned(256))) float nvit_tn_zero_block[64];
static const float* tn_zero_block() {
  static const float* cache[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!cache[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(nvit_tn_zero_block)) != hipSuccess) return nullptr;
    cache[dev] = (const float*)p;
  }
  return cache[dev];
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
  NVIT_REQUIRE(ws_bytes >= (int64_t)splits * N * K * 4 + 256, "gemm_tn: workspace too small (need splits*N*K*4 + 256 B)");
  NVIT_REQUIRE(perm == 0 || (perm == 1 && N % 32 == 0), "gemm_tn: perm=1 needs N %% 32 == 0");
  TnArgs g;
  g.A = (const char*)A;
  g.B = (const char*)B;
  g.ws = ws;
  g.zeros = tn_zero_block();
  if (!g.zeros) NVIT_FAIL(NVIT_EINVAL, "gemm_tn: cannot resolve the device zero block");
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
  // algorithmic bytes: both operands read once, the fp32 gradient written once (read as well when accumulating)
  const double tn_bytes = (double)es * Mred * ((double)N + K) + 4.0 * N * K * (accumulate ? 2.0 : 1.0);
  ProfScope ps(dt == NVIT_F32 ? NVIT_KID_GEMM_F32 : NVIT_KID_GEMM_TN, 2.0 * Mred * (double)N * K, tn_bytes, s);
  bool done = false;
  {
    // big 256-aligned weight shapes: persistent 256x256 kernel (gemm_tn_p.hip); NVIT_GEMM_TN_IMPL=0 disables
    if (g_tn_impl < 0) {
      const char* e = getenv("NVIT_GEMM_TN_IMPL");
      g_tn_impl = e ? atoi(e) : 1;
    }
    if (g_tn_impl == 1 && Mred >= 4096) {
      const int rc = nvit_gemm_tn_persistent_launch(dt, A, lda, B, ldb, ws, g.zeros, Mred, N, K,
                                                    splits, s);
      if (rc > 0) return rc;
      done = rc == NVIT_OK;
    }
  }
  if (!done) {
    if (dt == NVIT_BF16)
      hipLaunchKernelGGL(gemm_tn_kernel<bf16>, grid, dim3(256), 0, s, g);
    else
      hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, dim3(256), 0, s, g);
    NVIT_CHECK_LAUNCH("gemm_tn");
  }
  const long long total = (long long)N * (K / 4);
  int rblocks = cdiv(total, 256);
  if (rblocks > 4096) rblocks = 4096;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(rblocks), dim3(256), 0, s, ws, splits, N, K, G, ldg, perm,
                     accumulate);
  NVIT_CHECK_LAUNCH("slab_reduce");
  return NVIT_OK;
}

// ---- fused-epilogue GEMMs (bf16, persistent 256x256 kernel only; callers fall back to the unfused
//      sequence nvit_gemm_nt + nvit_swiglu_fwd / nvit_qknorm_fwd when these report "not eligible") ----
extern "C" int nvit_gemm_nt_fusable(int dt, int M, int N, int K) {
  return (dt == NVIT_BF16 && M >= 1 && N % 256 == 0 && K % 64 == 0) ? 1 : 0;
}

extern "C" int nvit_gemm_nt_swiglu(int dt, const void* A, int lda, const void* B, int ldb, void* uv, void* xm, int M,
                                   int F, int K, const float* gs, float gscale, void* stream) {
  NVIT_REQUIRE(nvit_gemm_nt_fusable(dt, M, 2 * F, K), "gemm_nt_swiglu: shape/dtype not eligible for the fused kernel");
  NVIT_REQUIRE((lda * 2) % 16 == 0 && (ldb * 2) % 16 == 0 && lda >= K && ldb >= K, "gemm_nt_swiglu: bad leading dims");
  NVIT_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)uv | (uintptr_t)xm | (uintptr_t)gs) & 15) == 0,
               "gemm_nt_swiglu: pointers must be 16-byte aligned");
  NtArgs g{};
  g.A = (const char*)A;
  g.B = (const char*)B;
  g.C = uv;
  g.M = M;
  g.N = 2 * F;
  g.K = K;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = 2 * F;
  g.out_dt = NVIT_BF16;
  g.xm = xm;
  g.ld_xm = F;
  g.gs = gs;
  g.gscale = gscale;
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: A, B read; raw uv [M,2F] and gated x [M,F] written (bf16)
  ProfScope ps(NVIT_KID_GEMM_SWIGLU, 2.0 * M * (2.0 * F) * K, 2.0 * ((double)M * K + 2.0 * F * K + 3.0 * M * F), s);
  return nvit_gemm_nt_fused_launch(g, 3, s);
}

extern "C" int nvit_gemm_nt_swiglu_bwd(int dt, const void* A, int lda, const void* B, int ldb, const void* uv,
                                       void* duv, float* part, int M, int F, int K, const float* gs, float gscale,
                                       void* stream) {
  NVIT_REQUIRE(nvit_gemm_nt_fusable(dt, M, F, K), "gemm_nt_swiglu_bwd: shape/dtype not eligible for the fused kernel");
  NVIT_REQUIRE((lda * 2) % 16 == 0 && (ldb * 2) % 16 == 0 && lda >= K && ldb >= K,
               "gemm_nt_swiglu_bwd: bad leading dims");
  NVIT_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)uv | (uintptr_t)duv | (uintptr_t)gs | (uintptr_t)part) & 15) == 0,
               "gemm_nt_swiglu_bwd: pointers must be 16-byte aligned");
  NVIT_REQUIRE(!gs || part, "gemm_nt_swiglu_bwd: part buffer missing");
  NtArgs g{};
  g.A = (const char*)A;
  g.B = (const char*)B;
  g.C = duv;
  g.M = M;
  g.N = F;
  g.K = K;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = 2 * F;
  g.out_dt = NVIT_BF16;
  g.uv_in = uv;
  g.ld_uv = 2 * F;
  g.Fh = F;
  g.gs = gs;
  g.gscale = gscale;
  g.part = gs ? part : nullptr;
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: A, B and the saved raw uv [M,2F] read; d(uv) [M,2F] written (bf16) - this kernel moves 4 bytes per
  // MAC-column pair and is judged against HBM as well as against the MFMA peak (bench.py roofline.families)
  ProfScope ps(NVIT_KID_GEMM_SWIGLU_BWD, 2.0 * M * (double)F * K, 2.0 * ((double)M * K + (double)F * K + 4.0 * M * F), s);
  return nvit_gemm_nt_fused_launch(g, 5, s);
}

extern "C" int nvit_gemm_nt_qknorm(int dt, const void* A, int lda, const void* B, int ldb, int M, int K, int nparts,
                                   int part0, const float* sqk, float c_q, float q_prescale, void* qh, void* kh, void* vh,
                                   float* rq, float* rk, int T, int H, int d, void* stream) {
  const int C = H * d;
  NVIT_REQUIRE(d == 64 && C % 256 == 0 && nparts >= 1 && part0 >= 0 && part0 + nparts <= 3,
               "gemm_nt_qknorm: needs head dim 64 and n_embd %% 256 == 0");
  NVIT_REQUIRE(nvit_gemm_nt_fusable(dt, M, nparts * C, K), "gemm_nt_qknorm: shape/dtype not eligible");
  NVIT_REQUIRE((lda * 2) % 16 == 0 && (ldb * 2) % 16 == 0 && lda >= K && ldb >= K, "gemm_nt_qknorm: bad leading dims");
  NVIT_REQUIRE(M % T == 0, "gemm_nt_qknorm: M must be a multiple of T");
  NVIT_REQUIRE(q_prescale > 0.f, "gemm_nt_qknorm: q_prescale must be positive");
  NVIT_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)qh | (uintptr_t)kh | (uintptr_t)vh | (uintptr_t)sqk) & 15) == 0,
               "gemm_nt_qknorm: pointers must be 16-byte aligned");
  NtArgs g{};
  g.A = (const char*)A;
  g.B = (const char*)B;
  g.M = M;
  g.N = nparts * C;
  g.K = K;
  g.lda = lda;
  g.ldb = ldb;
  g.out_dt = NVIT_BF16;
  g.q_prescale = q_prescale;
  g.qh = qh;
  g.kh = kh;
  g.vh = vh;
  g.rq = rq;
  g.rk = rk;
  g.sqk = sqk;
  g.c_q = c_q;
  g.part0 = part0;
  g.Cemb = C;
  g.Ttok = T;
  g.H = H;
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: A, B read; nparts head tensors [M,C] written (bf16)
  ProfScope ps(NVIT_KID_GEMM_QKNORM, 2.0 * M * (double)(nparts * C) * K,
               2.0 * ((double)M * K + (double)nparts * C * K + (double)nparts * M * C), s);
  return nvit_gemm_nt_fused_launch(g, 4, s);
}
