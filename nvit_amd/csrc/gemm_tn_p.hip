// Persistent weight-gradient GEMM:  G[n][k'] = sum_m A[m][n] * B[m][k']   (A [Mred,N], B [Mred,K]).
//
// Same machine as gemm_p.hip, turned for the "both operands are reduction-major" case:
//   * output tile 256 (n) x 256 (k'), 8 waves (2 x 4: 128 k' x 64 n each), one workgroup per CU;
//   * work items = (output tile, row split); a workgroup walks its items persistently, the LDS-DMA
//     stream is continuous across items.  Ring (template DEEP): 2 slots of 64 KiB stages (64 reduction rows of
//     both operands; default), or 4 slots of 32 KiB with three stages in flight behind a counted vmcnt
//     (NVIT_TN_RING=4).  vmcnt retires in order, so the look-ahead a wave can have IS the ring depth; the deeper
//     ring was built to test whether the 28 % of wave cycles this kernel spends in s_waitcnt/barrier (PMC) is memory
//     latency - it is not: at the same 128 KiB the 4-slot ring is 2 % SLOWER on the four block shapes
//     (tools/tn_ab.py, interleaved rounds), i.e. the operands arrive in time and the extra barrier per 32 rows costs
//     more than the look-ahead buys;
//   * tiles are stored in LDS exactly as they sit in memory ([m][col], 512-byte rows, 16-byte chunks
//     XOR-swizzled by the row so that the transposed reads are conflict free) and the MFMA operands
//     are fetched with ds_read_b64_tr_b16 (hardware 4x16 transpose); fp32 uses ds_read_b32;
//   * each item writes its fp32 256x256 partial tile to the slab of its split; slab_reduce (gemm.hip)
//     sums the splits in a fixed order (deterministic).
// Requires N % 256 == 0 and K % 256 == 0 (the Base/Large weight shapes); other shapes use gemm_tn.
#include "gemm_common.h"

namespace {

constexpr int TBN = 256, TBK = 256;
constexpr int TN_LDS = 131072;             // 128 KiB ring: 2 x 64 KiB or 4 x 32 KiB stages

struct TnpArgs {
  const char* A;
  const char* B;
  float* ws;
  const float* zeros;
  int Mred, N, K;
  int lda, ldb;
  int rows_per_split, splits;
  int tiles_n, tiles_k;
  int xcd_order;  // 1: XCD-contiguous deal of the items (default); 0: round-robin (NVIT_TN_ORDER=0, for A/B runs)
};

__device__ __forceinline__ int tnp_swz(int m) { return (((m & 3) | (((m >> 3) & 1) << 2)) << 1); }

template <typename T, int DEEP>
__global__ __launch_bounds__(512) void gemm_tn_persistent_kernel(TnpArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NSLOT = DEEP ? 4 : 2;
  constexpr int TSLOT_BYTES = TN_LDS / NSLOT;        // 64 / 32 KiB
  constexpr int OP_BYTES = TSLOT_BYTES / 2;          // one operand stage
  constexpr int ROW_BYTES = 256 * sizeof(T);         // 512 / 1024
  constexpr int RB = OP_BYTES / ROW_BYTES;           // reduction rows per stage: bf16 64 / 32, fp32 32 / 16
  constexpr int ODMA = OP_BYTES / 8192;              // DMA wave-instructions per operand per wave per stage: 4 / 2
  constexpr int DPS = 2 * ODMA;                      // ... per stage
  constexpr int KS = RB * (int)sizeof(T) / 64;       // MFMA k-steps per stage (32 bf16 / 16 fp32 rows each): 2 / 1
  constexpr int CHUNKS = ROW_BYTES / 16;             // 32 / 64
  constexpr int RPI = 1024 / ROW_BYTES;              // rows per DMA wave-instruction: 2 / 1
  constexpr int EPC = 16 / sizeof(T);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;  // wr: k' half (MFMA rows, 128 each); wc: n quarter (64 each)
  const int l15 = lane & 15, lg = lane >> 4;
  const int G = gridDim.x;
  const int tiles = g.tiles_n * g.tiles_k;
  const int nitems = tiles * g.splits;
  // XCD-contiguous deal (workgroups go round-robin to the 8 XCDs): within every round of G consecutive items, the
  // workgroups of one XCD take a contiguous eighth.  Items are ordered (split, n tile, k' tile), so the ~32 items an XCD
  // works on at one time are tiles of the SAME reduction rows that share A and B panels: each panel slice is fetched
  // once into that XCD's L2 and re-read there by its other users, instead of once per tile from the fabric
  // (r01: FETCH x2 + WRITE = 3.5x the algorithmic bytes with the round-robin deal).
  const int xq = G >> 3, xr = G & 7, xcd = blockIdx.x & 7;
  const int first = g.xcd_order
                        ? (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (int)(blockIdx.x >> 3)
                        : (int)blockIdx.x;
  const int my_items = first < nitems ? (nitems - first + G - 1) / G : 0;
  if (my_items == 0) return;

  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)smem);
  const unsigned wave_off = (unsigned)(wid * 1024);
  const int lrow = lane / CHUNKS, lslot = lane % CHUNKS;
  const char* zsrc = reinterpret_cast<const char*>(g.zeros);

  // item -> (n0, k0, mbeg, mend, split)
  auto item_of = [&](int it, int& n0, int& k0, int& mbeg, int& mend, int& split) {
    const int id = first + it * G;
    split = id / tiles;
    const int tile = id - split * tiles;
    n0 = (tile / g.tiles_k) * TBN;
    k0 = (tile % g.tiles_k) * TBK;
    mbeg = split * g.rows_per_split;
    mend = mbeg + g.rows_per_split;
    if (mend > g.Mred) mend = g.Mred;
  };
  auto stages_of = [&](int it) {
    int n0, k0, mbeg, mend, split;
    item_of(it, n0, k0, mbeg, mend, split);
    const int rows = mend - mbeg;
    return rows > 0 ? (rows + RB - 1) / RB : 0;
  };

  // ---- load cursor: a lane's rows / chunks never change (32-bit offsets, formed once); the item and the row slab are the
  // scalar part of the address
  // Only waves 0-3 issue LDS-DMA, for their own row groups and those of their SIMD partner (wave + 4): see gemm_p.hip.
  const bool loader = wid < 4;
  unsigned aoff[2][ODMA], boff[2][ODMA];
  int srow[2][ODMA];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < ODMA; ++i) {
      srow[h][i] = (i * 8 + (wid & 3) + 4 * h) * RPI + lrow;
      int ch;
      if constexpr (sizeof(T) == 2)
        ch = lslot ^ tnp_swz(srow[h][i]);
      else
        ch = lslot;
      aoff[h][i] = (unsigned)(((size_t)srow[h][i] * g.lda + (size_t)ch * EPC) * sizeof(T));
      boff[h][i] = (unsigned)(((size_t)srow[h][i] * g.ldb + (size_t)ch * EPC) * sizeof(T));
    }
  unsigned long long a_item = 0, b_item = 0;   // first row of the item's slab, its column block
  int l_it = 0, l_t = 0, l_nt = 0, l_mbeg = 0, l_mend = 0, l_slot = 0;
  size_t astep = (size_t)RB * g.lda * sizeof(T), bstep = (size_t)RB * g.ldb * sizeof(T);
  auto set_load_item = [&](int it) {
    int n0, k0, split;
    item_of(it, n0, k0, l_mbeg, l_mend, split);
    const int rows = l_mend - l_mbeg;
    l_nt = rows > 0 ? (rows + RB - 1) / RB : 0;
    a_item = (unsigned long long)(uintptr_t)g.A + (unsigned long long)(((size_t)l_mbeg * g.lda + n0) * sizeof(T));
    b_item = (unsigned long long)(uintptr_t)g.B + (unsigned long long)(((size_t)l_mbeg * g.ldb + k0) * sizeof(T));
  };
  auto advance_to_nonempty = [&]() {
    while (l_it < my_items && l_nt == 0) {
      ++l_it;
      if (l_it < my_items) set_load_item(l_it);
    }
  };
  auto issue_stage = [&]() {
    const unsigned bo0 = lds_base + (unsigned)((wid & 3) * 1024) + (unsigned)l_slot * TSLOT_BYTES;
    const unsigned long long sa = a_item + (unsigned long long)((size_t)l_t * astep);
    const unsigned long long sb = b_item + (unsigned long long)((size_t)l_t * bstep);
    if (loader) {
      if (l_mbeg + (l_t + 1) * RB <= l_mend) {   // (wave-uniform) a full slab: scalar base + lane offset
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          glds16s_n<ODMA>(sa, aoff[h], bo0 + h * 4096);
          glds16s_n<ODMA>(sb, boff[h], bo0 + h * 4096 + OP_BYTES);
        }
      } else {                                   // the item's ragged last slab: rows past its end come from the zero page
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < ODMA; ++i) {
            const bool mok = (l_mbeg + l_t * RB + srow[h][i]) < l_mend;
            const char* pa = mok ? reinterpret_cast<const char*>((uintptr_t)sa) + aoff[h][i] : zsrc;
            const char* pb = mok ? reinterpret_cast<const char*>((uintptr_t)sb) + boff[h][i] : zsrc;
            glds16(pa, bo0 + h * 4096 + i * 8192);
            glds16(pb, bo0 + h * 4096 + OP_BYTES + i * 8192);
          }
      }
    }
    l_slot = l_slot == NSLOT - 1 ? 0 : l_slot + 1;
    if (++l_t == l_nt) {
      l_t = 0;
      ++l_it;
      l_nt = 0;
      if (l_it < my_items) {
        set_load_item(l_it);
        advance_to_nonempty();
      }
    }
  };

  // total number of stages this workgroup will process
  int total_stages = 0;
  for (int it = 0; it < my_items; ++it) total_stages += stages_of(it);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // compute cursor (skips empty items, which still must write zeros to their slab tile)
  int c_it = 0, c_t = 0, c_nt = stages_of(0);
  auto store_item = [&](int it) {
    int n0, k0, mbeg, mend, split;
    item_of(it, n0, k0, mbeg, mend, split);
    float* out = g.ws + (size_t)split * g.N * g.K;
    // acc[i][j][r] = G[n = n0 + wc*64 + 16j + l15][k' = k0 + wr*128 + 16i + 4lg + r]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + l15;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int kb = k0 + wr * 128 + i * 16 + 4 * lg;
        *reinterpret_cast<f32x4*>(out + (size_t)n * g.K + kb) = acc[i][j];
        acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  while (c_it < my_items && c_nt == 0) {  // leading empty items
    store_item(c_it);
    ++c_it;
    c_nt = c_it < my_items ? stages_of(c_it) : 0;
  }
  if (total_stages == 0) return;

  set_load_item(0);
  advance_to_nonempty();
  // prologue: NSLOT-1 stages in flight, stage 0 landed
  int issued = 0;
  for (; issued < NSLOT - 1 && issued < total_stages; ++issued) issue_stage();
  // (a loader wave has 2 * DPS DMA instructions per stage in flight, the other waves none)
  if (issued >= 3 && loader)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * DPS) : "memory");
  else if (issued == 2 && loader)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int slot = 0;
  for (int s = 0; s < total_stages; ++s) {
    // DMA(s+NSLOT-1) into the slot vacated by stage s-1 (every wave passed the barrier that ended it)
    if (s + NSLOT - 1 < total_stages) issue_stage();
    const char* la = smem + slot * TSLOT_BYTES;  // A tile: [m][n]  -> MFMA B operand (cols = n)
    const char* lb = la + OP_BYTES;              // B tile: [m][k'] -> MFMA A operand (rows = k')
    if constexpr (sizeof(T) == 2) {
      // Software-pipelined fragment stream (same idea as gemm_p.hip): 16 steps of 4 MFMAs (one k' fragment of the
      // B tile against the four n fragments of the A tile); every ds_read_b64_tr_b16 pair is written in
      // consumption order and sched_group_barrier pins "4 MFMAs, then the reads needed ~3 steps later".
      const int q = l15 >> 2, p = l15 & 3;
      uint4 fa[KS][4], fb[KS][8];
#define TRF(dst_, base_, ks_, col0_)                                                                              \
  {                                                                                                               \
    const int mrow = (ks_) * 32 + lg * 8 + q;                                                                     \
    const int ch = ((col0_) >> 3) + (p >> 1);                                                                     \
    const int o0 = mrow * ROW_BYTES + ((ch ^ tnp_swz(mrow)) << 4) + ((p & 1) << 3);                               \
    const int o1 = (mrow + 4) * ROW_BYTES + ((ch ^ tnp_swz(mrow + 4)) << 4) + ((p & 1) << 3);                     \
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((base_) + o0)); \
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((base_) + o1)); \
    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);                                 \
    dst_ = make_uint4(l2.x, l2.y, h2.x, h2.y);                                                                    \
  }
#define FA(ks_, j_) TRF(fa[ks_][j_], la, ks_, wc * 64 + (j_) * 16)
#define FB(ks_, i_) TRF(fb[ks_][i_], lb, ks_, wr * 128 + (i_) * 16)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (KS == 2) {
        FA(0, 0) FA(0, 1) FA(0, 2) FA(0, 3) FB(0, 0) FB(0, 1) FB(0, 2)
        FB(0, 3)
        FB(0, 4) FA(1, 0) FB(0, 5) FA(1, 1) FB(0, 6) FA(1, 2) FB(0, 7) FA(1, 3)
        FB(1, 0) FB(1, 1) FB(1, 2) FB(1, 3) FB(1, 4) FB(1, 5) FB(1, 6) FB(1, 7)
      } else {
        FA(0, 0) FA(0, 1) FA(0, 2) FA(0, 3) FB(0, 0) FB(0, 1) FB(0, 2)
        FB(0, 3) FB(0, 4) FB(0, 5) FB(0, 6) FB(0, 7)
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) Mma<bf16>::run(fb[ks][i], fa[ks][j], acc[i][j]);
      __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
      if constexpr (KS == 2) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
          if (t == 0)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          else if (t < 5)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          else if (t < 13)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
      } else {
        // 8 steps of 4 MFMAs; the 5 fragments (10 reads) not yet requested follow the first five steps
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
          if (t < 5) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#undef FA
#undef FB
#undef TRF
    } else {
#pragma unroll
      for (int ks = 0; ks < RB / 4; ++ks) {
        const int mrow = ks * 4 + lg;
        float fa[4], fb[8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          fa[j] = *reinterpret_cast<const float*>(la + mrow * ROW_BYTES + (wc * 64 + j * 16 + l15) * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          fb[i] = *reinterpret_cast<const float*>(lb + mrow * ROW_BYTES + (wr * 128 + i * 16 + l15) * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[i], fa[j], acc[i][j], 0, 0, 0);
      }
    }
    bool stored = false;
    if (++c_t == c_nt) {
      store_item(c_it);
      stored = true;
      ++c_it;
      c_t = 0;
      c_nt = c_it < my_items ? stages_of(c_it) : 0;
      while (c_it < my_items && c_nt == 0) {  // empty items in between
        store_item(c_it);
        ++c_it;
        c_nt = c_it < my_items ? stages_of(c_it) : 0;
      }
    }
    // retire DMA(s+1); the stages issued after it stay in flight.  (After an item's slab stores - younger than every
    // DMA in the queue - drain everything: once per item.)
    const int ahead = total_stages - 1 - (s + 1);   // stages issued beyond s+1 that exist
    if (stored || NSLOT == 2 || ahead <= 0 || !loader)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (ahead == 1)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * DPS) : "memory");
    __syncthreads();
    slot = slot == NSLOT - 1 ? 0 : slot + 1;
  }
}

}  // namespace

// item deal of the persistent weight-gradient kernel: 1 = XCD-contiguous (default), 0 = round-robin (A/B measurements);
// bits 1/2 of the argument select the ring: +2 = 4 x 32 KiB slots, +4 = 2 x 64 KiB slots (default)
static int g_tn_order = -1, g_tn_deep = -1;
extern "C" int nvit_set_tn_order(int mode) {
  g_tn_order = (mode & 1) ? 1 : 0;
  if (mode & 2) g_tn_deep = 1;
  if (mode & 4) g_tn_deep = 0;
  return NVIT_OK;
}

// Returns NVIT_OK after launching, or a negative value (-1) when the shape is not eligible.
int nvit_gemm_tn_persistent_launch(int dt, const void* A, int lda, const void* B, int ldb, float* ws,
                                   const float* zeros, int Mred, int N, int K, int splits, hipStream_t s) {
  if (N % TBN != 0 || K % TBK != 0) return -1;
  static int n_cu = 0;
  if (n_cu == 0) {
    int devid = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&devid) != hipSuccess || hipGetDeviceProperties(&prop, devid) != hipSuccess)
      NVIT_FAIL(NVIT_EINVAL, "gemm_tn: cannot query device properties");
    n_cu = prop.multiProcessorCount;
  }
  if (g_tn_deep < 0) g_tn_deep = getenv("NVIT_TN_RING") ? (atoi(getenv("NVIT_TN_RING")) == 4) : 0;   // measured (tools/tn_ab.py): the deeper ring is 2 % slower
  const int deep = g_tn_deep;
  static bool attr_set[2][2] = {{false, false}, {false, false}};
  const int idx = dt == NVIT_BF16 ? 1 : 0;
  if (!attr_set[idx][deep]) {
    const void* fn = dt == NVIT_BF16 ? (deep ? (const void*)gemm_tn_persistent_kernel<bf16, 1> : (const void*)gemm_tn_persistent_kernel<bf16, 0>)
                                     : (deep ? (const void*)gemm_tn_persistent_kernel<float, 1> : (const void*)gemm_tn_persistent_kernel<float, 0>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS);
    if (e != hipSuccess) NVIT_FAIL((int)e, "gemm_tn: cannot raise LDS limit: %s", hipGetErrorString(e));
    attr_set[idx][deep] = true;
  }
  TnpArgs g;
  g.A = (const char*)A;
  g.B = (const char*)B;
  g.ws = ws;
  g.zeros = zeros;
  g.Mred = Mred;
  g.N = N;
  g.K = K;
  g.lda = lda;
  g.ldb = ldb;
  g.splits = splits;
  const int rb = dt == NVIT_BF16 ? 64 : 32;   // row-split granularity (a multiple of both ring variants' stage rows)
  int rps = cdiv(Mred, splits);
  rps = cdiv(rps, rb) * rb;
  g.rows_per_split = rps;
  g.tiles_n = N / TBN;
  g.tiles_k = K / TBK;
  if (g_tn_order < 0) g_tn_order = getenv("NVIT_TN_ORDER") ? atoi(getenv("NVIT_TN_ORDER")) : 1;
  g.xcd_order = g_tn_order;
  const int nitems = g.tiles_n * g.tiles_k * splits;
  const int grid = nitems < n_cu ? nitems : n_cu;
  if (dt == NVIT_BF16) {
    if (deep)
      hipLaunchKernelGGL((gemm_tn_persistent_kernel<bf16, 1>), dim3(grid), dim3(512), TN_LDS, s, g);
    else
      hipLaunchKernelGGL((gemm_tn_persistent_kernel<bf16, 0>), dim3(grid), dim3(512), TN_LDS, s, g);
  } else {
    if (deep)
      hipLaunchKernelGGL((gemm_tn_persistent_kernel<float, 1>), dim3(grid), dim3(512), TN_LDS, s, g);
    else
      hipLaunchKernelGGL((gemm_tn_persistent_kernel<float, 0>), dim3(grid), dim3(512), TN_LDS, s, g);
  }
  NVIT_CHECK_LAUNCH("gemm_tn_persistent");
  return NVIT_OK;
}
