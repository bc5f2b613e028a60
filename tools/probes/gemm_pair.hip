// Persistent NT GEMM, two independent workgroups per CU:  C[M,N] = A[M,K] * B[N,K]^T  (bf16 operands).
//
// gemm_p.hip runs ONE 512-thread workgroup per CU in lock step: while its 8 waves store a finished 256x256 tile
// (4.3 us for a bf16 tile, 9 us for an fp32 one on a 20 us K=768 main loop - tools/probes/gemm_parts.hip) no MFMA
// issues on that CU, and all 256 CUs hit that phase together (the fp32 store burst alone is HBM-write bound).
// Nothing in one workgroup can overlap the two phases: a second tile's accumulators do not fit the registers, a
// staged copy of the tile does not fit the LDS.  Here the CU is shared by TWO 256-thread workgroups instead
// (4 waves = one per SIMD each, 80 KiB of LDS each, 256 registers per wave): each walks its own 256x128 tiles with
// the same 128x64 wave tile and the same epilogues, and because they are independent they drift apart - one's
// epilogue, barrier waits and DMA issue run under the other's MFMAs.  Price: 1.5x the L2->LDS bytes per FLOP
// (384 operand rows per 32 768 outputs instead of 512 per 65 536), and K stages of 32 (64-byte LDS rows) so that a
// 3-slot ring fits: 3 x (256 + 128) x 64 B = 72 KiB + 4 x 2 KiB epilogue scratch = 80 KiB.
//   * 64-byte rows: ds_read_b128 serves 16 lanes per cycle from a 256-byte window = 4 rows; the 16-byte chunk c of
//     row r sits at position c ^ ((-(r >> 2)) & 3), which makes every lane group of the fragment read hit 16 distinct
//     chunks (derivation: DESIGN.md section 4);
//   * per stage: DMA(s+2) into the slot released by the barrier of stage s-1, 12 fragment reads + 32 MFMAs on slot
//     s % 3, counted vmcnt retiring DMA(s+1), one barrier of 4 waves;
//   * tile order, epilogues (EPI 1-5) and the counted wait after the epilogue stores are those of gemm_p.hip.
#include "../../nvit_amd/csrc/gemm_common.h"

namespace {

constexpr int QBM = 256, QBN = 128, QROWB = 64, QBK = 32;
constexpr int QA_BYTES = QBM * QROWB, QB_BYTES = QBN * QROWB, QSLOT = QA_BYTES + QB_BYTES;  // 16 + 8 = 24 KiB
constexpr int QNSLOT = 3, QA_DMA = 4, QB_DMA = 2, QDPS = QA_DMA + QB_DMA;
constexpr int QLDS = QNSLOT * QSLOT + 4 * 2048;   // 80 KiB: two workgroups per CU

template <int N>
__device__ __forceinline__ void q_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_pair_kernel(NtArgs g, int tiles_m, int ntiles) {
  constexpr int FM = 8;
  // global stores one wave issues in the epilogue of a tile that lies fully inside C
  constexpr int NST = EPI == 1 ? 2 * FM : EPI == 2 ? 4 * FM : EPI == 3 ? 3 * FM : EPI == 4 ? 2 * FM : EPI == 5 ? 8 : 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nt = g.K / QBK;
  const int G = gridDim.x;  // multiple of 8
  const int slot_in_round = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int my_tiles = slot_in_round < ntiles ? (ntiles - slot_in_round + G - 1) / G : 0;
  const int total_stages = my_tiles * nt;
  if (total_stages == 0) return;

  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)smem);
  auto tile_of = [&](int pid, int& m0, int& n0) {   // pid = position in the grouped tile order (8 m-tiles, m fastest)
    constexpr int GM = 8;
    const int per_group = GM * g.tiles_n;
    const int group = pid / per_group, first_m = group * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = pid - group * per_group;
    m0 = (first_m + in_g % gsz) * QBM;
    n0 = (in_g / gsz) * QBN;
  };

  // ---- load cursor: one wave-instruction = 16 rows x 64 B; wave w owns row groups 4*i + w
  const int drow = lane >> 2;
  const int dch = (lane & 3) ^ ((0 - (lane >> 4)) & 3);   // source chunk that lands at position lane & 3 of row drow
  const char* ap[QA_DMA];
  const char* bp[QB_DMA];
  int l_it = 0, l_k = 0, l_slot = 0;
  auto set_load_tile = [&](int pid) {
    int m0, n0;
    tile_of(pid, m0, n0);
#pragma unroll
    for (int i = 0; i < QA_DMA; ++i) {
      int ra = m0 + (i * 4 + wid) * 16 + drow;
      ra = ra < g.M ? ra : g.M - 1;
      ap[i] = g.A + ((size_t)ra * g.lda + (size_t)dch * 8) * sizeof(bf16);
    }
#pragma unroll
    for (int i = 0; i < QB_DMA; ++i) {
      int rb = n0 + (i * 4 + wid) * 16 + drow;
      rb = rb < g.N ? rb : g.N - 1;
      bp[i] = g.B + ((size_t)rb * g.ldb + (size_t)dch * 8) * sizeof(bf16);
    }
  };
  auto issue_stage = [&]() {
    const unsigned bo = lds_base + (unsigned)(l_slot * QSLOT + wid * 1024);
    const size_t ko = (size_t)l_k * QROWB;
#pragma unroll
    for (int i = 0; i < QA_DMA; ++i) glds16(ap[i] + ko, bo + i * 4096);
#pragma unroll
    for (int i = 0; i < QB_DMA; ++i) glds16(bp[i] + ko, bo + QA_BYTES + i * 4096);
    l_slot = l_slot == QNSLOT - 1 ? 0 : l_slot + 1;
    if (++l_k == nt) {
      l_k = 0;
      ++l_it;
      if (l_it < my_tiles) set_load_tile(l_it * G + slot_in_round);
    }
  };

  f32x4 acc[FM][4];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: two stages in flight, stage 0 landed
  set_load_tile(slot_in_round);
  issue_stage();
  if (total_stages > 1) {
    issue_stage();
    q_wait_vmcnt<QDPS>();
  } else {
    q_wait_vmcnt<0>();
  }
  __syncthreads();

  const int fsw = (lg ^ ((0 - (l15 >> 2)) & 3)) << 4;   // chunk position of this lane's 16 bytes in its fragment rows
  int c_it = 0, c_k = 0, c_tile = slot_in_round;
  int slot = 0;
  for (int s = 0;; ++s) {
    const bool issued_now = s + QNSLOT - 1 < total_stages;
    if (issued_now) issue_stage();
    {
      const char* la = smem + slot * QSLOT + (wr * 128 + l15) * QROWB + fsw;
      const char* lb = smem + slot * QSLOT + QA_BYTES + (wc * 64 + l15) * QROWB + fsw;
      uint4 fa[FM], fb[4];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(lb + j * 16 * QROWB);
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const uint4*>(la + i * 16 * QROWB);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<bf16>::run(fb[j], fa[i], acc[i][j]);
      // 6 reads up front (B + two A), then one A read after each step of 4 MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int t = 0; t < FM; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
        if (t < FM - 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    bool stored = false, full_tile = false, more = true;
    if (++c_k == nt) {
      int m0, n0;
      tile_of(c_tile, m0, n0);
      full_tile = NST > 0 && m0 + QBM <= g.M && n0 + QBN <= g.N;
      char* scratch = smem + QNSLOT * QSLOT + wid * 2048;
      if constexpr (EPI == 1)
        nt_store_tile_staged<FM, bf16>(g, acc, m0 + wr * 128, n0 + wc * 64, lane, scratch);
      else if constexpr (EPI == 2)
        nt_store_tile_staged<FM, float>(g, acc, m0 + wr * 128, n0 + wc * 64, lane, scratch);
      else if constexpr (EPI == 3)
        nt_store_tile_swiglu<FM>(g, acc, m0 + wr * 128, n0 + wc * 64, lane, scratch);
      else if constexpr (EPI == 4)
        nt_store_tile_qknorm<FM>(g, acc, m0 + wr * 128, n0 + wc * 64, lane, scratch);
      else if constexpr (EPI == 5)
        nt_store_tile_swiglu_bwd<FM>(g, acc, m0 + wr * 128, n0 + wc * 64, lane, scratch);
      else
        nt_store_tile<FM>(g, acc, m0 + wr * 128, n0 + wc * 64, l15, lg);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      c_k = 0;
      stored = true;
      ++c_it;
      c_tile = c_it * G + slot_in_round;
      more = c_it < my_tiles;
    }
    // retire DMA(s+1); DMA(s+2) - and, after a full tile, the epilogue's stores, which are younger than every DMA
    // issued so far - stay in flight
    if (stored) {
      if (full_tile && issued_now)
        q_wait_vmcnt<QDPS + NST>();
      else
        q_wait_vmcnt<0>();
    } else if (issued_now) {
      q_wait_vmcnt<QDPS>();
    } else {
      q_wait_vmcnt<0>();
    }
    if (!more) break;
    __syncthreads();
    slot = slot == QNSLOT - 1 ? 0 : slot + 1;
  }
}

template <int EPI>
int launch_pair(const NtArgs& g_in, int n_cu, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_pair_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, QLDS);
    if (e != hipSuccess) NVIT_FAIL((int)e, "gemm_nt: cannot raise LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  NtArgs g = g_in;
  g.tiles_n = cdiv(g.N, QBN);
  const int tiles_m = cdiv(g.M, QBM);
  hipLaunchKernelGGL((gemm_nt_pair_kernel<EPI>), dim3(2 * n_cu), dim3(256), QLDS, s, g, tiles_m, tiles_m * g.tiles_n);
  NVIT_CHECK_LAUNCH("gemm_nt_pair");
  return NVIT_OK;
}

int pair_num_cu() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int devid = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&devid) != hipSuccess || hipGetDeviceProperties(&prop, devid) != hipSuccess) return 0;
    n_cu = prop.multiProcessorCount;
    n_cu -= n_cu % 8;
    if (n_cu < 8) n_cu = 8;
    if (const char* e = getenv("NVIT_GEMM_CUS")) n_cu = atoi(e);  // experiments: restrict the persistent grid
  }
  return n_cu;
}

}  // namespace

// bf16 operands, K % 32 == 0.  epi: 1 / 2 (plain, by g.out_dt; needs the staged-store alignment), 3, 4, 5 (fused).
int nvit_gemm_nt_pair_launch(const NtArgs& g, int epi, hipStream_t s) {
  const int n_cu = pair_num_cu();
  if (n_cu == 0) NVIT_FAIL(NVIT_EINVAL, "gemm_nt: cannot query device properties");
  switch (epi) {
    case 1: return launch_pair<1>(g, n_cu, s);
    case 2: return launch_pair<2>(g, n_cu, s);
    case 3: return launch_pair<3>(g, n_cu, s);
    case 4: return launch_pair<4>(g, n_cu, s);
    case 5: return launch_pair<5>(g, n_cu, s);
    default: return launch_pair<0>(g, n_cu, s);
  }
}
