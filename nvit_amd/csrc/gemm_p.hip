// Persistent NT GEMM for the large token-major GEMMs:  C[M,N] = A[M,K] * B[N,K]^T.
//
// Why a second kernel: with K = 768 a 128x128 tile has only 12 K-stages, so the per-tile prologue
// (first LDS-DMA round trip) and the one-stage-ahead prefetch leave the MFMA pipes waiting on
// L2/HBM latency, and a 128-wide tile needs ~2x the L2->LDS bytes per FLOP that a CU can sustain.
//   * one 512-thread workgroup per CU (8 waves, 2 per SIMD) walks its tiles persistently; the stage
//     stream is continuous across tile boundaries, so there is no per-tile prologue and the epilogue
//     stores overlap the LDS-DMA of the next tile;
//   * two tile shapes (template FM):  256x256 (waves 2x4, 128x64 each, 2-slot LDS ring of 64 KiB
//     stages = 128 KiB)  and  256x128 (waves 4x2, 64x64 each, 3-slot ring of 48 KiB = 144 KiB);
//   * per stage (BK = 64 bf16 / 32 fp32): issue DMA(s+NSLOT-1) into the slot vacated by the previous
//     stage, 2 x (fragment reads + MFMAs), counted s_waitcnt vmcnt retiring DMA(s+1) while younger
//     DMA stays in flight, one barrier; at a tile's last stage the epilogue re-shapes the accumulators
//     through a private 2 KiB LDS scratch per wave so that global stores are whole 128-byte rows;
//   * tiles are dealt so that the 32 workgroups of an XCD work on 32 consecutive tiles of an
//     8(m) x tiles_n m-fastest walk (shared A/B panels stay in that XCD's L2).
#include "gemm_common.h"

namespace {

template <int FM>
struct PCfg;
template <>
struct PCfg<4> {  // 256 x 128
  static constexpr int PBM = 256, PBN = 128, WR = 4, WC = 2, NSLOT = 3, A_DMA = 4, B_DMA = 2;
};
template <>
struct PCfg<8> {  // 256 x 256
  static constexpr int PBM = 256, PBN = 256, WR = 2, WC = 4, NSLOT = 2, A_DMA = 4, B_DMA = 4;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// EPI: 0 = generic direct epilogue, 1 = LDS-staged bf16 output, 2 = LDS-staged fp32 output,
//      3 = fused SwiGLU (bf16), 4 = fused q/k-normalise + head split (bf16), 5 = fused SwiGLU backward (bf16)
// DYN: tiles are handed out at run time instead of statically.  Each XCD owns a contiguous eighth of the (grouped)
// tile order and a ticket counter; a workgroup's first tile is static, every further one is `Gx + ticket` inside its
// XCD's range.  The ticket for the tile after next is drawn (one atomic by thread 0) while the current tile is being
// multiplied and published through LDS one stage later, so the scheduler costs no stall.  What it buys: when other
// kernels hold some CUs (the RCCL all-reduce of the data-parallel step), workgroups that start late simply take fewer
// tiles instead of delaying the whole launch by their start offset.  The last workgroup out resets the counters, so
// a launch leaves them at zero (hipGraph-replay safe).
template <typename T, int FM, int EPI, bool DYN>
__global__ __launch_bounds__(512) void gemm_nt_persistent_kernel(NtArgs g, int tiles_m, int ntiles, unsigned* sched) {
  using Cfg = PCfg<FM>;
  constexpr int PBM = Cfg::PBM, PBN = Cfg::PBN, NSLOT = Cfg::NSLOT;
  constexpr int A_BYTES = PBM * ROWB, B_BYTES = PBN * ROWB, SLOT_BYTES = A_BYTES + B_BYTES;
  constexpr int DPS = Cfg::A_DMA + Cfg::B_DMA;  // DMA wave-instructions per wave per stage
  constexpr int WROWS = 16 * FM;                // rows of a wave's sub-tile
  // global stores one wave issues in the epilogue of a tile that lies fully inside C (lower bound; 0 = unknown)
  constexpr int NST = EPI == 1 ? 2 * FM : EPI == 2 ? 4 * FM : EPI == 3 ? 3 * FM : EPI == 4 ? 2 * FM : EPI == 5 ? 8 : 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int EPC = 16 / sizeof(T);
  constexpr int BK = ROWB / sizeof(T);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / Cfg::WC, wc = wid % Cfg::WC;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nt = g.K / BK;
  const int G = gridDim.x;  // multiple of 8
  // static: workgroup -> position inside a round of G consecutive tiles, XCD x takes the x-th eighth of every round
  const int slot_in_round = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int my_tiles = slot_in_round < ntiles ? (ntiles - slot_in_round + G - 1) / G : 0;
  const int total_stages = my_tiles * nt;
  // dynamic: XCD x owns tiles [x_start, x_start + x_len)
  const int xcd = blockIdx.x & 7, Gx = G >> 3;
  const int x_start = (int)(((long long)ntiles * xcd) >> 3);
  const int x_len = (int)(((long long)ntiles * (xcd + 1)) >> 3) - x_start;
  const int first_tile = (int)(blockIdx.x >> 3) < x_len ? x_start + (int)(blockIdx.x >> 3) : -1;
  int* s_next = reinterpret_cast<int*>(smem + NSLOT * SLOT_BYTES + 8 * 2048);  // [2] next-tile announcements
  auto sched_exit = [&]() {   // last workgroup out resets the scheduler state
    if (tid == 0) {
      // (no __threadfence here: an agent-scope release writes back the XCD's dirty L2 lines, ~9 us per launch; the
      //  counters are only ever touched by atomics, and the reset below is ordered by the kernel boundary)
      if (atomicAdd(&sched[8], 1u) == (unsigned)(G - 1)) {
#pragma unroll
        for (int i = 0; i < 9; ++i) sched[i] = 0u;
      }
    }
  };
  if constexpr (DYN) {
    if (first_tile < 0) {
      sched_exit();
      return;
    }
  } else {
    if (total_stages == 0) return;
  }

  const int srow = lane >> 3;
  const int gc = (lane & 7) ^ srow;
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)smem);
  const unsigned wave_off = (unsigned)(wid * 1024);

  auto tile_of = [&](int pid, int& m0, int& n0) {   // pid = position in the grouped tile order
    constexpr int GM = 8;
    const int per_group = GM * g.tiles_n;
    const int group = pid / per_group, first_m = group * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = pid - group * per_group;
    m0 = (first_m + in_g % gsz) * PBM;
    n0 = (in_g / gsz) * PBN;
  };

  // ---- load cursor: LDS-DMA, one wave-instruction = 8 rows of 128 B; wave w owns groups 8*i + w
  // (a tile's rows as 32-bit lane offsets from the tile's first row; the tile / k-slab part of the address is scalar)
  // Only waves 0-3 issue LDS-DMA - their own row groups and those of the wave that shares their SIMD (wave + 4): an LDS-DMA
  // instruction stalls the issuing wave for ~50 cycles and a wave cannot multiply meanwhile, but its SIMD partner can:
  // the loader's DMA block runs beside the partner's first MFMAs instead of both waves issuing DMA, then both
  // contending for the matrix pipe.
  const bool loader = wid < 4;
  unsigned aoff[2][Cfg::A_DMA], boff[2][Cfg::B_DMA];
  unsigned long long abase = 0, bbase = 0;
  int l_it = 0, l_k = 0, l_slot = 0;
  // dynamic scheduling state (all wave-uniform except `ticket`, which only thread 0 uses)
  bool l_valid = true;      // the load cursor still has a tile
  int pending_c = -1;       // tile the compute cursor moves to at its next wrap (-1: none)
  int c_tile = DYN ? first_tile : slot_in_round;
  int wraps = 0;            // load-cursor wraps so far (parity selects the s_next entry)
  unsigned ticket = 0;
  int ticket_age = -1;      // stages since the draw was issued (-1: none pending)
  constexpr int PUBLISH_AGE = 3;   // the atomic's round trip under load is longer than one stage: consume it late
  auto draw = [&]() {       // issue the atomic now, publish its result PUBLISH_AGE stages later
    if (tid == 0) ticket = atomicAdd(&sched[xcd], 1u);
    ticket_age = 0;
  };
  auto publish = [&]() {    // call before every stage-end barrier
    if (ticket_age >= 0 && ++ticket_age > PUBLISH_AGE) {
      if (tid == 0) {
        const long long idx = (long long)Gx + ticket;
        s_next[wraps & 1] = idx < x_len ? x_start + (int)idx : -1;
      }
      ticket_age = -1;
    }
  };
  auto set_load_tile = [&](int pid) {
    int m0, n0;
    tile_of(pid, m0, n0);
    if (loader) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int w = wid + 4 * h;
#pragma unroll
        for (int i = 0; i < Cfg::A_DMA; ++i) {
          int ra = m0 + (i * 8 + w) * 8 + srow;
          ra = ra < g.M ? ra : g.M - 1;
          aoff[h][i] = (unsigned)(((size_t)(ra - m0) * g.lda + (size_t)gc * EPC) * sizeof(T));
        }
#pragma unroll
        for (int i = 0; i < Cfg::B_DMA; ++i) {
          int rb = n0 + (i * 8 + w) * 8 + srow;
          rb = rb < g.N ? rb : g.N - 1;
          boff[h][i] = (unsigned)(((size_t)(rb - n0) * g.ldb + (size_t)gc * EPC) * sizeof(T));
        }
      }
    }
    abase = (unsigned long long)(uintptr_t)g.A + (unsigned long long)((size_t)m0 * g.lda * sizeof(T));
    bbase = (unsigned long long)(uintptr_t)g.B + (unsigned long long)((size_t)n0 * g.ldb * sizeof(T));
  };
  auto issue_stage = [&]() {
    const unsigned bo = lds_base + wave_off + (unsigned)l_slot * SLOT_BYTES;
    const size_t ko = (size_t)l_k * ROWB;
    {
      if (loader) {
        glds16s_n<Cfg::A_DMA>(abase + ko, aoff[0], bo);
        glds16s_n<Cfg::A_DMA>(abase + ko, aoff[1], bo + 4096);
        glds16s_n<Cfg::B_DMA>(bbase + ko, boff[0], bo + A_BYTES);
        glds16s_n<Cfg::B_DMA>(bbase + ko, boff[1], bo + A_BYTES + 4096);
      }
    }
    l_slot = l_slot == NSLOT - 1 ? 0 : l_slot + 1;
    if (++l_k == nt) {
      l_k = 0;
      if constexpr (DYN) {
        // the announcement for this wrap was published >= 1 barrier ago (host: nt >= NSLOT + PUBLISH_AGE + 1)
        const int nxt = __builtin_amdgcn_readfirstlane(s_next[wraps & 1]);
        ++wraps;
        pending_c = nxt;
        if (nxt >= 0) {
          set_load_tile(nxt);
          draw();
        } else {
          l_valid = false;
        }
      } else {
        ++l_it;
        if (l_it < my_tiles) set_load_tile(l_it * G + slot_in_round);
      }
    }
  };

  f32x4 acc[FM][4];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment reads of one k-half (kk) of the stage sitting in ring slot `slot_`
#define LOAD_FRAGS(FA, FB, slot_, kk_)                                                          \
  {                                                                                             \
    const char* la_ = smem + (slot_) * SLOT_BYTES;                                              \
    const char* lb_ = la_ + A_BYTES;                                                            \
    const int c_ = (kk_) * 4 + lg;                                                              \
    _Pragma("unroll") for (int i = 0; i < FM; ++i) {                                            \
      const int r = wr * WROWS + i * 16 + l15;                                                  \
      FA[i] = *reinterpret_cast<const uint4*>(la_ + r * ROWB + ((c_ ^ (r & 7)) << 4));          \
    }                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                             \
      const int r = wc * 64 + j * 16 + l15;                                                     \
      FB[j] = *reinterpret_cast<const uint4*>(lb_ + r * ROWB + ((c_ ^ (r & 7)) << 4));          \
    }                                                                                           \
  }
#define MMA_FRAGS(FA, FB)                                                                       \
  _Pragma("unroll") for (int i = 0; i < FM; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)  \
      Mma<T>::run(FB[j], FA[i], acc[i][j]);

  // prologue: NSLOT-1 stages in flight, wait for stage 0
  set_load_tile(c_tile);
  if constexpr (DYN) draw();   // ticket for this workgroup's second tile; published at the end of stage 0
  issue_stage();
  const bool two_ahead = NSLOT > 2 && (DYN || total_stages > 1);   // DYN requires nt >= NSLOT + 4 (host check)
  if (two_ahead) issue_stage();
  // (counted waits: a loader wave has 2 * DPS DMA instructions per stage in flight, the other waves none)
  if (two_ahead && loader)
    wait_vmcnt<(NSLOT > 2 ? 2 * DPS : 0)>();
  else
    wait_vmcnt<0>();
  __syncthreads();

  // Per stage s (ring slot s % NSLOT): issue DMA(s+NSLOT-1) into the slot vacated by stage s-1,
  // multiply both k-halves of stage s, retire DMA(s+1) with a counted vmcnt (younger DMA stays in
  // flight), one barrier.
  int c_it = 0, c_k = 0;
  int slot = 0;
  for (int s = 0;; ++s) {
    // a stage is issued at the top of this iteration iff the load cursor still has work
    const bool issued_now = DYN ? l_valid : (s + NSLOT - 1 < total_stages);
    if (issued_now) issue_stage();
    {
      // Software-pipelined fragment stream.  The stage is 2*FM "steps" of 4 MFMAs (one A fragment against the
      // four B fragments of its k-half).  All ds_read_b128 are written first, in the order the steps consume
      // them; sched_group_barrier then pins the interleave: 7 reads up front (B of k-half 0 + three A), after
      // that every step's 4 MFMAs are followed by the read(s) needed ~3 steps later, so LDS latency hides
      // under ~12 MFMAs instead of stalling every 4 (what hipcc's own just-in-time schedule did).
      const char* la_ = smem + slot * SLOT_BYTES;
      const char* lb_ = la_ + A_BYTES;
      uint4 fa[2][FM], fb[2][4];
#define RA(kk_, i_)                                                                                       \
  {                                                                                                       \
    const int r = wr * WROWS + (i_) * 16 + l15;                                                           \
    fa[kk_][i_] = *reinterpret_cast<const uint4*>(la_ + r * ROWB + ((((kk_) * 4 + lg) ^ (r & 7)) << 4));  \
  }
#define RB(kk_, j_)                                                                                       \
  {                                                                                                       \
    const int r = wc * 64 + (j_) * 16 + l15;                                                              \
    fb[kk_][j_] = *reinterpret_cast<const uint4*>(lb_ + r * ROWB + ((((kk_) * 4 + lg) ^ (r & 7)) << 4));  \
  }
      __builtin_amdgcn_sched_barrier(0);
      RB(0, 0) RB(0, 1) RB(0, 2) RB(0, 3) RA(0, 0) RA(0, 1) RA(0, 2)
      if constexpr (FM == 8) {
        RA(0, 3) RB(1, 0) RA(0, 4) RB(1, 1) RA(0, 5) RB(1, 2) RA(0, 6) RB(1, 3) RA(0, 7)
        RA(1, 0) RA(1, 1) RA(1, 2) RA(1, 3) RA(1, 4) RA(1, 5) RA(1, 6) RA(1, 7)
      } else {
        RA(0, 3) RB(1, 0) RB(1, 1) RB(1, 2) RB(1, 3) RA(1, 0) RA(1, 1) RA(1, 2) RA(1, 3)
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) Mma<T>::run(fb[kk][j], fa[kk][i], acc[i][j]);
      constexpr int MPS = sizeof(T) == 2 ? 4 : 16;  // MFMA instructions per step (fp32: 4 per fragment pair)
      __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
#pragma unroll
      for (int t = 0; t < 2 * FM; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x8, MPS, 0);
        if constexpr (FM == 8) {
          if (t < 4)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          else if (t < 13)
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        } else {
          if (t < 3)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          else if (t < 6)
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#undef RA
#undef RB
    }
    bool stored = false, full_tile = false, more = true;
    if (++c_k == nt) {
      int m0, n0;
      tile_of(c_tile, m0, n0);
      full_tile = NST > 0 && m0 + PBM <= g.M && n0 + PBN <= g.N;
      {
        char* scratch = smem + NSLOT * SLOT_BYTES + wid * 2048;
        if constexpr (EPI == 1)
          nt_store_tile_staged<FM, bf16>(g, acc, m0 + wr * WROWS, n0 + wc * 64, lane, scratch);
        else if constexpr (EPI == 2)
          nt_store_tile_staged<FM, float>(g, acc, m0 + wr * WROWS, n0 + wc * 64, lane, scratch);
        else if constexpr (EPI == 3)
          nt_store_tile_swiglu<FM>(g, acc, m0 + wr * WROWS, n0 + wc * 64, lane, scratch);
        else if constexpr (EPI == 4)
          nt_store_tile_qknorm<FM>(g, acc, m0 + wr * WROWS, n0 + wc * 64, lane, scratch);
        else if constexpr (EPI == 5)
          nt_store_tile_swiglu_bwd<FM>(g, acc, m0 + wr * WROWS, n0 + wc * 64, lane, scratch);
        else
          nt_store_tile<FM>(g, acc, m0 + wr * WROWS, n0 + wc * 64, l15, lg);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      c_k = 0;
      stored = true;
      if constexpr (DYN) {
        c_tile = pending_c;   // set by the load cursor's wrap, which is always ahead of this one
        more = c_tile >= 0;
      } else {
        ++c_it;
        c_tile = c_it * G + slot_in_round;
        more = c_it < my_tiles;
      }
    }
    if constexpr (DYN) publish();
    // retire DMA(s+1).  vmcnt counts stores too, in issue order, and the epilogue's stores are younger than
    // every DMA issued so far: after a full tile (a known number of store instructions per wave) wait only
    // for what is older than them, so the write-back drains under the next tile's first stage instead of
    // stalling the whole workgroup on HBM write acknowledgements.
    if (stored) {
      if (full_tile && issued_now) {
        if (loader)
          wait_vmcnt<(NSLOT - 2) * 2 * DPS + NST>();
        else
          wait_vmcnt<NST>();
      } else {
        wait_vmcnt<0>();
      }
    } else if (NSLOT > 2 && issued_now && loader) {
      wait_vmcnt<(NSLOT > 2 ? (NSLOT - 2) * 2 * DPS : 0)>();
    } else {
      wait_vmcnt<0>();
    }
    if (!more) break;
    __syncthreads();
    slot = slot == NSLOT - 1 ? 0 : slot + 1;
  }
  if constexpr (DYN) sched_exit();
#undef LOAD_FRAGS
#undef MMA_FRAGS
}

// Scheduler state for the dynamic mode: 64 slots of 16 counters ([0..7] per-XCD tickets, [8] workgroups finished),
// a static device array owned by the library; consecutive launches rotate through the slots, and every launch leaves
// its slot zeroed.
__device__ unsigned nvit_sched_slots[64][16];
int g_gemm_sched_dynamic = -1;  // -1: read NVIT_GEMM_SCHED on first use

unsigned* sched_slot() {
  static unsigned* base[16] = {};
  static unsigned next = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!base[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(nvit_sched_slots)) != hipSuccess) return nullptr;
    base[dev] = (unsigned*)p;
  }
  return base[dev] + 16 * (next++ & 63);
}

bool sched_dynamic() {
  if (g_gemm_sched_dynamic < 0) {
    const char* e = getenv("NVIT_GEMM_SCHED");
    g_gemm_sched_dynamic = (e && (e[0] == 'd' || e[0] == '1')) ? 1 : 0;
  }
  return g_gemm_sched_dynamic == 1;
}

template <typename T, int FM, int EPI>
int launch_p2(const NtArgs& g_in, int n_cu, hipStream_t s) {
  using Cfg = PCfg<FM>;
  constexpr int LDS_STATIC = Cfg::NSLOT * (Cfg::PBM + Cfg::PBN) * ROWB + 8 * 2048;  // ring + epilogue scratch
  constexpr bool CAN_DYN = LDS_STATIC + 16 <= 160 * 1024;   // the 256x128 configuration fills the LDS: static only
  constexpr int LDS_BYTES = CAN_DYN ? LDS_STATIC + 16 : LDS_STATIC;                 // + s_next
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_persistent_kernel<T, FM, EPI, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess && CAN_DYN)
      e = hipFuncSetAttribute((const void*)gemm_nt_persistent_kernel<T, FM, EPI, CAN_DYN>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) NVIT_FAIL((int)e, "gemm_nt: cannot raise LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  NtArgs g = g_in;
  g.tiles_n = cdiv(g.N, Cfg::PBN);
  const int tiles_m = cdiv(g.M, Cfg::PBM);
  const int nt = g.K / (ROWB / (int)sizeof(T));
  unsigned* sched = (CAN_DYN && sched_dynamic() && nt >= Cfg::NSLOT + 4) ? sched_slot() : nullptr;
  if (sched)
    hipLaunchKernelGGL((gemm_nt_persistent_kernel<T, FM, EPI, CAN_DYN>), dim3(n_cu), dim3(512), LDS_BYTES, s, g, tiles_m,
                       tiles_m * g.tiles_n, sched);
  else
    hipLaunchKernelGGL((gemm_nt_persistent_kernel<T, FM, EPI, false>), dim3(n_cu), dim3(512), LDS_BYTES, s, g, tiles_m,
                       tiles_m * g.tiles_n, (unsigned*)nullptr);
  NVIT_CHECK_LAUNCH("gemm_nt_persistent");
  return NVIT_OK;
}

template <typename T, int FM>
int launch_p(const NtArgs& g, int n_cu, hipStream_t s) {
  const int eo = g.out_dt == NVIT_F32 ? 4 : 8;  // output elements per 16-byte chunk
  static const bool no_stage = getenv("NVIT_GEMM_DIRECT") != nullptr;  // experiments
  const bool staged = (g.N % eo) == 0 && (g.ldc % eo) == 0 && !no_stage;
  if (!staged) return launch_p2<T, FM, 0>(g, n_cu, s);
  return g.out_dt == NVIT_F32 ? launch_p2<T, FM, 2>(g, n_cu, s) : launch_p2<T, FM, 1>(g, n_cu, s);
}

}  // namespace

static int p_num_cu() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int devid = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&devid) != hipSuccess || hipGetDeviceProperties(&prop, devid) != hipSuccess) return 0;
    n_cu = prop.multiProcessorCount;
    n_cu -= n_cu % 8;
    if (n_cu < 8) n_cu = 8;
  }
  return n_cu;
}

// scheduling mode of the persistent NT GEMMs: 1 = dynamic tile hand-out (see the kernel comment), 0 = static
extern "C" int nvit_set_gemm_sched(int dynamic) {
  g_gemm_sched_dynamic = dynamic ? 1 : 0;
  return NVIT_OK;
}

// fused-epilogue launches (bf16 operands, 256x256 tiles): epi = 3 (SwiGLU), 4 (q/k normalise), 5 (SwiGLU backward)
int nvit_gemm_nt_fused_launch(const NtArgs& g, int epi, hipStream_t s) {
  const int n_cu = p_num_cu();
  if (n_cu == 0) NVIT_FAIL(NVIT_EINVAL, "gemm_nt: cannot query device properties");
  if (epi == 5) return launch_p2<bf16, 8, 5>(g, n_cu, s);
  return epi == 3 ? launch_p2<bf16, 8, 3>(g, n_cu, s) : launch_p2<bf16, 8, 4>(g, n_cu, s);
}

// tile_n: 128 or 256
int nvit_gemm_nt_persistent_launch(int dt, const NtArgs& g, int tile_n, hipStream_t s) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int devid = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&devid) != hipSuccess || hipGetDeviceProperties(&prop, devid) != hipSuccess)
      NVIT_FAIL(NVIT_EINVAL, "gemm_nt: cannot query device properties");
    n_cu = prop.multiProcessorCount;
    n_cu -= n_cu % 8;
    if (n_cu < 8) n_cu = 8;
    if (const char* e = getenv("NVIT_GEMM_CUS")) n_cu = atoi(e);  // experiments: restrict the persistent grid
  }
  if (dt == NVIT_BF16) return tile_n == 256 ? launch_p<bf16, 8>(g, n_cu, s) : launch_p<bf16, 4>(g, n_cu, s);
  return tile_n == 256 ? launch_p<float, 8>(g, n_cu, s) : launch_p<float, 4>(g, n_cu, s);
}
