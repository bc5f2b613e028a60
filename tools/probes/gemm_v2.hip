// PROBE (not part of libnvit_hip.so): a one-wave-per-SIMD main loop for the persistent NT GEMM, built to test the
// hypothesis of DESIGN.md section 7 (round 2) that 128x128 wave tiles would lift the LDS-port co-limit of the 8-wave
// kernel.  Measured on MI355X against gemm_p.hip (tools/probes/gemm_v2_bench.hip, Base shapes, M = 100 352,
// gpurun_out/v2_bench1.log, v2_bench2.log; interleaved rounds in one process):
//   * hipcc cannot allocate it: with the builtin MFMA the 256 accumulators are shuttled between the two halves of
//     the register file at every iteration (254 v_accvgpr/scratch instructions in a 440-line stage body): bit-exact
//     with gemm_p.hip on every shape and epilogue, 1.5-2.0x SLOWER;
//   * with the MFMAs issued from inline asm ("a" constraints, V2_ASM_MFMA) the steady-state stage body is clean
//     (128 MFMA, 32 ds_read_b128, 16 LDS-DMA pieces, 8 accvgpr moves) and still 1.34-1.75x slower than the 8-wave
//     kernel (K=6144: 670 vs 1 057 TF/s); the allocator additionally parks one accumulator quad in VGPRs in the
//     tail variant of the loop, next to MFMAs whose hazards it cannot see (1 024 wrong elements in the last tile
//     of every workgroup) - inline-asm MFMAs with compiler-managed accumulators are not a safe product form;
//   * without any LDS-DMA after the ring is primed (NVIT_PROBE_V2_NO_DMA; same switch on gemm_p.hip) it is still
//     slower: K=6144 1 237 vs 1 411 TF/s-equivalent, K=3072 1 091 vs 1 347.  One wave per SIMD does NOT keep the
//     matrix pipe fuller than two: the fragment-read / MFMA overlap a partner wave gives for free has to be
//     hand-placed instruction by instruction (the guide's 1-wave attention kernel is hand-scheduled asm), and the
//     16 LDS-DMA pieces per stage per wave are issued in the wave's own MFMA time.
// Verdict: not the next step for this GEMM; kept as the record of the measurement.
//
// Persistent NT GEMM, second main loop:  C[M,N] = A[M,K] * B[N,K]^T, bf16 operands, 256x256 tiles.
//
// What differs from gemm_p.hip (the cross-check kernel, same ring, same epilogues):
//   * 4 waves per workgroup, ONE wave per SIMD, each wave a 128x128 sub-tile: 256 accumulator registers (the
//     accumulation half of the 512-entry file), operand fragments in the other half.  Per 64-deep stage the workgroup
//     reads 128 KiB of fragments from LDS instead of 192 KiB (a third fewer LDS bytes per MFMA: the persistent 8-wave
//     kernel is LDS-port co-limited) and no wave shares its matrix pipe with a partner.
//   * the fragment stream is software-pipelined ACROSS the stage barrier: a stage is two phases of 64 MFMAs (k-halves);
//     phase 0 multiplies k-half 0 while reading k-half 1, then the barrier, then phase 1 multiplies k-half 1 while it
//     reads k-half 0 of the NEXT stage and issues the LDS-DMA of the stage after next into the slot just vacated.  The
//     only time the matrix pipe waits for LDS is the first phase of a launch.
//   * every wave issues its 16 LDS-DMA pieces (1 KiB each) of a stage one per 4 MFMAs, base address in scalar
//     registers, per-lane offsets recomputed once per tile.
#include "../../nvit_amd/csrc/gemm_common.h"

namespace {

template <int N>
__device__ __forceinline__ void v2_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS-DMA of 16 bytes per lane, source = wave-uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset,
// destination = lds_off (wave-uniform, through M0) + lane * 16.
__device__ __forceinline__ void v2_glds(unsigned long long sbase, unsigned voff, unsigned lds_off) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_off)
      : "memory");
}

__device__ __forceinline__ unsigned long long v2_uniform64(unsigned long long x) {
  return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x) |
         ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32)) << 32);
}

constexpr int V2_BM = 256, V2_BN = 256;
constexpr int V2_A_BYTES = V2_BM * ROWB, V2_SLOT = (V2_BM + V2_BN) * ROWB;   // 32 KiB, 64 KiB
constexpr int V2_LDS = 2 * V2_SLOT + 4 * 2048;                                // ring + per-wave epilogue scratch

// EPI as in gemm_p.hip: 0 generic, 1 staged bf16, 2 staged fp32, 3 SwiGLU, 4 q/k-normalise, 5 SwiGLU backward
template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_nt_v2_kernel(NtArgs g,
                                                                                                      int tiles_m,
                                                                                                      int ntiles) {
  // global stores one wave issues in the epilogue of a full tile (its two 128x64 halves)
  // (a lower bound is all the counted wait below needs; the counter itself is 6 bits wide)
  constexpr int NST_ = 2 * (EPI == 1 ? 16 : EPI == 2 ? 32 : EPI == 3 ? 24 : EPI == 4 ? 16 : EPI == 5 ? 8 : 0);
  constexpr int NST = NST_ > 48 ? 48 : NST_;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nt = g.K / 64;
  const int G = gridDim.x;   // multiple of 8
  const int slot_in_round = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int my_tiles = slot_in_round < ntiles ? (ntiles - slot_in_round + G - 1) / G : 0;
  const int total = my_tiles * nt;
  if (total == 0) return;

  const int srow = lane >> 3;
  const int gc = (lane & 7) ^ srow;
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)smem);
  const unsigned wave_off = (unsigned)(wid * 1024);

  auto tile_of = [&](int pid, int& m0, int& n0) {
    constexpr int GM = 8;
    const int per_group = GM * g.tiles_n;
    const int group = pid / per_group, first_m = group * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = pid - group * per_group;
    m0 = (first_m + in_g % gsz) * V2_BM;
    n0 = (in_g / gsz) * V2_BN;
  };

  // ---- load cursor: wave w owns the 8-row groups 4*i + w (i = 0..7) of each operand
  unsigned voa[8], vob[8];
  int l_it = 0, l_k = 0;
  auto set_load_tile = [&](int pid) {
    int m0, n0;
    tile_of(pid, m0, n0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int ra = m0 + (i * 4 + wid) * 8 + srow;
      ra = ra < g.M ? ra : g.M - 1;
      voa[i] = (unsigned)ra * (unsigned)(g.lda * 2) + (unsigned)gc * 16u;
      int rb = n0 + (i * 4 + wid) * 8 + srow;
      rb = rb < g.N ? rb : g.N - 1;
      vob[i] = (unsigned)rb * (unsigned)(g.ldb * 2) + (unsigned)gc * 16u;
    }
  };
  // stage being issued: scalar bases + LDS slot offset, fixed for the 16 pieces of the stage
  unsigned long long sA = 0, sB = 0;
  unsigned l_bo = 0;
  int l_stage = 0;   // index (in this workgroup's stage stream) of the stage being issued
  auto begin_stage = [&]() {
    sA = v2_uniform64((unsigned long long)(uintptr_t)g.A + (unsigned long long)l_k * ROWB);
    sB = v2_uniform64((unsigned long long)(uintptr_t)g.B + (unsigned long long)l_k * ROWB);
    l_bo = lds_base + wave_off + (unsigned)(l_stage & 1) * V2_SLOT;
    // the bases come out of v_readfirstlane: five wait states before a vector-memory instruction reads them
    asm volatile("s_nop 4" : "+s"(sA), "+s"(sB));
  };
  auto end_stage = [&]() {
    ++l_stage;
    if (++l_k == nt) {
      l_k = 0;
      ++l_it;
      if (l_it < my_tiles) set_load_tile(l_it * G + slot_in_round);
    }
  };
#ifdef NVIT_PROBE_V2_NO_DMA   // timing probe (tools/probes/gemm_v2_bench.hip): only the first two stages are fetched
#define V2_LIVE (l_stage < 2)
#else
#define V2_LIVE true
#endif
#define V2_PIECE(p_)                                                                                     \
  if (V2_LIVE) {                                                                                         \
    if ((p_) < 8)                                                                                        \
      v2_glds(sA, voa[(p_) & 7], l_bo + (unsigned)(((p_) & 7) * 4096));                                  \
    else                                                                                                 \
      v2_glds(sB, vob[(p_) & 7], l_bo + (unsigned)(V2_A_BYTES + ((p_) & 7) * 4096));                     \
  }

  // The accumulators live in the accumulation half of the register file for the whole kernel.  The MFMAs are issued
  // from inline asm with "a" constraints: left to hipcc (builtin MFMA in a 512-register kernel), the allocator shuttles
  // the 256 accumulators between the two halves at every loop iteration and spills ~300 registers.
  f32x4 acc[2][8][4];   // [column half][m fragment][n fragment of the half]
#define V2_ZERO_ACC()                                                                                   \
  _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)     \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) acc[h_][i_][j_] = (f32x4){0.f, 0.f, 0.f, 0.f};
  V2_ZERO_ACC()

  // fragment addresses: row r of an operand tile, 16-byte chunk c -> r * 128 + ((c ^ (r & 7)) << 4); (r & 7) = (l15 & 7)
  const int a_row0 = wr * 128 + l15, b_row0 = wc * 128 + l15;
  const int sw = l15 & 7;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 fa0[8], fb0[8], fa1[8], fb1[8];
#define V2_RA(dst_, slot_, kk_, i_) \
  dst_[i_] = *reinterpret_cast<const u32x4*>(smem + (slot_) * V2_SLOT + (a_row0 + (i_) * 16) * ROWB + ((((kk_) * 4 + lg) ^ sw) << 4));
#define V2_RB(dst_, slot_, kk_, j_) \
  dst_[j_] = *reinterpret_cast<const u32x4*>(smem + (slot_) * V2_SLOT + V2_A_BYTES + (b_row0 + (j_) * 16) * ROWB + ((((kk_) * 4 + lg) ^ sw) << 4));
#ifdef V2_ASM_MFMA
#define V2_MFMA(acc_, a_, b_) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc_) : "v"(a_), "v"(b_));
#else
#define V2_MFMA(acc_, a_, b_) \
  acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc_, 0, 0, 0);
#endif
  // MFMA "A" operand = the B-matrix fragment (rows n), "B" operand = the A-matrix fragment (columns m): as gemm_p.hip
#define V2_MMA4(FA_, FB_, i_, jh_)                                                                        \
  _Pragma("unroll") for (int jj_ = 0; jj_ < 4; ++jj_)                                                     \
      V2_MFMA(acc[jh_][i_][jj_], FB_[(jh_) * 4 + jj_], FA_[i_])

  // ---- prologue: stages 0 and 1 in flight, wait for stage 0, read its k-half 0
  set_load_tile(slot_in_round);
  begin_stage();
#pragma unroll
  for (int p = 0; p < 16; ++p) V2_PIECE(p)
  end_stage();
  if (total > 1) {
    begin_stage();
#pragma unroll
    for (int p = 0; p < 16; ++p) V2_PIECE(p)
    end_stage();
    v2_wait_vmcnt<16>();
  } else {
    v2_wait_vmcnt<0>();
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) V2_RB(fb0, 0, 0, j)
#pragma unroll
  for (int i = 0; i < 8; ++i) V2_RA(fa0, 0, 0, i)

  bool stored_prev = false;   // the previous tile was a full one: its NST stores are younger than DMA(s+1)
  int s = 0;

  // One stage.  FAST_: steady state (stages s+1 and s+2 exist: no conditions inside the phases).
#define V2_STAGE(FAST_)                                                                                    \
  {                                                                                                        \
    const int slot = s & 1;                                                                                \
    /* ---------------- phase 0: k-half 0 of stage s from registers, k-half 1 of stage s from LDS */       \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    _Pragma("unroll") for (int b = 0; b < 16; ++b) {                                                       \
      if (b < 8) {                                                                                         \
        V2_RB(fb1, slot, 1, b)                                                                             \
      } else {                                                                                             \
        V2_RA(fa1, slot, 1, b - 8)                                                                         \
      }                                                                                                    \
      V2_MMA4(fa0, fb0, b >> 1, b & 1)                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
    }                                                                                                      \
    /* every LDS read of stage s is issued; retire them and this wave's DMA pieces of stage s+1, then the  \
       barrier: after it slot (s & 1) is free for stage s+2 and stage s+1 is complete in the other slot */ \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    if (FAST_ || s + 1 < total) {                                                                          \
      if (stored_prev)                                                                                     \
        v2_wait_vmcnt<NST>();                                                                              \
      else                                                                                                 \
        v2_wait_vmcnt<0>();                                                                                \
    }                                                                                                      \
    stored_prev = false;                                                                                   \
    __syncthreads();                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    /* ---------------- phase 1: k-half 1 of stage s; DMA of stage s+2; k-half 0 of stage s+1 */           \
    const bool issue = FAST_ || s + 2 < total;                                                             \
    const bool next = FAST_ || s + 1 < total;                                                              \
    if (issue) begin_stage();                                                                              \
    _Pragma("unroll") for (int b = 0; b < 16; ++b) {                                                       \
      if (next) {                                                                                          \
        if (b < 8) {                                                                                       \
          V2_RB(fb0, slot ^ 1, 0, b)                                                                       \
        } else {                                                                                           \
          V2_RA(fa0, slot ^ 1, 0, b - 8)                                                                   \
        }                                                                                                  \
      }                                                                                                    \
      if (issue) V2_PIECE(b)                                                                               \
      V2_MMA4(fa1, fb1, b >> 1, b & 1)                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
    }                                                                                                      \
    if (issue) end_stage();                                                                                \
    ++s;                                                                                                   \
  }
  // (two explicit epilogue calls: left as a loop over the halves, hipcc does not unroll it and the accumulators,
  //  indexed at run time, go to scratch)
#define V2_EPILOGUE(h_)                                                                    \
  {                                                                                        \
    const int mb = m0 + wr * 128, nb = n0 + wc * 128 + (h_) * 64;                          \
    if constexpr (EPI == 1)                                                                \
      nt_store_tile_staged<8, bf16>(g, acc[h_], mb, nb, lane, scratch);                    \
    else if constexpr (EPI == 2)                                                           \
      nt_store_tile_staged<8, float>(g, acc[h_], mb, nb, lane, scratch);                   \
    else if constexpr (EPI == 3)                                                           \
      nt_store_tile_swiglu<8>(g, acc[h_], mb, nb, lane, scratch);                          \
    else if constexpr (EPI == 4)                                                           \
      nt_store_tile_qknorm<8>(g, acc[h_], mb, nb, lane, scratch);                          \
    else if constexpr (EPI == 5)                                                           \
      nt_store_tile_swiglu_bwd<8>(g, acc[h_], mb, nb, lane, scratch);                      \
    else                                                                                   \
      nt_store_tile<8>(g, acc[h_], mb, nb, l15, lg);                                       \
  }
  // The K loop of a tile holds no epilogue code: with the epilogue inside the stage loop hipcc's allocator keeps the
  // loop-carried fragments in scratch for the whole kernel.  Every tile but the last runs the condition-free stage
  // body (its stages s+1, s+2 exist because nt >= 2, checked on the host).
  for (int it = 0; it < my_tiles; ++it) {
    if (it + 1 < my_tiles) {
      for (int k = 0; k < nt; ++k) V2_STAGE(true)
    } else {
      for (int k = 0; k < nt; ++k) V2_STAGE(false)
    }
    int m0, n0;
    tile_of(it * G + slot_in_round, m0, n0);
    const bool full_tile = NST > 0 && m0 + V2_BM <= g.M && n0 + V2_BN <= g.N;
    char* scratch = smem + 2 * V2_SLOT + wid * 2048;
    // the last four MFMAs wrote acc[1][7][*]: their results need the matrix pipe's write-back latency before a
    // vector instruction may read them (inline-asm MFMAs get no compiler-inserted wait states)
#ifdef V2_ASM_MFMA
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[1][7][0]), "+a"(acc[1][7][1]), "+a"(acc[1][7][2]), "+a"(acc[1][7][3]));
#endif
    V2_EPILOGUE(0)
    V2_EPILOGUE(1)
    V2_ZERO_ACC()
    stored_prev = full_tile;
  }
#undef V2_STAGE
#undef V2_EPILOGUE
#undef V2_ZERO_ACC
#undef V2_PIECE
#undef V2_RA
#undef V2_RB
#undef V2_MMA4
}

template <int EPI>
int launch_v2(const NtArgs& g_in, int n_cu, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_v2_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       V2_LDS);
    if (e != hipSuccess) NVIT_FAIL((int)e, "gemm_nt_v2: cannot raise LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  NtArgs g = g_in;
  g.tiles_n = cdiv(g.N, V2_BN);
  const int tiles_m = cdiv(g.M, V2_BM);
  hipLaunchKernelGGL((gemm_nt_v2_kernel<EPI>), dim3(n_cu), dim3(256), V2_LDS, s, g, tiles_m, tiles_m * g.tiles_n);
  NVIT_CHECK_LAUNCH("gemm_nt_v2");
  return NVIT_OK;
}

}  // namespace

// epi: 0 = choose by output type (1 staged bf16 / 2 staged fp32 / generic when the row pitch is not 16-byte friendly),
// 3 / 4 / 5 = the fused epilogues.  bf16 operands only; K a multiple of 64; operand panels below 4 GiB.
int nvit_gemm_nt_v2_launch(const NtArgs& g, int epi, hipStream_t s) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int devid = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&devid) != hipSuccess || hipGetDeviceProperties(&prop, devid) != hipSuccess)
      NVIT_FAIL(NVIT_EINVAL, "gemm_nt_v2: cannot query device properties");
    n_cu = prop.multiProcessorCount;
    n_cu -= n_cu % 8;
    if (n_cu < 8) n_cu = 8;
  }
  NVIT_REQUIRE(g.K % 64 == 0 && g.K >= 128, "gemm_nt_v2: K must be a multiple of 64, at least 128 (got %d)", g.K);
  NVIT_REQUIRE((unsigned long long)g.M * g.lda * 2ull < (1ull << 32) && (unsigned long long)g.N * g.ldb * 2ull < (1ull << 32),
               "gemm_nt_v2: operand panels must be below 4 GiB");
  if (epi == 3) return launch_v2<3>(g, n_cu, s);
  if (epi == 4) return launch_v2<4>(g, n_cu, s);
  if (epi == 5) return launch_v2<5>(g, n_cu, s);
  const int eo = g.out_dt == NVIT_F32 ? 4 : 8;
  const bool staged = (g.N % eo) == 0 && (g.ldc % eo) == 0;
  if (!staged) return launch_v2<0>(g, n_cu, s);
  return g.out_dt == NVIT_F32 ? launch_v2<2>(g, n_cu, s) : launch_v2<1>(g, n_cu, s);
}
