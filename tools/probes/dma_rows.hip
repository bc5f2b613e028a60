// Probe: how fast can a CU pull GEMM operand tiles into LDS by LDS-DMA, as a function of the row slice per stage
// (128 B = 64 bf16 of K per row, or 64 B = 32 bf16 with consecutive stages reading the two halves of each 128-B line),
// workgroups per CU (1 x 512 threads or 2 x 256 threads) and stages in flight?  Operand model: an A panel streamed once
// from HBM (each workgroup its own 256 rows, K bytes long) and a B panel shared by all workgroups (L2 resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_off);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(m) : "memory");
}

// ROWB: bytes of each row fetched per stage; AROWS / BROWS: rows of the two operands per stage; NW: waves per workgroup;
// AHEAD: stages in flight beyond the one being "consumed" (ring of AHEAD + 1 slots)
template <int ROWB, int AROWS, int BROWS, int NW, int AHEAD>
__global__ __launch_bounds__(NW * 64) void pull(const char* A, const char* B, size_t lda, size_t ldb, int kbytes, int tiles,
                                                int amod, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SLOT = (AROWS + BROWS) * ROWB;
  constexpr int RPI = 1024 / ROWB;                 // rows per wave-instruction
  constexpr int AI = AROWS / RPI / NW, BI = BROWS / RPI / NW;   // instructions per wave per stage
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)smem);
  const int lrow = lane / (ROWB / 16), lch = lane % (ROWB / 16);
  const int nst = kbytes / ROWB;
  int slot = 0, inflight = 0;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const char* a0 = A + (size_t)(t % amod) * AROWS * lda;   // amod small: the A tiles stay L2 resident
    for (int s = 0; s < nst; ++s) {
      const unsigned bo = lds_base + slot * SLOT + wid * 1024;
#pragma unroll
      for (int i = 0; i < AI; ++i)
        glds16(a0 + (size_t)((i * NW + wid) * RPI + lrow) * lda + (size_t)s * ROWB + lch * 16, bo + i * NW * 1024);
#pragma unroll
      for (int i = 0; i < BI; ++i)
        glds16(B + (size_t)((i * NW + wid) * RPI + lrow) * ldb + (size_t)s * ROWB + lch * 16,
               bo + AROWS * ROWB + i * NW * 1024);
      slot = slot == AHEAD ? 0 : slot + 1;
      if (inflight < AHEAD) {
        ++inflight;
      } else {
        if constexpr (AHEAD == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (AHEAD == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + BI) : "memory");
        if constexpr (AHEAD == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AI + BI)) : "memory");
        __syncthreads();
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = *(unsigned*)(smem + 64);
}

template <int ROWB, int AROWS, int BROWS, int NW, int AHEAD>
void run(const char* name, const char* A, const char* B, int K, int M, int wg_per_cu, unsigned* sink, int amod = 1 << 30) {
  constexpr int LDS = (AROWS + BROWS) * ROWB * (AHEAD + 1);
  auto fn = pull<ROWB, AROWS, BROWS, NW, AHEAD>;
  (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  const int tiles = M / AROWS, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), LDS, 0, A, B, (size_t)K * 2, (size_t)K * 2, K * 2, tiles, amod, sink);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double bytes = (double)tiles * (AROWS + BROWS) * K * 2;
  printf("%-58s K=%5d  %8.1f us  %6.2f TB/s into LDS  = %5.1f B/clk/CU @2.4GHz  (LDS %d KiB/WG)\n", name, K, best * 1e3,
         bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 256 / 2.4e9, LDS / 1024);
}

// The real tile walk of gemm_p.hip (256x256 tiles, 8 m-tiles x tiles_n grouped order, XCD-contiguous deal), feed only:
// every workgroup pulls the A and B slabs of its tiles, nothing else.  AHEAD as above.
template <int AHEAD>
__global__ __launch_bounds__(512) void pull_gemm(const char* A, const char* B, int M, int N, int K, int ld, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SLOT = 65536;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)smem);
  const int srow = lane >> 3, gc = (lane & 7) ^ srow;
  const int tiles_m = M / 256, tiles_n = N / 256, ntiles = tiles_m * tiles_n, nst = K / 64, G = gridDim.x;
  const int slot_in_round = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  int slot = 0, inflight = 0;
  for (int pid = slot_in_round; pid < ntiles; pid += G) {
    const int per_group = 8 * tiles_n, group = pid / per_group, first_m = group * 8;
    const int gsz = (tiles_m - first_m) < 8 ? (tiles_m - first_m) : 8;
    const int in_g = pid - group * per_group;
    const int m0 = (first_m + in_g % gsz) * 256, n0 = (in_g / gsz) * 256;
    for (int s = 0; s < nst; ++s) {
      const unsigned bo = lds_base + slot * SLOT + wid * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        glds16(A + ((size_t)(m0 + (i * 8 + wid) * 8 + srow) * ld + gc * 8) * 2 + (size_t)s * 128, bo + i * 8192);
        glds16(B + ((size_t)(n0 + (i * 8 + wid) * 8 + srow) * ld + gc * 8) * 2 + (size_t)s * 128, bo + 32768 + i * 8192);
      }
      slot = slot == AHEAD ? 0 : slot + 1;
      if (inflight < AHEAD) {
        ++inflight;
      } else {
        if constexpr (AHEAD == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (AHEAD == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = *(unsigned*)(smem + 64);
}

template <int AHEAD>
void run_gemm(const char* A, const char* B, int M, int N, int K, unsigned* sink, int pad = 0) {
  constexpr int LDS = 65536 * (AHEAD + 1);
  auto fn = pull_gemm<AHEAD>;
  (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fn, dim3(256), dim3(512), LDS, 0, A, B, M, N, K, K + pad, sink);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double bytes = (double)(M / 256) * (N / 256) * 512.0 * K * 2;
  printf("GEMM walk N=%5d K=%5d ld=K+%3d, %d stage(s) in flight: %8.1f us  %6.2f TB/s into LDS = %5.1f B/clk/CU  (MFMA time at 2516 TF/s: %.1f us)\n",
         N, K, pad, AHEAD + 1, best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 256 / 2.4e9,
         2.0 * M * N * K / 2516.6e12 * 1e6);
}

int main() {
  const int M = 100352;
  char *A, *B;
  unsigned* sink;
  (void)hipMalloc(&A, (size_t)M * 6500 * 2);
  (void)hipMalloc(&B, (size_t)6144 * 6500 * 2);
  (void)hipMalloc(&sink, 4096);
  (void)hipMemset(A, 1, (size_t)M * 6500 * 2);
  (void)hipMemset(B, 1, (size_t)6144 * 6500 * 2);
  for (int K : {768, 3072, 6144}) {
    run_gemm<0>(A, B, M, 768, K, sink);
    run_gemm<1>(A, B, M, 768, K, sink);
  }
  for (int pad : {64, 128, 192, 320}) {
    run_gemm<1>(A, B, M, 768, 3072, sink, pad);
    run_gemm<1>(A, B, M, 768, 6144, sink, pad);
  }
  run_gemm<1>(A, B, M, 768, 768, sink, 64);
  run_gemm<1>(A, B, M, 2304, 768, sink);
  run_gemm<1>(A, B, M, 6144, 768, sink);
  for (int K : {3072}) {
    run<128, 256, 256, 8, 0>("1 WG x 8 waves, 256+256 rows x 128 B, 1 stage in flight", A, B, K, M, 1, sink);
    run<128, 256, 256, 8, 1>("1 WG x 8 waves, 256+256 rows x 128 B, 2 stages in flight", A, B, K, M, 1, sink);
    run<128, 256, 128, 4, 0>("2 WG x 4 waves, 256+128 rows x 128 B, 1 stage in flight", A, B, K, M, 2, sink);
    run<64, 256, 128, 4, 0>("2 WG x 4 waves, 256+128 rows x  64 B, 1 stage in flight", A, B, K, M, 2, sink);
    run<64, 256, 128, 4, 1>("2 WG x 4 waves, 256+128 rows x  64 B, 2 stages in flight", A, B, K, M, 2, sink);
    run<64, 256, 128, 4, 2>("2 WG x 4 waves, 256+128 rows x  64 B, 3 stages in flight", A, B, K, M, 2, sink);
    run<64, 256, 256, 8, 2>("1 WG x 8 waves, 256+256 rows x  64 B, 3 stages in flight", A, B, K, M, 1, sink);
    run<128, 256, 256, 8, 1>("L2-resident A (8 tiles): 1 WG x 8 waves, 128 B rows, 2 in flight", A, B, K, M, 1, sink, 8);
    run<128, 256, 256, 8, 1>("L2-resident A (1 tile) : 1 WG x 8 waves, 128 B rows, 2 in flight", A, B, K, M, 1, sink, 1);
    run<128, 256, 256, 8, 1>("A: 64 tiles (8 per XCD): 1 WG x 8 waves, 128 B rows, 2 in flight", A, B, K, M, 1, sink, 64);
    run<64, 256, 128, 4, 2>("L2-resident A (8 tiles): 2 WG x 4 waves,  64 B rows, 3 in flight", A, B, K, M, 2, sink, 8);
  }
  return 0;
}
