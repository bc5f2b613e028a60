// Probe: can two 256-thread workgroups with 80 KiB of LDS each share one gfx950 CU?  (grid = 2 x #CU, each WG spins
// ~100 us; the launch takes ~100 us when both fit, ~200 us when they serialise.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void spin(long long ticks, int* out) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = (char)threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0) out[blockIdx.x] = smem[3];
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  int rate = 0;
  hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
  int* out;
  hipMalloc(&out, 4096 * sizeof(int));
  for (int lds : {65536, 73728, 80 * 1024, 81920 - 1024, 81920 - 2048}) {
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const long long ticks = (long long)rate * 100 / 1000;  // 100 us
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t a, b;
      hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(spin, dim3(2 * p.multiProcessorCount), dim3(256), lds, 0, ticks, out);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep) printf("lds=%d B  grid=%d  time=%.1f us (%s)\n", lds, 2 * p.multiProcessorCount, ms * 1e3,
                      ms < 0.15 ? "2 WG/CU" : "serialised");
    }
  }
  return 0;
}
