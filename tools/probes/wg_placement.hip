// Probe: which workgroups of a 2-per-CU persistent grid share a CU?  Prints, per CU, the blockIdx values it hosted.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void where(unsigned* out, long long ticks) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = 1;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
  }
}
int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  int rate = 0;
  (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  const int G = 2 * p.multiProcessorCount;
  unsigned* out;
  (void)hipMalloc(&out, 2 * G * sizeof(unsigned));
  (void)hipFuncSetAttribute((const void*)where, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
  hipLaunchKernelGGL(where, dim3(G), dim3(256), 81920, 0, out, (long long)rate * 50 / 1000);
  (void)hipDeviceSynchronize();
  std::vector<unsigned> h(2 * G);
  (void)hipMemcpy(h.data(), out, 2 * G * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<int>> cu;
  for (int b = 0; b < G; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    const unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu_id].push_back(b);
  }
  printf("distinct CUs: %zu\n", cu.size());
  int shown = 0, pairs_half = 0, pairs_other = 0;
  for (auto& kv : cu) {
    if (kv.second.size() == 2 && kv.second[1] - kv.second[0] == G / 2) ++pairs_half; else ++pairs_other;
    if (shown++ < 12) {
      printf("xcc %llu se %llu cu %llu:", kv.first >> 16, (kv.first >> 8) & 0xff, kv.first & 0xf);
      for (int b : kv.second) printf(" %d", b);
      printf("\n");
    }
  }
  printf("CUs whose two WGs are (b, b+G/2): %d, other: %d\n", pairs_half, pairs_other);
  return 0;
}
