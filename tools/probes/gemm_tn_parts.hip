// Probe: the persistent weight-gradient (TN) GEMM without its LDS-DMA after the ring is primed (NVIT_PROBE_NO_DMA) and
// without its fragment reads + MFMAs (NVIT_PROBE_NO_MFMA), on the Base block shapes.  Only the times mean something.
//   build: tools/probes/build_gemm_parts.sh     run: gemm_tn_parts_{full,nodma,nomfma}
#include "../../nvit_amd/csrc/gemm_tn_p.hip"
#include <vector>

int main() {
  const int M = 100352;
  struct Shape { int N, K; const char* name; };
  const Shape shapes[] = {{768, 768, "o-proj wgrad   N=768  K=768 "}, {2304, 768, "qkv wgrad      N=2304 K=768 "},
                          {6144, 768, "c_fc wgrad     N=6144 K=768 "}, {768, 3072, "mlp_c_proj wg  N=768  K=3072"}};
  std::vector<uint16_t> h((size_t)M * 6144);
  unsigned x = 12345u;
  for (auto& v : h) {
    x = x * 1664525u + 1013904223u;
    v = (uint16_t)(((x >> 31) << 15) | ((0x78 + ((x >> 8) & 7)) << 7) | ((x >> 16) & 0x7f));
  }
  char *A, *B;
  float *ws, *zeros;
  (void)hipMalloc(&A, (size_t)M * 6144 * 2);
  (void)hipMalloc(&B, (size_t)M * 3072 * 2);
  (void)hipMalloc(&ws, (size_t)64 * 6144 * 768 * 4);
  (void)hipMalloc(&zeros, 4096);
  (void)hipMemset(zeros, 0, 4096);
  (void)hipMemcpy(A, h.data(), (size_t)M * 6144 * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, h.data(), (size_t)M * 3072 * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (const Shape& sh : shapes) {
    const int tiles = (sh.N / 256) * (sh.K / 256);
    int splits = 256 / tiles;
    if (splits < 1) splits = 1;
    auto launch = [&]() { return nvit_gemm_tn_persistent_launch(NVIT_BF16, A, sh.N, B, sh.K, ws, zeros, M, sh.N, sh.K, splits, 0); };
    for (int i = 0; i < 3; ++i) launch();
    (void)hipDeviceSynchronize();
    const int reps = 20;
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-30s splits %2d  %8.1f us  %7.1f TF/s-equivalent\n", sh.name, splits, us, 2.0 * M * sh.N * sh.K / (us * 1e-6) / 1e12);
  }
  return 0;
}
