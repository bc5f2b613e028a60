// Probe: what bounds the persistent NT GEMM main loop?  The product kernel (nvit_amd/csrc/gemm_p.hip, included as is)
// is built four times - whole, without the LDS-DMA after the ring is primed (NVIT_PROBE_NO_DMA: MFMA + LDS reads +
// barriers + epilogue), without fragment reads and MFMAs (NVIT_PROBE_NO_MFMA: the operand feed + epilogue alone) and
// without the epilogue stores (NVIT_PROBE_NO_EPI) - and timed on the Base block shapes.  Results of the cut-down
// builds are garbage by construction; only the times mean something.
//   build: see tools/probes/build_gemm_parts.sh     run: gemm_parts_{full,nodma,nomfma,noepi}
#include "../../nvit_amd/csrc/gemm_p.hip"
#include "gemm_pair.hip"
#include <vector>

static int launch(const NtArgs& g) {
  static const bool pair = getenv("PAIR") != nullptr;   // PAIR=1: the two-workgroups-per-CU kernel (gemm_pair.hip)
  if (pair) return nvit_gemm_nt_pair_launch(g, g.out_dt == NVIT_F32 ? 2 : 1, 0);
  return nvit_gemm_nt_persistent_launch(NVIT_BF16, g, 256, 0);
}

int main() {
  const int M = 100352;
  struct Shape { int N, K, out_dt; const char* name; };
  const Shape shapes[] = {{768, 768, NVIT_F32, "o-proj      N=768  K=768  f32"},
                          {768, 3072, NVIT_F32, "mlp_c_proj  N=768  K=3072 f32"},
                          {768, 6144, NVIT_F32, "c_fc dgrad  N=768  K=6144 f32"},
                          {2304, 768, NVIT_BF16, "qkv-like    N=2304 K=768  bf16"},
                          {6144, 768, NVIT_BF16, "c_fc-like   N=6144 K=768  bf16"}};
  std::vector<uint16_t> h((size_t)M * 6144);
  unsigned x = 12345u;
  for (auto& v : h) {   // random bf16 in (-1, 1): sign + exponent 0x3c..0x3f + mantissa
    x = x * 1664525u + 1013904223u;
    v = (uint16_t)(((x >> 31) << 15) | ((0x78 + ((x >> 8) & 7)) << 7) | ((x >> 16) & 0x7f));
  }
  char *A, *B, *C;
  (void)hipMalloc(&A, (size_t)M * 6144 * 2);
  (void)hipMalloc(&B, (size_t)6144 * 6144 * 2);
  (void)hipMalloc(&C, (size_t)M * 6144 * 4);
  (void)hipMemcpy(A, h.data(), (size_t)M * 6144 * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, h.data(), (size_t)6144 * 6144 * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (const Shape& sh : shapes) {
    NtArgs g = {};
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = sh.N; g.K = sh.K;
    g.lda = sh.K; g.ldb = sh.K; g.ldc = sh.N;
    g.out_dt = sh.out_dt;
    for (int i = 0; i < 3; ++i) launch(g);
    (void)hipDeviceSynchronize();
    const int reps = 20;
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch(g);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, tf = 2.0 * M * sh.N * sh.K / (us * 1e-6) / 1e12;
    const int tiles = (M / 256) * (sh.N / 256);
    printf("%-34s %8.1f us  %7.1f TF/s-equivalent   %.2f us per tile-round (%d tiles on 256 CUs)\n", sh.name, us, tf,
           us / ((tiles + 255) / 256), tiles);
  }
  return 0;
}
