// Probe: the fused-epilogue GEMMs (EPI 3 c_fc + SwiGLU, EPI 5 mlp_c_proj dgrad + SwiGLU backward) on the two-workgroups-
// per-CU kernel (gemm_pair.hip) against the product kernel (gemm_p.hip): these two spend 25-50 % of a tile in an
// epilogue that one workgroup per CU cannot overlap with MFMAs.  Times both, compares the outputs bit for bit.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/gemm_pair_fused.hip nvit_amd/csrc/core.hip -o tools/probes/bin/gemm_pair_fused
#include "../../nvit_amd/csrc/gemm_p.hip"
#include "gemm_pair.hip"
#include <vector>

int main() {
  const int M = 100352, C = 768, F = 3072;
  std::vector<uint16_t> h((size_t)M * 2 * F);
  unsigned x = 12345u;
  for (auto& v : h) {   // random bf16 in (-1, 1)
    x = x * 1664525u + 1013904223u;
    v = (uint16_t)(((x >> 31) << 15) | ((0x78 + ((x >> 8) & 7)) << 7) | ((x >> 16) & 0x7f));
  }
  std::vector<float> hs(2 * F);
  for (int i = 0; i < 2 * F; ++i) hs[i] = 0.9f + 0.2f * (float)(i % 17) / 17.0f;
  char *X, *W, *UV, *XM, *DUV, *UV2, *XM2, *DUV2;
  float *gs, *part, *part2;
  const size_t uvb = (size_t)M * 2 * F * 2, xmb = (size_t)M * F * 2, partb = (size_t)2 * (M / 256) * 2 * F * 4;
  (void)hipMalloc(&X, (size_t)M * F * 2);
  (void)hipMalloc(&W, (size_t)2 * F * F * 2);
  (void)hipMalloc(&UV, uvb); (void)hipMalloc(&UV2, uvb);
  (void)hipMalloc(&XM, xmb); (void)hipMalloc(&XM2, xmb);
  (void)hipMalloc(&DUV, uvb); (void)hipMalloc(&DUV2, uvb);
  (void)hipMalloc(&gs, 2 * F * 4); (void)hipMalloc(&part, partb); (void)hipMalloc(&part2, partb);
  (void)hipMemcpy(X, h.data(), (size_t)M * F * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, h.data() + 777, (size_t)2 * F * F * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(gs, hs.data(), 2 * F * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto args3 = [&](char* uv, char* xm) {
    NtArgs g{};
    g.A = X; g.B = W; g.C = uv; g.M = M; g.N = 2 * F; g.K = C; g.lda = C; g.ldb = C; g.ldc = 2 * F; g.out_dt = NVIT_BF16;
    g.xm = xm; g.ld_xm = F; g.gs = gs; g.gscale = 0.05f;
    return g;
  };
  auto args5 = [&](const char* uv, char* duv, float* pt) {
    NtArgs g{};
    g.A = X; g.B = W; g.C = duv; g.M = M; g.N = F; g.K = C; g.lda = C; g.ldb = C; g.ldc = 2 * F; g.out_dt = NVIT_BF16;
    g.uv_in = uv; g.ld_uv = 2 * F; g.Fh = F; g.gs = gs; g.gscale = 0.05f; g.part = pt;
    return g;
  };
  auto timeit = [&](const char* name, auto fn, double flops) {
    for (int i = 0; i < 3; ++i) fn();
    (void)hipEventRecord(e0, 0);
    const int it = 20;
    for (int i = 0; i < it; ++i) fn();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.1f us  %7.1f TF/s\n", name, ms / it * 1e3, flops / (ms / it) / 1e9);
    fflush(stdout);
  };
  auto differ = [&](const char* name, const void* a, const void* b, size_t bytes) {
    std::vector<unsigned char> ha(bytes), hb(bytes);
    (void)hipMemcpy(ha.data(), a, bytes, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hb.data(), b, bytes, hipMemcpyDeviceToHost);
    size_t nd = 0;
    for (size_t i = 0; i < bytes; ++i) nd += ha[i] != hb[i];
    printf("  %-30s %zu differing bytes of %zu\n", name, nd, bytes);
  };
  for (int round = 0; round < 2; ++round) {
    timeit("EPI3 product (1 WG/CU, 256x256)", [&] { nvit_gemm_nt_fused_launch(args3(UV, XM), 3, 0); }, 2.0 * M * 2 * F * C);
    timeit("EPI3 pair    (2 WG/CU, 256x128)", [&] { nvit_gemm_nt_pair_launch(args3(UV2, XM2), 3, 0); }, 2.0 * M * 2 * F * C);
    timeit("EPI5 product (1 WG/CU, 256x256)", [&] { nvit_gemm_nt_fused_launch(args5(UV, DUV, part), 5, 0); }, 2.0 * M * F * C);
    timeit("EPI5 pair    (2 WG/CU, 256x128)", [&] { nvit_gemm_nt_pair_launch(args5(UV, DUV2, part2), 5, 0); }, 2.0 * M * F * C);
  }
  (void)hipDeviceSynchronize();
  differ("EPI3 uv", UV, UV2, uvb / 16);
  differ("EPI3 xm", XM, XM2, xmb / 16);
  differ("EPI5 duv", DUV, DUV2, uvb / 16);
  differ("EPI5 part", part, part2, partb);
  return 0;
}
