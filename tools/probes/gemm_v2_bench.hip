// Probe + cross-check: the one-wave-per-SIMD NT GEMM (nvit_amd/csrc/gemm_v2.hip) against the 8-wave persistent kernel
// (gemm_p.hip) on the Base block shapes (M = 100 352): interleaved timing rounds in one process and a bit-for-bit
// comparison of the outputs (same K order, same MFMA instruction: they must be identical).
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/gemm_v2_bench.hip nvit_amd/csrc/core.hip -o tools/probes/bin/gemm_v2_bench
#include "../../nvit_amd/csrc/gemm_p.hip"
#include "gemm_v2.hip"
#include <vector>
#include <algorithm>
#include <string.h>

struct Shape { int N, K, out_dt, epi, acc; const char* name; };

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 100352;
  const Shape shapes[] = {{768, 768, NVIT_F32, 0, 0, "o-proj      N=768  K=768  f32 "},
                          {768, 768, NVIT_BF16, 0, 0, "o-dgrad     N=768  K=768  bf16"},
                          {768, 3072, NVIT_F32, 0, 0, "mlp_c_proj  N=768  K=3072 f32 "},
                          {768, 6144, NVIT_F32, 0, 1, "c_fc dgrad  N=768  K=6144 f32+="},
                          {768, 2304, NVIT_F32, 0, 1, "qkv dgrad   N=768  K=2304 f32+="},
                          {2304, 768, NVIT_BF16, 4, 0, "qkv EPI4    N=2304 K=768      "},
                          {6144, 768, NVIT_BF16, 3, 0, "c_fc EPI3   N=6144 K=768      "},
                          {3072, 768, NVIT_BF16, 5, 0, "p-dgrad EPI5 N=3072 K=768     "},
                          {6144, 768, NVIT_BF16, 0, 0, "plain bf16  N=6144 K=768      "}};
  std::vector<uint16_t> h((size_t)M * 6144);
  unsigned x = 12345u;
  for (auto& v : h) {
    x = x * 1664525u + 1013904223u;
    v = (uint16_t)(((x >> 31) << 15) | ((0x78 + ((x >> 8) & 7)) << 7) | ((x >> 16) & 0x7f));
  }
  char *A, *B, *C0, *C1, *X0, *X1, *UV;
  float *gs, *part0, *part1, *rq, *rk, *sqk;
  const size_t cbytes = (size_t)M * 6144 * 4;
  (void)hipMalloc(&A, (size_t)M * 6144 * 2);
  (void)hipMalloc(&B, (size_t)6144 * 6144 * 2);
  (void)hipMalloc(&C0, cbytes);
  (void)hipMalloc(&C1, cbytes);
  (void)hipMalloc(&X0, (size_t)M * 3072 * 2);
  (void)hipMalloc(&X1, (size_t)M * 3072 * 2);
  (void)hipMalloc(&UV, (size_t)M * 6144 * 2);
  (void)hipMalloc(&gs, 6144 * 4);
  (void)hipMalloc(&sqk, 6144 * 4);
  (void)hipMalloc(&part0, (size_t)2 * (M / 128 + 2) * 6144 * 4);
  (void)hipMalloc(&part1, (size_t)2 * (M / 128 + 2) * 6144 * 4);
  (void)hipMalloc(&rq, (size_t)M * 64 * 4);
  (void)hipMalloc(&rk, (size_t)M * 64 * 4);
  (void)hipMemcpy(A, h.data(), (size_t)M * 6144 * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, h.data() + 777, (size_t)6144 * 6144 * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(UV, h.data() + 4242, (size_t)M * 6144 * 2 - 8484, hipMemcpyHostToDevice);
  {
    std::vector<float> f(6144);
    for (int i = 0; i < 6144; ++i) f[i] = 0.5f + (i % 7) * 0.1f;
    (void)hipMemcpy(gs, f.data(), 6144 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(sqk, f.data(), 6144 * 4, hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  std::vector<char> h0, h1;
  for (const Shape& sh : shapes) {
    auto mk = [&](char* C, char* X, float* part) {
      NtArgs g = {};
      g.A = A; g.B = B; g.C = C;
      g.M = M; g.N = sh.N; g.K = sh.K;
      g.lda = sh.K; g.ldb = sh.K; g.ldc = sh.N;
      g.out_dt = sh.out_dt;
      g.accumulate = sh.acc;
      if (sh.epi == 3) { g.xm = X; g.ld_xm = sh.N / 2; g.gs = gs; g.gscale = 1.7f; }
      if (sh.epi == 4) {
        g.q_prescale = 11.54f; g.qh = C; g.kh = C + (size_t)M * 768 * 2; g.vh = C + (size_t)M * 768 * 4;
        g.rq = rq; g.rk = rk; g.sqk = sqk; g.c_q = 32.f; g.part0 = 0; g.Cemb = 768; g.Ttok = 784; g.H = 12;
      }
      if (sh.epi == 5) { g.ldc = 2 * sh.N; g.uv_in = UV; g.ld_uv = 2 * sh.N; g.Fh = sh.N; g.gs = gs; g.gscale = 1.7f; g.part = part; }
      return g;
    };
    auto run = [&](int which, const NtArgs& g) {
      if (which == 1) return nvit_gemm_nt_v2_launch(g, sh.epi, 0);
      if (sh.epi) return nvit_gemm_nt_fused_launch(g, sh.epi, 0);
      return nvit_gemm_nt_persistent_launch(NVIT_BF16, g, 256, 0);
    };
    const NtArgs g0 = mk(C0, X0, part0), g1 = mk(C1, X1, part1);
    // correctness: identical start state, one launch each, compare every output byte
    const size_t out_bytes = sh.epi == 4 ? (size_t)M * 2304 * 2 : (size_t)M * (sh.epi == 5 ? 2 * sh.N : sh.N) * (sh.out_dt == NVIT_F32 ? 4 : 2);
    (void)hipMemset(C0, 0x11, out_bytes);
    (void)hipMemset(C1, 0x11, out_bytes);
    if (sh.acc) { (void)hipMemcpy(C0, A, out_bytes, hipMemcpyDeviceToDevice); (void)hipMemcpy(C1, A, out_bytes, hipMemcpyDeviceToDevice); }
    int rc0 = run(0, g0), rc1 = run(1, g1);
    (void)hipDeviceSynchronize();
    if (rc0 || rc1) { printf("%s launch failed %d %d: %s\n", sh.name, rc0, rc1, nvit_last_error()); continue; }
    h0.resize(out_bytes); h1.resize(out_bytes);
    (void)hipMemcpy(h0.data(), C0, out_bytes, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h1.data(), C1, out_bytes, hipMemcpyDeviceToHost);
    size_t diff = 0;
    for (size_t i = 0; i < out_bytes; ++i) diff += h0[i] != h1[i];
    if (diff && !sh.epi) {   // where do the two kernels disagree?  (element coordinates, tile, position inside the tile)
      const int es = sh.out_dt == NVIT_F32 ? 4 : 2;
      size_t shown = 0, nel = 0;
      std::vector<int> per_tile((M / 256) * (sh.N / 256), 0);
      for (size_t e = 0; e < out_bytes / es; ++e) {
        if (memcmp(&h0[e * es], &h1[e * es], es) == 0) continue;
        const int row = (int)(e / sh.N), col = (int)(e % sh.N);
        ++nel;
        ++per_tile[(row / 256) * (sh.N / 256) + col / 256];
        if (shown < 12) {
          float a, b;
          if (es == 4) { memcpy(&a, &h0[e * 4], 4); memcpy(&b, &h1[e * 4], 4); }
          else { unsigned ua = (unsigned)(*(uint16_t*)&h0[e * 2]) << 16, ub = (unsigned)(*(uint16_t*)&h1[e * 2]) << 16; memcpy(&a, &ua, 4); memcpy(&b, &ub, 4); }
          printf("   diff at row %d col %d (tile %d,%d; in-tile %d,%d): p %.6g  v2 %.6g\n", row, col, row / 256, col / 256, row % 256, col % 256, a, b);
          ++shown;
        }
      }
      int ntl = 0, mx = 0;
      for (int v : per_tile) { ntl += v > 0; mx = v > mx ? v : mx; }
      printf("   %zu differing elements in %d of %zu tiles (max %d per tile)\n", nel, ntl, per_tile.size(), mx);
    }
    if (sh.epi == 3) {
      const size_t xb = (size_t)M * (sh.N / 2) * 2;
      h0.resize(xb); h1.resize(xb);
      (void)hipMemcpy(h0.data(), X0, xb, hipMemcpyDeviceToHost);
      (void)hipMemcpy(h1.data(), X1, xb, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < xb; ++i) diff += h0[i] != h1[i];
    }
    if (sh.epi == 5) {
      const size_t pb = (size_t)2 * (M / 256) * 2 * sh.N * 4;
      h0.resize(pb); h1.resize(pb);
      (void)hipMemcpy(h0.data(), part0, pb, hipMemcpyDeviceToHost);
      (void)hipMemcpy(h1.data(), part1, pb, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < pb; ++i) diff += h0[i] != h1[i];
    }
    // timing: interleaved rounds
    const int rounds = 5, reps = 6;
    std::vector<float> t[2];
    for (int r = 0; r < rounds; ++r)
      for (int w = 0; w < 2; ++w) {
        const NtArgs& g = w ? g1 : g0;
        run(w, g);
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) run(w, g);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        t[w].push_back(ms * 1e3f / reps);
      }
    for (int w = 0; w < 2; ++w) std::sort(t[w].begin(), t[w].end());
    const double fl = 2.0 * M * sh.N * (double)sh.K;
    printf("%-32s p %8.1f us %7.1f TF/s | v2 %8.1f us %7.1f TF/s | v2/p time %.3f | differing bytes %zu\n", sh.name,
           t[0][rounds / 2], fl / t[0][rounds / 2] * 1e-6, t[1][rounds / 2], fl / t[1][rounds / 2] * 1e-6,
           t[1][rounds / 2] / t[0][rounds / 2], diff);
    fflush(stdout);
  }
  return 0;
}
