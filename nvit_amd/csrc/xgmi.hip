// Gradient all-reduce as direct peer reads over xGMI (reference intent nvit/train.py:438-446).
//
// MI355X nodes are a fully connected mesh: every GPU has one xGMI link to each of its 7 peers.  A ring all-reduce sends
// 2(N-1)/N of the buffer over ONE link per direction; the direct form below uses all 7 links of a GPU at once:
//   reduce-scatter: rank r owns chunk r (1/N of the flat gradient buffer).  It reads that chunk from every rank's buffer
//                   (its own from HBM, the other N-1 straight over their links, S/N bytes per link), sums the N values in
//                   rank order 0..N-1 (fixed order: every element is summed by exactly one rank, so after the gather all
//                   replicas are bit-identical), scales, and writes the result into its own buffer's chunk r;
//   all-gather    : rank r copies chunk j from rank j's buffer into its own, for every j != r (again S/N per link).
// Both are plain grid-stride kernels of 16-byte accesses with all N peer loads of an element group in flight together.
// The buffers are "symmetric": the same layout on every rank, exported once as IPC handles (host side:
// nvit_amd/xgmi.py).  Phase separation (all gradients written -> reduce-scatter -> all-gather -> buffers reusable) is the
// caller's job; the Python wrapper uses stream synchronisation + a host barrier, which is correct everywhere and is what
// could be verified without a multi-GPU node (ranks sharing one device).
#include "common.h"

namespace {

constexpr int XGMI_MAX_RANKS = 8;

struct XgmiPeers {
  float* p[XGMI_MAX_RANKS];
};

__global__ __launch_bounds__(256) void xgmi_reduce_scatter_kernel(XgmiPeers peers, int nranks, int rank, long long c0,
                                                                  long long c1, float scale) {
  // elements [c0, c1) of the flat buffer (c0, c1 multiples of 4)
  const long long n4 = (c1 - c0) >> 2;
  float* out = peers.p[rank] + c0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v[XGMI_MAX_RANKS];
#pragma unroll
    for (int r = 0; r < XGMI_MAX_RANKS; ++r)
      if (r < nranks) v[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(peers.p[r] + c0) + i);
    f32x4 acc = v[0];
#pragma unroll
    for (int r = 1; r < XGMI_MAX_RANKS; ++r)
      if (r < nranks) acc += v[r];
    reinterpret_cast<f32x4*>(out)[i] = acc * scale;
  }
}

__global__ __launch_bounds__(256) void xgmi_all_gather_kernel(XgmiPeers peers, int nranks, int rank, long long chunk,
                                                              long long n) {
  float* mine = peers.p[rank];
  for (int j = 0; j < nranks; ++j) {
    if (j == rank) continue;
    const long long c0 = (long long)j * chunk, c1 = c0 + chunk < n ? c0 + chunk : n;
    if (c1 <= c0) continue;
    const long long n4 = (c1 - c0) >> 2;
    const f32x4* src = reinterpret_cast<const f32x4*>(peers.p[j] + c0);
    f32x4* dst = reinterpret_cast<f32x4*>(mine + c0);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
      dst[i] = __builtin_nontemporal_load(src + i);
  }
}

int fill_peers(const int64_t* peer_ptrs, int nranks, XgmiPeers& p) {
  for (int r = 0; r < XGMI_MAX_RANKS; ++r) p.p[r] = r < nranks ? reinterpret_cast<float*>(peer_ptrs[r]) : nullptr;
  for (int r = 0; r < nranks; ++r)
    if (!p.p[r] || (reinterpret_cast<uintptr_t>(p.p[r]) & 15)) return 1;
  return 0;
}

}  // namespace

// chunk size used by both phases: ceil(n / nranks) rounded up to 4 elements
extern "C" int64_t nvit_xgmi_chunk(int64_t n, int nranks) {
  if (n <= 0 || nranks <= 0) return 0;
  const int64_t c = (n + nranks - 1) / nranks;
  return (c + 3) / 4 * 4;
}

// peer_ptrs: HOST array of nranks device pointers (this process's mappings of every rank's buffer; [rank] = own buffer).
extern "C" int nvit_xgmi_reduce_scatter(const int64_t* peer_ptrs, int nranks, int rank, int64_t n, float scale,
                                        void* stream) {
  NVIT_REQUIRE(peer_ptrs && nranks >= 1 && nranks <= XGMI_MAX_RANKS && rank >= 0 && rank < nranks && n > 0 && n % 4 == 0,
               "xgmi_reduce_scatter: bad arguments (1..8 ranks, n %% 4 == 0)");
  XgmiPeers p;
  NVIT_REQUIRE(fill_peers(peer_ptrs, nranks, p) == 0, "xgmi_reduce_scatter: peer pointers must be non-null and 16-byte aligned");
  const int64_t chunk = nvit_xgmi_chunk(n, nranks);
  const int64_t c0 = (int64_t)rank * chunk, c1 = c0 + chunk < n ? c0 + chunk : n;
  if (c1 <= c0) return NVIT_OK;
  hipStream_t s = (hipStream_t)stream;
  int blocks = cdiv((c1 - c0) / 4, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(xgmi_reduce_scatter_kernel, dim3(blocks), dim3(256), 0, s, p, nranks, rank, (long long)c0,
                     (long long)c1, scale);
  NVIT_CHECK_LAUNCH("xgmi_reduce_scatter");
  return NVIT_OK;
}

extern "C" int nvit_xgmi_all_gather(const int64_t* peer_ptrs, int nranks, int rank, int64_t n, void* stream) {
  NVIT_REQUIRE(peer_ptrs && nranks >= 1 && nranks <= XGMI_MAX_RANKS && rank >= 0 && rank < nranks && n > 0 && n % 4 == 0,
               "xgmi_all_gather: bad arguments (1..8 ranks, n %% 4 == 0)");
  XgmiPeers p;
  NVIT_REQUIRE(fill_peers(peer_ptrs, nranks, p) == 0, "xgmi_all_gather: peer pointers must be non-null and 16-byte aligned");
  if (nranks == 1) return NVIT_OK;
  const int64_t chunk = nvit_xgmi_chunk(n, nranks);
  hipStream_t s = (hipStream_t)stream;
  int blocks = cdiv(chunk / 4, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(xgmi_all_gather_kernel, dim3(blocks), dim3(256), 0, s, p, nranks, rank, (long long)chunk, (long long)n);
  NVIT_CHECK_LAUNCH("xgmi_all_gather");
  return NVIT_OK;
}
