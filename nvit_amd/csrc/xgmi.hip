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
// nvit_amd/xgmi.py).
//
// Phase separation comes in two forms:
//   * nvit_xgmi_reduce_scatter / nvit_xgmi_all_gather: plain kernels, the CALLER separates the phases (stream
//     synchronisation + a host barrier of the process group);
//   * nvit_xgmi_reduce_scatter_sync / nvit_xgmi_all_gather_sync / nvit_xgmi_wait_gathered: the phases are separated ON
//     THE DEVICE by flags, so a call is just two kernel launches on a stream: no host round trip, several regions
//     ("slots": one per gradient bucket) in flight at once, overlappable with backward.  Every rank owns a small flag
//     block in UNCACHED device memory (hipDeviceMallocUncached: coherent across devices without cache maintenance),
//     exported over IPC like the data buffer; a rank announces a phase by STORING its epoch into the flag blocks of
//     all ranks (a remote store over the link) and waits by polling its OWN block (local loads):
//        ready[slot][r]    = e   rank r's region is completely written          (set when r's reduce-scatter starts:
//                                                                                   stream order puts it after the writers)
//        reduced[slot][r]  = e   chunk r of rank r's region holds the reduced sum (set by the last block of r's
//                                reduce-scatter, after a system-scope release of its stores); it also means r has
//                                finished READING chunk r of every other region
//        gathered[slot][r] = e   rank r has finished reading every other rank's reduced chunk
//     reduce-scatter waits for ready[*], all-gather for reduced[*], and nvit_xgmi_wait_gathered (issued before the
//     region is written again) for gathered[*].  Data is only ever WRITTEN locally and READ remotely; readers take a
//     system-scope acquire after their wait (stale remote lines in their caches), writers a system-scope release before
//     their flag.  Epochs count the calls per slot (same on every rank); every spin is bounded by wall time (the
//     100 MHz constant clock); a timeout is fatal for the whole job and fails CLOSED (see the flag-block layout below).
#include <stdlib.h>
#include <string.h>

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


// ------------------------------------------------------------------------------------------------------------------
// Device-synchronised form (see the header comment).  Flag block layout (32-bit words, one block per rank):
//   [0 .. 3*S*8)      phase p (0 ready, 1 reduced, 2 gathered), slot s, source rank r  ->  word (p*S + s)*8 + r
//   [3*S*8 .. +S)     per-slot arrival counter of this rank's own kernels (last-block detection), returns to 0
//   [3*S*8 + S]       error word: 0, or (code << 8 | slot + 1) of the first timed-out wait OF ANY RANK
//
// Failure is CLOSED, not open (round 4): a wait that times out (a dead, diverged or very late peer) stores the error word
// into the flag block of EVERY rank; every wait polls its own error word next to the flag it waits for and gives up as
// soon as it is set; a kernel whose wait failed moves no data AND publishes no phase flag - a rank that timed out never
// tells its peers "chunk reduced", so nobody gathers an unreduced chunk: all ranks end up with the error word set, every
// later kernel of the run returns at once, and the host sees the word through a pinned host copy that
// nvit_xgmi_wait_gathered refreshes at the end of every backward (DataParallel raises from it, on every rank).  The
// timeout is configurable (nvit_xgmi_set_timeout; default 1800 s = the 30-minute process-group timeout of the reference,
// train.py:224: rank 0 alone runs evaluation and checkpoints, train.py:878, so minutes of skew are legitimate).
namespace {

constexpr int XGMI_MAX_SLOTS = 64;
unsigned long long g_timeout_ticks = 1800ull * 100000000ull;   // of the 100 MHz constant clock

struct XgmiFlags {
  unsigned* f[XGMI_MAX_RANKS];   // this process's mappings of every rank's flag block ([rank] = own)
};

struct XgmiWaitList {            // passed by value (kernel argument): no device array, no host-to-device copy
  int n;
  int slot[XGMI_MAX_SLOTS];
  unsigned epoch[XGMI_MAX_SLOTS];
};

__device__ __forceinline__ unsigned* flag_word(unsigned* base, int S, int phase, int slot, int r) {
  return base + ((size_t)phase * S + slot) * XGMI_MAX_RANKS + r;
}
__device__ __forceinline__ unsigned* err_word(unsigned* base, int S) { return base + (size_t)3 * S * XGMI_MAX_RANKS + S; }

// thread 0 of the block: wait until flags[phase][slot][r] >= epoch for every r but skip_rank; false if any rank's wait
// has failed (own error word set) or this one times out - the failure is then broadcast to every rank's error word
__device__ __forceinline__ bool wait_flags(const XgmiFlags& fl, int rank, int S, int phase, int slot, int nranks,
                                           unsigned skip_rank, unsigned epoch, int code, unsigned long long timeout) {
  unsigned* own = fl.f[rank];
  unsigned* err = err_word(own, S);
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return false;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < nranks; ++r) {
    if ((unsigned)r == skip_rank) continue;
    unsigned* w = flag_word(own, S, phase, slot, r);
    // (>= in wrap-around arithmetic: a peer can never be a whole call ahead on a slot - it would have had to see this
    //  rank's `gathered` flag of the current call - but equality is not what the protocol means)
    while ((int)(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      __builtin_amdgcn_s_sleep(32);
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return false;   // a peer gave up
      if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) {
        const unsigned what = (unsigned)(code << 8 | (slot + 1));
        for (int j = 0; j < nranks; ++j) {   // first failure wins on every rank (remote CAS on uncached memory)
          unsigned expect = 0u;
          __hip_atomic_compare_exchange_strong(err_word(fl.f[j], S), &expect, what, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return false;
      }
    }
  }
  return true;
}

// all threads of the block: every wave drains its stores, the block's last arrival publishes flags[phase][slot][rank]
// = epoch into EVERY rank's block (own included) behind a system-scope release - unless a wait has failed anywhere
__device__ __forceinline__ void publish_when_last(const XgmiFlags& fl, int S, int phase, int slot, int nranks, int rank,
                                                  unsigned epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // system scope: write this XCD's dirty lines back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the fence's own wait may be dropped by the compiler)
    unsigned* cnt = fl.f[rank] + (size_t)3 * S * XGMI_MAX_RANKS + slot;
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // fail closed: some block of this kernel (or any rank) gave up -> this phase is never announced
      if (__hip_atomic_load(err_word(fl.f[rank], S), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u)
        for (int j = 0; j < nranks; ++j)
          __hip_atomic_store(flag_word(fl.f[j], S, phase, slot, rank), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ __launch_bounds__(256) void xgmi_rs_sync_kernel(XgmiPeers peers, XgmiFlags fl, int S, int slot, unsigned epoch,
                                                           int nranks, int rank, long long off, long long c0,
                                                           long long c1, float scale, unsigned long long timeout) {
  __shared__ int ok_s;
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)nranks)   // this rank's region is complete (stream order): tell everyone
    __hip_atomic_store(flag_word(fl.f[threadIdx.x], S, 0, slot, rank), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (threadIdx.x == 0) {
    ok_s = wait_flags(fl, rank, S, 0, slot, nranks, 0xffffffffu, epoch, 1, timeout) ? 1 : 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");        // drop stale copies of the peers' lines (system scope)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (ok_s) {
    const long long n4 = (c1 - c0) >> 2;
    float* out = peers.p[rank] + off + c0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
      f32x4 v[XGMI_MAX_RANKS];
#pragma unroll
      for (int r = 0; r < XGMI_MAX_RANKS; ++r)
        if (r < nranks) v[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(peers.p[r] + off + c0) + i);
      f32x4 acc = v[0];
#pragma unroll
      for (int r = 1; r < XGMI_MAX_RANKS; ++r)
        if (r < nranks) acc += v[r];
      reinterpret_cast<f32x4*>(out)[i] = acc * scale;
    }
  }
  publish_when_last(fl, S, 1, slot, nranks, rank, epoch);   // (announces nothing once any wait has failed)
}

__global__ __launch_bounds__(256) void xgmi_ag_sync_kernel(XgmiPeers peers, XgmiFlags fl, int S, int slot, unsigned epoch,
                                                           int nranks, int rank, long long off, long long chunk,
                                                           long long n, unsigned long long timeout) {
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    ok_s = wait_flags(fl, rank, S, 1, slot, nranks, (unsigned)rank, epoch, 2, timeout) ? 1 : 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (ok_s) {
    float* mine = peers.p[rank] + off;
    for (int j = 0; j < nranks; ++j) {
      if (j == rank) continue;
      const long long c0 = (long long)j * chunk, c1 = c0 + chunk < n ? c0 + chunk : n;
      if (c1 <= c0) continue;
      const long long n4 = (c1 - c0) >> 2;
      const f32x4* src = reinterpret_cast<const f32x4*>(peers.p[j] + off + c0);
      f32x4* dst = reinterpret_cast<f32x4*>(mine + c0);
      for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        dst[i] = __builtin_nontemporal_load(src + i);
    }
  }
  publish_when_last(fl, S, 2, slot, nranks, rank, epoch);
}

// one block: every listed slot's region has been read by every peer (their all-gathers are done); then the error word is
// copied to the caller's pinned host word, so the host can look at it after every step without touching the device
__global__ __launch_bounds__(64) void xgmi_wait_gathered_kernel(XgmiFlags fl, int S, int rank, int nranks, XgmiWaitList wl,
                                                                unsigned* host_err, unsigned long long timeout) {
  if (threadIdx.x == 0) {
    for (int k = 0; k < wl.n; ++k)
      if (!wait_flags(fl, rank, S, 2, wl.slot[k], nranks, (unsigned)rank, wl.epoch[k], 3, timeout)) break;
    if (host_err) {
      const unsigned e = __hip_atomic_load(err_word(fl.f[rank], S), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(host_err, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

int fill_flags(const int64_t* flag_ptrs, int nranks, XgmiFlags& f) {
  for (int r = 0; r < XGMI_MAX_RANKS; ++r) f.f[r] = r < nranks ? reinterpret_cast<unsigned*>(flag_ptrs[r]) : nullptr;
  for (int r = 0; r < nranks; ++r)
    if (!f.f[r] || (reinterpret_cast<uintptr_t>(f.f[r]) & 3)) return 1;
  return 0;
}

int xgmi_blocks(long long n4) {
  static int cap = 0;
  if (cap == 0) {
    const char* e = getenv("NVIT_XGMI_BLOCKS");   // blocks per collective kernel: few enough to leave the CUs to backward
    cap = e ? atoi(e) : 32;
    if (cap < 1) cap = 1;
    if (cap > 2048) cap = 2048;
  }
  int b = cdiv(n4, 256);
  return b < 1 ? 1 : (b > cap ? cap : b);
}

}  // namespace

extern "C" int64_t nvit_xgmi_flag_bytes(int nslots) {
  if (nslots < 1 || nslots > XGMI_MAX_SLOTS) return 0;
  const int64_t words = (int64_t)3 * nslots * XGMI_MAX_RANKS + nslots + 1;
  return (words * 4 + 255) / 256 * 256;
}

// Bound of every device-side wait, in seconds (0.001 .. 86400); the same value must be set on every rank.
extern "C" int nvit_xgmi_set_timeout(double seconds) {
  NVIT_REQUIRE(seconds >= 1e-3 && seconds <= 86400.0, "xgmi_set_timeout: 0.001 .. 86400 seconds");
  g_timeout_ticks = (unsigned long long)(seconds * 1e8);
  return NVIT_OK;
}

// One pinned, device-visible host word (zeroed) that nvit_xgmi_wait_gathered refreshes with the error word; *host_ptr is
// what the host reads (no synchronisation needed), *dev_ptr what the kernel is given.
extern "C" int nvit_xgmi_errword_alloc(void** host_ptr, void** dev_ptr) {
  NVIT_REQUIRE(host_ptr && dev_ptr, "xgmi_errword_alloc: null argument");
  void* h = nullptr;
  hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) NVIT_FAIL((int)e, "xgmi_errword_alloc: hipHostMalloc: %s", hipGetErrorString(e));
  memset(h, 0, 64);
  void* d = nullptr;
  e = hipHostGetDevicePointer(&d, h, 0);
  if (e != hipSuccess) {
    (void)hipHostFree(h);
    NVIT_FAIL((int)e, "xgmi_errword_alloc: hipHostGetDevicePointer: %s", hipGetErrorString(e));
  }
  *host_ptr = h;
  *dev_ptr = d;
  return NVIT_OK;
}

extern "C" int nvit_xgmi_errword_free(void* host_ptr) {
  hipError_t e = hipHostFree(host_ptr);
  if (e != hipSuccess) NVIT_FAIL((int)e, "xgmi_errword_free: %s", hipGetErrorString(e));
  return NVIT_OK;
}

// Uncached (cross-device coherent) device memory for the flag block, zero-filled, with its IPC handle (64 bytes).
extern "C" int nvit_xgmi_flags_alloc(int nslots, void** dev_ptr, void* ipc_handle_out) {
  const int64_t bytes = nvit_xgmi_flag_bytes(nslots);
  NVIT_REQUIRE(bytes > 0 && dev_ptr && ipc_handle_out, "xgmi_flags_alloc: 1..%d slots", XGMI_MAX_SLOTS);
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) NVIT_FAIL((int)e, "xgmi_flags_alloc: hipExtMallocWithFlags(uncached): %s", hipGetErrorString(e));
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(ipc_handle_out), p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    NVIT_FAIL((int)e, "xgmi_flags_alloc: %s", hipGetErrorString(e));
  }
  *dev_ptr = p;
  return NVIT_OK;
}

extern "C" int nvit_xgmi_flags_open(const void* ipc_handle, void** dev_ptr) {
  NVIT_REQUIRE(ipc_handle && dev_ptr, "xgmi_flags_open: null argument");
  hipIpcMemHandle_t h;
  memcpy(&h, ipc_handle, sizeof(h));
  hipError_t e = hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) NVIT_FAIL((int)e, "xgmi_flags_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
  return NVIT_OK;
}

extern "C" int nvit_xgmi_flags_close(void* dev_ptr) {
  hipError_t e = hipIpcCloseMemHandle(dev_ptr);
  if (e != hipSuccess) NVIT_FAIL((int)e, "xgmi_flags_close: %s", hipGetErrorString(e));
  return NVIT_OK;
}

extern "C" int nvit_xgmi_flags_free(void* dev_ptr) {
  hipError_t e = hipFree(dev_ptr);
  if (e != hipSuccess) NVIT_FAIL((int)e, "xgmi_flags_free: %s", hipGetErrorString(e));
  return NVIT_OK;
}

// error word of this rank's own flag block (0 = no wait has timed out on any rank); synchronises the stream first
extern "C" int nvit_xgmi_flags_error(const void* own_flags, int nslots, unsigned* out, void* stream) {
  NVIT_REQUIRE(own_flags && out && nvit_xgmi_flag_bytes(nslots) > 0, "xgmi_flags_error: bad arguments");
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e == hipSuccess)
    e = hipMemcpy(out, (const unsigned*)own_flags + (size_t)3 * nslots * XGMI_MAX_RANKS + nslots, 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) NVIT_FAIL((int)e, "xgmi_flags_error: %s", hipGetErrorString(e));
  return NVIT_OK;
}

// Region [off, off + n) of the symmetric buffers (elements; off and n multiples of 4), slot `slot`, call number `epoch`
// (>= 1, the same on every rank, increasing per slot).  chunk = nvit_xgmi_chunk(n, nranks).
extern "C" int nvit_xgmi_reduce_scatter_sync(const int64_t* peer_ptrs, const int64_t* flag_ptrs, int nranks, int rank,
                                             int nslots, int slot, unsigned epoch, int64_t off, int64_t n, float scale,
                                             void* stream) {
  NVIT_REQUIRE(peer_ptrs && flag_ptrs && nranks >= 1 && nranks <= XGMI_MAX_RANKS && rank >= 0 && rank < nranks && n > 0 &&
                   n % 4 == 0 && off >= 0 && off % 4 == 0 && nslots >= 1 && nslots <= XGMI_MAX_SLOTS && slot >= 0 &&
                   slot < nslots && epoch != 0,
               "xgmi_reduce_scatter_sync: bad arguments");
  XgmiPeers p;
  XgmiFlags f;
  NVIT_REQUIRE(fill_peers(peer_ptrs, nranks, p) == 0 && fill_flags(flag_ptrs, nranks, f) == 0,
               "xgmi_reduce_scatter_sync: peer / flag pointers must be non-null and aligned");
  const int64_t chunk = nvit_xgmi_chunk(n, nranks);
  int64_t c0 = (int64_t)rank * chunk, c1 = c0 + chunk < n ? c0 + chunk : n;
  if (c1 < c0) c1 = c0;   // (a rank without a chunk still takes part in the flag protocol)
  hipLaunchKernelGGL(xgmi_rs_sync_kernel, dim3(xgmi_blocks((c1 - c0) / 4)), dim3(256), 0, (hipStream_t)stream, p, f, nslots,
                     slot, epoch, nranks, rank, (long long)off, (long long)c0, (long long)c1, scale, g_timeout_ticks);
  NVIT_CHECK_LAUNCH("xgmi_reduce_scatter_sync");
  return NVIT_OK;
}

extern "C" int nvit_xgmi_all_gather_sync(const int64_t* peer_ptrs, const int64_t* flag_ptrs, int nranks, int rank, int nslots,
                                         int slot, unsigned epoch, int64_t off, int64_t n, void* stream) {
  NVIT_REQUIRE(peer_ptrs && flag_ptrs && nranks >= 1 && nranks <= XGMI_MAX_RANKS && rank >= 0 && rank < nranks && n > 0 &&
                   n % 4 == 0 && off >= 0 && off % 4 == 0 && nslots >= 1 && nslots <= XGMI_MAX_SLOTS && slot >= 0 &&
                   slot < nslots && epoch != 0,
               "xgmi_all_gather_sync: bad arguments");
  XgmiPeers p;
  XgmiFlags f;
  NVIT_REQUIRE(fill_peers(peer_ptrs, nranks, p) == 0 && fill_flags(flag_ptrs, nranks, f) == 0,
               "xgmi_all_gather_sync: peer / flag pointers must be non-null and aligned");
  const int64_t chunk = nvit_xgmi_chunk(n, nranks);
  hipLaunchKernelGGL(xgmi_ag_sync_kernel, dim3(xgmi_blocks(chunk / 4)), dim3(256), 0, (hipStream_t)stream, p, f, nslots, slot,
                     epoch, nranks, rank, (long long)off, (long long)chunk, (long long)n, g_timeout_ticks);
  NVIT_CHECK_LAUNCH("xgmi_all_gather_sync");
  return NVIT_OK;
}

// slots / epochs: HOST arrays of nwait entries (which slots, and the epoch each must have reached); they travel to the
// kernel by value.  host_err_dev: device pointer of a pinned host word (nvit_xgmi_errword_alloc) that receives the error
// word when the wait is over, or NULL.
extern "C" int nvit_xgmi_wait_gathered(const int64_t* flag_ptrs, int nranks, int rank, int nslots, const int* slots,
                                       const unsigned* epochs, int nwait, void* host_err_dev, void* stream) {
  NVIT_REQUIRE(flag_ptrs && nranks >= 1 && nranks <= XGMI_MAX_RANKS && rank >= 0 && rank < nranks && nslots >= 1 &&
                   nslots <= XGMI_MAX_SLOTS && slots && epochs && nwait >= 1 && nwait <= XGMI_MAX_SLOTS,
               "xgmi_wait_gathered: bad arguments");
  XgmiFlags f;
  NVIT_REQUIRE(fill_flags(flag_ptrs, nranks, f) == 0, "xgmi_wait_gathered: flag pointers must be non-null and aligned");
  XgmiWaitList wl;
  wl.n = nwait;
  for (int k = 0; k < XGMI_MAX_SLOTS; ++k) {
    wl.slot[k] = k < nwait ? slots[k] : 0;
    wl.epoch[k] = k < nwait ? epochs[k] : 0u;
    NVIT_REQUIRE(wl.slot[k] >= 0 && wl.slot[k] < nslots, "xgmi_wait_gathered: slot out of range");
  }
  hipLaunchKernelGGL(xgmi_wait_gathered_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, f, nslots, rank, nranks, wl,
                     reinterpret_cast<unsigned*>(host_err_dev), g_timeout_ticks);
  NVIT_CHECK_LAUNCH("xgmi_wait_gathered");
  return NVIT_OK;
}
