// Error plumbing, version and HIP-event timing of kernel families.
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void nvit_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* nvit_last_error(void) { return g_err; }
extern "C" int nvit_version(void) { return 100; }

// ---- event timing ----------------------------------------------------------------
// Events are recorded on the stream the kernel is launched on, immediately before and
// after the launch(es) of one C-ABI call; collect() synchronises them and sums per family.
namespace {
struct Rec {
  int kid;
  hipEvent_t a, b;
  double flops, bytes;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
volatile unsigned g_on = 0;   // bit k: launches of kernel family k are timed
thread_local Rec g_open[NVIT_KID_COUNT];
thread_local bool g_has_open[NVIT_KID_COUNT];

hipEvent_t get_event() {
  std::lock_guard<std::mutex> l(g_mu);
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void nvit_prof_begin(int kid, double flops, double bytes, hipStream_t s) {
  if (!((g_on >> kid) & 1u)) return;
  Rec r;
  r.kid = kid;
  r.a = get_event();
  r.b = get_event();
  r.flops = flops;
  r.bytes = bytes;
  (void)hipEventRecord(r.a, s);
  g_open[kid] = r;
  g_has_open[kid] = true;
}

void nvit_prof_end(int kid, hipStream_t s) {
  if (!g_on || !g_has_open[kid]) return;
  Rec r = g_open[kid];
  g_has_open[kid] = false;
  (void)hipEventRecord(r.b, s);
  std::lock_guard<std::mutex> l(g_mu);
  g_recs.push_back(r);
}

extern "C" void nvit_prof_enable(int on) { g_on = on ? ~0u : 0u; }
extern "C" void nvit_prof_select(unsigned mask) { g_on = mask; }

extern "C" int nvit_prof_collect(double* ms, double* flops, double* bytes, int64_t* launches) {
  std::vector<Rec> recs;
  {
    std::lock_guard<std::mutex> l(g_mu);
    recs.swap(g_recs);
  }
  for (int i = 0; i < NVIT_KID_COUNT; ++i) {
    ms[i] = 0;
    flops[i] = 0;
    bytes[i] = 0;
    launches[i] = 0;
  }
  for (auto& r : recs) {
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) NVIT_FAIL((int)e, "prof_collect: %s", hipGetErrorString(e));
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.a, r.b);
    ms[r.kid] += t;
    flops[r.kid] += r.flops;
    bytes[r.kid] += r.bytes;
    launches[r.kid] += 1;
  }
  std::lock_guard<std::mutex> l(g_mu);
  for (auto& r : recs) {
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  return NVIT_OK;
}

extern "C" const char* nvit_prof_name(int kid) {
  static const char* names[NVIT_KID_COUNT] = {"gemm_nt", "gemm_tn", "attn_fwd", "attn_bwd", "rowops",
                                              "renorm",  "shadow",  "patchify", "misc",     "gemm_f32", "gemm_swiglu",
                                              "gemm_qknorm", "gemm_swiglu_bwd", "optim"};
  return (kid >= 0 && kid < NVIT_KID_COUNT) ? names[kid] : "?";
}
