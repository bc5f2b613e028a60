// Shared device/host helpers for the nViT gfx950 kernels.  CDNA4 only: 64-wide
// wavefronts are assumed everywhere (no dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nvit_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define WAVE 64

// Timing probes (cut-down builds whose results are garbage by design) never live in these sources: a probe is its own
// translation unit under tools/probes/.  tests/test_cabi_symbols.py asserts that no NVIT_PROBE switch exists here.

// ---- host-side error plumbing (no exceptions cross the C boundary) -------------
void nvit_set_error(const char* fmt, ...);
#define NVIT_FAIL(code, ...)                \
  do {                                      \
    nvit_set_error(__VA_ARGS__);            \
    return (code);                          \
  } while (0)
#define NVIT_REQUIRE(cond, ...)                          \
  do {                                                   \
    if (!(cond)) NVIT_FAIL(NVIT_EINVAL, __VA_ARGS__);    \
  } while (0)
#define NVIT_CHECK_LAUNCH(name)                                                     \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) NVIT_FAIL((int)e__, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---- per-kernel HIP-event timing (bench.py roofline leg) -----------------------
void nvit_prof_begin(int kid, double flops, double bytes, hipStream_t s);
void nvit_prof_end(int kid, hipStream_t s);
struct ProfScope {
  int kid;
  hipStream_t s;
  ProfScope(int k, double flops, double bytes, hipStream_t st) : kid(k), s(st) { nvit_prof_begin(k, flops, bytes, st); }
  ~ProfScope() { nvit_prof_end(kid, s); }
};

// ---- device helpers -------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over aligned groups of G consecutive lanes (G power of two <= 64)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float group_sum_dyn(float v, int G) {
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T>
__device__ __forceinline__ f32x4 load4(const T* p);
template <>
__device__ __forceinline__ f32x4 load4<float>(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}
template <>
__device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
template <typename T>
__device__ __forceinline__ void store4(T* p, f32x4 v);
template <>
__device__ __forceinline__ void store4<float>(float* p, f32x4 v) {
  *reinterpret_cast<f32x4*>(p) = v;
}
template <>
__device__ __forceinline__ void store4<bf16>(bf16* p, f32x4 v) {
  bf16x4 r = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *reinterpret_cast<bf16x4*>(p) = r;
}
template <typename T>
__device__ __forceinline__ float ld1(const T* p) {
  return (float)(*p);
}
template <typename T>
__device__ __forceinline__ void st1(T* p, float v) {
  *p = (T)v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
