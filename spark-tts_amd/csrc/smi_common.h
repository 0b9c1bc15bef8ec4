// Shared host/device helpers for libsparkmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include "sparkmi.h"
#ifdef SMI_DIAG
#include "sparkmi_debug.h"
#endif

// Environment switches exist only in the diagnostics build (libsparkmi_diag.so, -DSMI_DIAG: A/B variants, ablations, the
// test-only stage dumps).  The product library reads nothing from the environment: every behaviour a maintainer binding
// include/sparkmi.h sees comes from the config structs and the call arguments.
static inline const char* smi_env(const char* name) {
#ifdef SMI_DIAG
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

void smi_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define SMI_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      smi_set_error("%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);       \
      return SMI_EHIP;                                                                         \
    }                                                                                          \
  } while (0)

#define SMI_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      smi_set_error(__VA_ARGS__);         \
      return SMI_EINVAL;                  \
    }                                     \
  } while (0)

#define SMI_LAUNCH_CHECK()                                                                     \
  do {                                                                                         \
    hipError_t e_ = hipGetLastError();                                                         \
    if (e_ != hipSuccess) {                                                                    \
      smi_set_error("kernel launch: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__);   \
      return SMI_EHIP;                                                                         \
    }                                                                                          \
  } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static inline size_t smi_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#ifdef __HIPCC__
__device__ __forceinline__ float smi_bf16_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
// round-to-nearest-even, finite inputs (same bits as torch's .to(bfloat16))
__device__ __forceinline__ uint32_t smi_f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// Cross-lane sums on the DPP path (one VALU op per step, no LDS crossbar round trip).
template <int CTRL>
__device__ __forceinline__ float smi_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over aligned groups of 8 / 16 lanes; every lane of the group gets the result
__device__ __forceinline__ float smi_sum8(float v) {
  v += smi_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += smi_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += smi_dpp<0x141>(v);   // row_half_mirror
  return v;
}
__device__ __forceinline__ float smi_sum16(float v) {
  v = smi_sum8(v);
  v += smi_dpp<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float smi_readlane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// whole-wave sum (fixed association order => deterministic); the result is wave-uniform
__device__ __forceinline__ float smi_wave_sum(float v) {
  v = smi_sum16(v);
  return (smi_readlane(v, 0) + smi_readlane(v, 16)) + (smi_readlane(v, 32) + smi_readlane(v, 48));
}
// touch [ptr, ptr+bytes) so it is resident in the Infinity Cache for a later kernel
__device__ __forceinline__ void smi_prefetch_range(const void* ptr, size_t bytes, int worker, int nworkers) {
  const uint4* q = (const uint4*)ptr;
  const size_t n = bytes >> 4;
  uint32_t acc = 0;
  size_t i = (size_t)worker;
  for (; i + 7 * (size_t)nworkers < n; i += 8 * (size_t)nworkers) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = q[i + (size_t)u * nworkers];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u].x;
  }
  for (; i < n; i += nworkers) acc ^= q[i].x;
  asm volatile("" ::"v"(acc));
}
#endif
