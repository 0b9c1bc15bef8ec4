// Shared host/device helpers for libsparkmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "sparkmi.h"

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
__device__ __forceinline__ float smi_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#endif
