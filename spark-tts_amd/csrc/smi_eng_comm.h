// smi_eng_comm.h -- hand-offs between the workgroups of ONE launch (the one-row decode engine, smi_eng.hip).
//
// A vector produced by many CUs and consumed by many CUs travels as 8-byte granules {tag : value}: ONE naturally aligned
// agent-scope (sc1, write-through) store per element, the tag in the upper half.  The data is its own flag: a consumer
// re-reads its granules with agent-scope (sc1, L1-bypassing) loads until every tag equals the epoch it waits for; no
// counter, no fence, no second word (MI355X_MICROARCH.md, visibility section, form R2; price-list rows handoff-1to1 / allgather).
// Tags are 32-bit epochs that never repeat between two consecutive uses of a granule slot, so nothing is zeroed between
// launches.  Every spin is bounded by the 100 MHz wall clock: a wave that gives up raises the launch's error word, which
// every poller also looks at, so a lost hand-off ends the launch instead of hanging the GPU.
#pragma once
#include "smi_common.h"

typedef unsigned long long smi_u64;

struct EngSync {
  unsigned* err;            // [4] device words: 0 = timeout code (0: none), 1 = where, 2 = spare, 3 = spare
  unsigned long long t_end; // s_memrealtime deadline of this launch (set per wave at kernel entry)
};

__device__ __forceinline__ smi_u64 eng_gload(const smi_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void eng_gstore(smi_u64* p, unsigned tag, unsigned value) {
  __hip_atomic_store(p, ((smi_u64)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned eng_err_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Sweep: this thread owns granules g[first + k * stride], k < NPT, those with index < n.  Returns once all of the WAVE's
// granules carry `tag` (values in v[]), or false after the deadline / when another workgroup has given up.
// Only granules still missing are re-read.
template <int NPT>
__device__ __forceinline__ bool eng_sweep(const smi_u64* g, int first, int stride, int n, unsigned tag, unsigned (&v)[NPT],
                                          const EngSync& sy, unsigned where) {
  unsigned pend = 0;
#pragma unroll
  for (int k = 0; k < NPT; ++k) pend |= (first + k * stride < n) ? (1u << k) : 0u;
  for (unsigned it = 0;; ++it) {
    smi_u64 x[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k)
      if (pend & (1u << k)) x[k] = eng_gload(g + first + k * stride);
#pragma unroll
    for (int k = 0; k < NPT; ++k)
      if ((pend & (1u << k)) && (unsigned)(x[k] >> 32) == tag) { v[k] = (unsigned)x[k]; pend &= ~(1u << k); }
    if (!__any(pend != 0)) return true;
    if ((it & 31u) == 31u) {   // bounded spin: the deadline and the other workgroups' give-up word
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      const unsigned e = eng_err_load(sy.err);
      if (e != 0) return false;
      if (now > sy.t_end) {
        if ((threadIdx.x & 63) == 0) { atomicCAS(sy.err, 0u, 1u); atomicCAS(sy.err + 1, 0u, where); }
        return false;
      }
    }
  }
}

// The same sweep over arbitrary granule indices (idx[k] < 0: none).
template <int NPT>
__device__ __forceinline__ bool eng_sweep_idx(const smi_u64* g, const int (&idx)[NPT], unsigned tag, unsigned (&v)[NPT],
                                              const EngSync& sy, unsigned where) {
  unsigned pend = 0;
#pragma unroll
  for (int k = 0; k < NPT; ++k) { pend |= idx[k] >= 0 ? (1u << k) : 0u; v[k] = 0u; }
  for (unsigned it = 0;; ++it) {
    smi_u64 x[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k)
      if (pend & (1u << k)) x[k] = eng_gload(g + idx[k]);
#pragma unroll
    for (int k = 0; k < NPT; ++k)
      if ((pend & (1u << k)) && (unsigned)(x[k] >> 32) == tag) { v[k] = (unsigned)x[k]; pend &= ~(1u << k); }
    if (!__any(pend != 0)) return true;
    if ((it & 31u) == 31u) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      const unsigned e = eng_err_load(sy.err);
      if (e != 0) return false;
      if (now > sy.t_end) {
        if ((threadIdx.x & 63) == 0) { atomicCAS(sy.err, 0u, 1u); atomicCAS(sy.err + 1, 0u, where); }
        return false;
      }
    }
  }
}
