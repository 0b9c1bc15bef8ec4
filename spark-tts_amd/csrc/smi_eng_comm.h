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

// Words in LDS that several waves of a workgroup poll or set: always through an LDS-typed pointer and a relaxed workgroup-scope
// atomic.  (A `volatile` access through a generic pointer compiles to flat_load / flat_store ... sc0 sc1 followed by
// s_waitcnt vmcnt(0): in a wave with LDS-DMA fills or sweep loads in flight every look at such a word would wait for ALL of
// them -- this cost the engine 10 us per layer before it was found in the ISA.)
typedef __attribute__((address_space(3))) unsigned eng_lds_u32;
__device__ __forceinline__ eng_lds_u32* eng_lds_ptr(const unsigned* p) { return (eng_lds_u32*)(__attribute__((address_space(3))) void*)(void*)p; }
__device__ __forceinline__ unsigned eng_lds_load(const unsigned* p) { return __hip_atomic_load(eng_lds_ptr(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void eng_lds_store(unsigned* p, unsigned v) { __hip_atomic_store(eng_lds_ptr(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct EngSync {
  unsigned* err;            // [4] device words: 0 = timeout code (0: none), 1 = where, 2 = spare, 3 = spare
  unsigned long long t_end; // s_memrealtime deadline of this launch (set per wave at kernel entry)
  int quiet;                // 1: watch one granule first (two-stage sweep); 0: sweep the whole set from the start;
                            // 2: sleep `predelay` x 64 cycles (the phase's expected length), then sweep with pauses
  int predelay;
  unsigned* go;             // LDS word of the workgroup (eng_lds_load / eng_lds_store): the epoch whose first granule the watching wave has seen
  bool watcher;             // this wave watches global memory for the workgroup (the others wait on `go`)
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

// The same sweep over arbitrary granule indices (idx[k] < 0: none), in two stages.  While the producers are still computing,
// hundreds of waves re-reading whole vectors with L1-bypassing loads is traffic on the very lines the producers are about to
// write and beside the weight stream (price-list row polling-cost), so a wave first watches ONE granule of its set -- one lane,
// one 8-byte load, a short sleep between looks -- and sweeps its whole set only once that one shows the epoch; the producers of
// an edge finish within a microsecond of each other, so the full sweep then rarely needs a second pass.
template <int NPT>
__device__ __forceinline__ bool eng_sweep_idx(const smi_u64* g, const int (&idx)[NPT], unsigned tag, unsigned (&v)[NPT],
                                              const EngSync& sy, unsigned where) {
  unsigned pend = 0;
#pragma unroll
  for (int k = 0; k < NPT; ++k) { pend |= idx[k] >= 0 ? (1u << k) : 0u; v[k] = 0u; }
  auto give_up = [&]() -> bool {   // the deadline and the other workgroups' give-up word
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (eng_err_load(sy.err) != 0) return true;
    if (now > sy.t_end) {
      if ((threadIdx.x & 63) == 0) { atomicCAS(sy.err, 0u, 1u); atomicCAS(sy.err + 1, 0u, where); }
      return true;
    }
    return false;
  };
  if (sy.quiet == 2) {
    for (int z = 0; z < sy.predelay; ++z) __builtin_amdgcn_s_sleep(1);
  } else if (sy.quiet) {   // stage 1: ONE wave of the workgroup watches the first granule of its set (the lane that owns it); the others wait in LDS
    const unsigned long long have = __ballot(pend != 0);
    if (have == 0) return true;
    if (sy.watcher) {
      const int first = __ffsll((long long)have) - 1;
      const bool me = (int)(threadIdx.x & 63) == first;
      int k0 = 0;
#pragma unroll
      for (int k = NPT - 1; k >= 0; --k) k0 = (pend & (1u << k)) ? k : k0;
      for (unsigned it = 0;; ++it) {
        bool seen = false;
        if (me) seen = (unsigned)(eng_gload(g + idx[k0]) >> 32) == tag;
        if (__any(seen)) break;
        __builtin_amdgcn_s_sleep(2);
        if ((it & 63u) == 63u && give_up()) { if ((threadIdx.x & 63) == 0) eng_lds_store(sy.go, 0xffffffffu); return false; }
      }
      if ((threadIdx.x & 63) == 0) eng_lds_store(sy.go, tag);
    } else {
      for (unsigned it = 0;; ++it) {
        const unsigned gv = eng_lds_load(sy.go);
        if (gv == tag) break;
        if (gv == 0xffffffffu) return false;
        __builtin_amdgcn_s_sleep(1);
        if ((it & 1023u) == 1023u && give_up()) return false;
      }
    }
  }
  for (unsigned it = 0;; ++it) {   // stage 2: the whole set; only granules still missing are re-read
    smi_u64 x[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k)
      if (pend & (1u << k)) x[k] = eng_gload(g + idx[k]);
#pragma unroll
    for (int k = 0; k < NPT; ++k)
      if ((pend & (1u << k)) && (unsigned)(x[k] >> 32) == tag) { v[k] = (unsigned)x[k]; pend &= ~(1u << k); }
    if (!__any(pend != 0)) return true;
    if (sy.quiet == 2) __builtin_amdgcn_s_sleep(3);
    else if (sy.quiet) __builtin_amdgcn_s_sleep(1);
    if ((it & 31u) == 31u && give_up()) return false;
  }
}
