// diag_flow.hip -- dataflow-chain microbenchmark (diagnostics only; not part of the product path).
//
// Prices the hand-off a one-launch decode step would be made of: ONE kernel whose blocks are the
// blocks of a chain of dependent phases (per layer: phase sizes like QKV / attention / o_proj /
// gate_up / down).  A block first requests its "weights" (a private, HBM-cold slice of a big buffer),
// then waits until every block of the previous phase has arrived on that phase's counters, reads the
// previous phase's output vector, writes its slice of its own output and arrives.  Blocks are
// dispatched in index order, so the blocks of later phases run ahead, hold their weights in
// registers and spin: the chain's critical path is the hand-off alone.
//
// Hand-off protocol (MI355X_MICROARCH.md, "Valid forms", first table row): payload stores and loads
// are agent-scope (sc1), every storing wave drains vmcnt, workgroup barrier, one lane adds to an
// agent-scope counter (8 shards, one 128-B line each); the consumer polls the shards with sc1 loads
// from one lane group, then a workgroup barrier, then sc1 loads of the payload.
// Every spin is bounded: a block that times out raises `err` and every other poller then leaves too.
#include "../smi_common.h"
#include <string.h>
#include <vector>

namespace {

constexpr int kShards = 8, kShardStride = 32;   // counters: [phase][8 shards][32 uints] (one 128-B line per shard)
constexpr int kVec = 1024;                      // floats in a phase's output vector
constexpr int kMaxPat = 8;

struct FlowArg {
  const uint4* big; size_t big_vecs;   // weight stand-in
  unsigned* ctr;                       // [nphases + 1][kShards][kShardStride]
  float* out;                          // [nbuf][kVec]
  unsigned* epoch;                     // [1] completed launches
  unsigned* err;                       // [4]: 0 timeout flag, 1 stale-payload count, 2 max poll iterations seen
  int npat, per_layer, nlayers, reuse; // pattern of phase sizes repeated per layer; reuse: two payload buffers ping-pong
  int size[kMaxPat], start[kMaxPat], loads[kMaxPat];
  unsigned spin_limit;
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one lane group of wave 0 polls the 8 shards (+ the abort word); returns with the block released
__device__ __forceinline__ void flow_wait(const unsigned* ctr, unsigned target, unsigned* err, unsigned limit) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    unsigned it = 0;
    for (;;) {
      unsigned v = 0;
      if (lane < kShards) v = ld_sc1(ctr + lane * kShardStride);
      else if (lane == kShards) v = ld_sc1(err) ? 0x40000000u : 0u;   // somebody timed out: everyone leaves
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      v = __builtin_amdgcn_readfirstlane(v);
      if ((v & 0x3fffffffu) >= target || (v & 0x40000000u)) break;
      if (++it > limit) { if (lane == 0) atomicExch(err, 1u); break; }
      __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) atomicMax(err + 2, it);
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void ub_flow(FlowArg a) {
  const int tid = threadIdx.x;
  const int total = a.per_layer * a.nlayers;
  int phase, bip, psize, loads;   // global phase index, block in phase
  if ((int)blockIdx.x >= total) {   // the closing block: bumps the epoch after everything has arrived
    phase = a.npat * a.nlayers; bip = 0; psize = 1; loads = 0;
  } else {
    const int layer = (int)blockIdx.x / a.per_layer, r = (int)blockIdx.x % a.per_layer;
    int k = 0;
#pragma unroll
    for (int i = 1; i < kMaxPat; ++i) if (i < a.npat && r >= a.start[i]) k = i;
    phase = layer * a.npat + k; bip = r - a.start[k]; psize = a.size[k]; loads = a.loads[k];
  }
  const unsigned ep = *a.epoch;
  // ---- "weights": issued before the wait, consumed after it
  uint32_t acc = 0;
  {
    const size_t base = ((size_t)blockIdx.x * 4096 + (size_t)ep * 1237 * 4096) % (a.big_vecs - (size_t)8 * 1024);
    const uint4* p = a.big + base + tid;
    uint4 w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = i < loads ? p[(size_t)i * blockDim.x] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(w[i].x), "+v"(w[i].y));
    // ---- wait for the previous phase
    float in = 0.f;
    if (phase > 0) {
      const int pk = (phase - 1) % a.npat;
      const unsigned prev_size = (unsigned)a.size[pk];
      flow_wait(a.ctr + (size_t)(phase - 1) * kShards * kShardStride, (ep + 1) * prev_size, a.err, a.spin_limit);
      const float* src = a.out + (size_t)(a.reuse ? ((phase - 1) & 1) : (phase - 1)) * kVec;
      in = ld_sc1(src + (tid & (kVec - 1)));
      const float want = (float)(ep * 1000u + (unsigned)(phase - 1));
      if (in != want) atomicAdd(a.err + 1, 1u);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
  }
  if ((int)blockIdx.x >= total) {
    if (tid == 0) { *a.epoch = ep + 1; if (acc == 0x9e3779b9u) a.err[3] = 1; }
    return;
  }
  // ---- this block's slice of the phase's output vector
  {
    float* dst = a.out + (size_t)(a.reuse ? (phase & 1) : phase) * kVec;
    const int lo = (int)((long)bip * kVec / psize), hi = (int)((long)(bip + 1) * kVec / psize);
    const float val = (float)(ep * 1000u + (unsigned)phase) + (acc == 0x9e3779b9u ? 1.f : 0.f);
    for (int i = lo + tid; i < hi; i += blockDim.x) st_sc1(dst + i, val);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0)
    __hip_atomic_fetch_add(a.ctr + ((size_t)phase * kShards + (blockIdx.x & (kShards - 1))) * kShardStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

// sizes/loads: the per-layer pattern (npat phases); the launch has nlayers * sum(sizes) + 1 blocks of `block` threads.
// out: us per launch (average over `launches` back-to-back launches after one warm-up), err[4] as in FlowArg.
extern "C" int smi_diag_flow(const int* sizes, const int* loads, int npat, int nlayers, int block, int reuse, int launches,
                             const void* big, size_t big_bytes, unsigned spin_limit, float* us_per_launch, unsigned* err_out,
                             void* stream) {
  if (npat < 1 || npat > kMaxPat || nlayers < 1 || block < 64 || block > 1024 || block % 64) { smi_set_error("diag_flow: bad shape"); return SMI_EINVAL; }
  FlowArg a;
  memset(&a, 0, sizeof(a));
  a.npat = npat; a.nlayers = nlayers; a.reuse = reuse; a.spin_limit = spin_limit;
  int s = 0;
  for (int i = 0; i < npat; ++i) {
    if (sizes[i] < 1 || loads[i] < 0 || loads[i] > 8) { smi_set_error("diag_flow: bad phase %d", i); return SMI_EINVAL; }
    a.size[i] = sizes[i]; a.loads[i] = loads[i]; a.start[i] = s; s += sizes[i];
  }
  a.per_layer = s;
  const int nphases = npat * nlayers;
  a.big = (const uint4*)big; a.big_vecs = big_bytes / 16;
  if (a.big_vecs < (size_t)64 * 1024) { smi_set_error("diag_flow: buffer too small"); return SMI_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  const size_t cbytes = (size_t)(nphases + 1) * kShards * kShardStride * 4, obytes = (size_t)(nphases + 2) * kVec * 4;
  unsigned char* mem = nullptr;
  SMI_HIP(hipMalloc((void**)&mem, cbytes + obytes + 256));
  SMI_HIP(hipMemsetAsync(mem, 0, cbytes + obytes + 256, st));
  a.ctr = (unsigned*)mem; a.out = (float*)(mem + cbytes); a.epoch = (unsigned*)(mem + cbytes + obytes); a.err = a.epoch + 16;
  hipEvent_t e0, e1;
  SMI_HIP(hipEventCreate(&e0)); SMI_HIP(hipEventCreate(&e1));
  const dim3 grid((unsigned)(a.per_layer * nlayers + 1));
  hipLaunchKernelGGL(ub_flow, grid, dim3(block), 0, st, a);
  SMI_LAUNCH_CHECK();
  SMI_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(ub_flow, grid, dim3(block), 0, st, a);
  SMI_LAUNCH_CHECK();
  SMI_HIP(hipEventRecord(e1, st));
  SMI_HIP(hipEventSynchronize(e1));
  float ms = 0;
  SMI_HIP(hipEventElapsedTime(&ms, e0, e1));
  *us_per_launch = ms * 1e3f / (float)launches;
  SMI_HIP(hipMemcpy(err_out, a.err, 16, hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(mem);
  return SMI_OK;
}
