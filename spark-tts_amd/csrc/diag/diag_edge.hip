// diag_edge.hip -- hand-off skeleton of the one-row decode engine (diagnostics only; not part of the product path).
//
// ONE persistent launch, one 512-thread workgroup per CU, runs `layers` decode layers that consist of NOTHING but the five
// all-to-all edges of a Qwen2.5-0.5B layer at one row, with the engine's producer / consumer sets and granule counts:
//   A  h        896 values   224 CUs -> 256 CUs         (down_proj -> next QKV)
//   B  q|k|v   1152 values   256 CUs -> 14 head CUs     (QKV -> attention; a head reads its 64 + 64 + 64)
//   C  attn     896 values    14 CUs -> 224 CUs         (attention -> o_proj)
//   D  h_mid    896 values   224 CUs -> 256 CUs         (o_proj -> gate_up)
//   E  act     4864 values   256 CUs -> 224 CUs         (gate_up -> down_proj)
// Every value carries what it should be, so a consumer can tell a stale granule from a fresh one; optional background
// LDS-DMA stream per wave (the weight stream stand-in).  What it prices: the critical path of smi_eng.hip without its compute.
#include "../smi_eng_comm.h"
#include <string.h>
#include <vector>

namespace {

constexpr int kH = 896, kQKV = 1152, kI = 4864, kHeads = 14, kDownCU = 224, kThreads = 512;

struct EdgeArg {
  smi_u64* gran;          // [2][kH + kQKV + kH + kH + kI] granules, double-buffered by layer parity
  unsigned* serial;       // [1] launches so far (tag base)
  unsigned* err;          // [4]
  unsigned* bad;          // [1] granules whose value was not the expected one
  unsigned long long* stamps;   // [2][layers][8] s_memrealtime of CU 0 and of head CU 0 (or null)
  const unsigned char* big; size_t big_bytes;   // background stream source
  int layers, dma_kib, ncu;
  unsigned timeout_ticks;
};

constexpr int kOffA = 0, kOffB = kH, kOffC = kH + kQKV, kOffD = 2 * kH + kQKV, kOffE = 3 * kH + kQKV, kGranPerBuf = 3 * kH + kQKV + kI;

__device__ __forceinline__ unsigned want_val(int layer, int edge, int i) { return ((unsigned)layer << 20) ^ ((unsigned)edge << 16) ^ (unsigned)i; }

__device__ __forceinline__ void dma1k(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NPT>
__device__ __forceinline__ bool gather_check(const smi_u64* g, int n, unsigned tag, int layer, int edge, int base_index, const EdgeArg& a,
                                             const EngSync& sy) {
  unsigned v[NPT];
  const int tid = threadIdx.x;
  if (!eng_sweep<NPT>(g, tid, kThreads, n, tag, v, sy, (unsigned)(layer * 8 + edge + 1))) return false;
  unsigned nb = 0;
#pragma unroll
  for (int k = 0; k < NPT; ++k)
    if (tid + k * kThreads < n && v[k] != want_val(layer, edge, base_index + tid + k * kThreads)) ++nb;
  if (nb) atomicAdd(a.bad, nb);
  return true;
}

__global__ __launch_bounds__(kThreads) void ub_edge(EdgeArg a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cu = blockIdx.x;
  EngSync sy;
  sy.err = a.err;
  sy.quiet = 0; sy.predelay = 0; sy.go = nullptr; sy.watcher = false;
  sy.t_end = __builtin_amdgcn_s_memrealtime() + a.timeout_ticks;
  const unsigned serial = *(volatile unsigned*)a.serial;
  const unsigned tbase = serial * (unsigned)(a.layers * 8 + 16) + 1u;
  const bool down_cu = cu < kDownCU;
  const int head = cu - kDownCU;                    // 0..13 on the head CUs
  const bool head_cu = head >= 0 && head < kHeads;
  // gate_up parts (2 act values each): 9 on the down CUs, 13 on the others (224 * 9 + 32 * 13 = 2432)
  const int gu0 = down_cu ? cu * 9 : kDownCU * 9 + (cu - kDownCU) * 13, gun = down_cu ? 9 : 13;
  // QKV parts (4 values each): one on the down CUs, two on the others
  const int q0 = down_cu ? cu : kDownCU + 2 * (cu - kDownCU), qn = down_cu ? 1 : 2;
  bool ok = true;
  // layer 0's input: published by the down CUs themselves
  if (down_cu && tid < 4) eng_gstore(a.gran + kOffA + 4 * cu + tid, tbase + 0, want_val(0, 0, 4 * cu + tid));
  size_t dma_pos = ((size_t)cu * 8 + wave) * (size_t)a.dma_kib * 1024 * (size_t)a.layers;
  for (int layer = 0; layer < a.layers && ok; ++layer) {
    smi_u64* gb = a.gran + (size_t)(layer & 1) * kGranPerBuf;
    smi_u64* gnext = a.gran + (size_t)((layer + 1) & 1) * kGranPerBuf;
    const unsigned tl = tbase + (unsigned)layer * 8u;
    const bool st = a.stamps && tid == 0 && (cu == 0 || cu == kDownCU);
    unsigned long long* sp = a.stamps ? a.stamps + ((size_t)(cu == 0 ? 0 : 1) * a.layers + layer) * 8 : nullptr;
    // ---- edge A: h
    ok = gather_check<2>(gb + kOffA, kH, tl + 0, layer, 0, 0, a, sy);
    ok = __syncthreads_and(ok);
    if (st) sp[0] = __builtin_amdgcn_s_memrealtime();
    if (!ok) break;
    // background stream: this layer's share of the weight stand-in (in flight across the edges below)
    for (int i = 0; i < a.dma_kib; ++i) {
      const size_t off = (dma_pos + (size_t)i * 1024) % (a.big_bytes - 4096);
      typedef __attribute__((address_space(3))) void* lptr_t;
      const uint32_t lbase = (uint32_t)(size_t)(lptr_t)smem;   // absolute LDS address of the dynamic segment (static LDS precedes it)
      dma1k(a.big + (off & ~(size_t)1023) + lane * 16, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lbase + wave * 16384 + (i & 15) * 1024)));
    }
    dma_pos += (size_t)a.dma_kib * 1024;
    // QKV "epilogue": publish B
    if (tid < 4 * qn) eng_gstore(gb + kOffB + 4 * q0 + tid, tl + 1, want_val(layer, 1, 4 * q0 + tid));
    // ---- edge B (head CUs): q head, k and v of the head's group
    if (head_cu) {
      const int grp = head / 7;
      bool okb = true;
      if (tid < 192) {
        const int i = tid < 64 ? head * 64 + tid : tid < 128 ? kH + grp * 64 + (tid - 64) : kH + 128 + grp * 64 + (tid - 128);
        unsigned v[1];
        okb = eng_sweep<1>(gb + kOffB, i, 1 << 30, kQKV, tl + 1, v, sy, (unsigned)(layer * 8 + 2));
        if (okb && v[0] != want_val(layer, 1, i)) atomicAdd(a.bad, 1u);
      }
      ok = __syncthreads_and(okb);
      if (st) sp[1] = __builtin_amdgcn_s_memrealtime();
      if (!ok) break;
      if (tid < 64) eng_gstore(gb + kOffC + head * 64 + tid, tl + 2, want_val(layer, 2, head * 64 + tid));
    }
    // ---- edge C (o_proj CUs): attention output
    if (down_cu) {
      ok = gather_check<2>(gb + kOffC, kH, tl + 2, layer, 2, 0, a, sy);
      ok = __syncthreads_and(ok);
      if (st) sp[2] = __builtin_amdgcn_s_memrealtime();
      if (!ok) break;
      if (tid < 4) eng_gstore(gb + kOffD + 4 * cu + tid, tl + 3, want_val(layer, 3, 4 * cu + tid));
    }
    // ---- edge D: h_mid
    ok = gather_check<2>(gb + kOffD, kH, tl + 3, layer, 3, 0, a, sy);
    ok = __syncthreads_and(ok);
    if (st) sp[3] = __builtin_amdgcn_s_memrealtime();
    if (!ok) break;
    if (tid < 2 * gun) eng_gstore(gb + kOffE + 2 * gu0 + tid, tl + 4, want_val(layer, 4, 2 * gu0 + tid));
    // ---- edge E (down CUs): act
    if (down_cu) {
      ok = gather_check<10>(gb + kOffE, kI, tl + 4, layer, 4, 0, a, sy);
      ok = __syncthreads_and(ok);
      if (st) sp[4] = __builtin_amdgcn_s_memrealtime();
      if (!ok) break;
      if (tid < 4) eng_gstore(gnext + kOffA + 4 * cu + tid, tl + 8, want_val(layer + 1, 0, 4 * cu + tid));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the background stream has landed before the LDS is given back
  if (a.layers < 0) a.bad[1] = smem[tid];            // (keeps the dynamic LDS segment referenced)
  // the last arriver bumps the serial (every workgroup has read it: each needed everyone's layer-0 granules)
  __syncthreads();
  if (tid == 0) {
    const unsigned done = atomicAdd(a.err + 2, 1u);
    if (done == (unsigned)a.ncu - 1) { a.err[2] = 0; __threadfence(); atomicAdd(a.serial, 1u); }
  }
}

}  // namespace

// One launch = `layers` layers of the five edges; `launches` back-to-back launches are timed.  err_out[0] = timeout flag,
// [1] = where ((layer * 8 + edge + 1) of the first wave that gave up), [3] = granules with a wrong value.
// stamps_out: [2][layers][8] ticks (10 ns) of the LAST launch: CU 0 and the first head CU, after edges A, B, C, D, E.
extern "C" int smi_diag_edge(int layers, int launches, int dma_kib, const void* big, size_t big_bytes, float timeout_ms,
                             float* us_per_launch, unsigned* err_out, unsigned long long* stamps_out, void* stream) {
  if (layers < 1 || layers > 1024 || launches < 1 || dma_kib < 0 || dma_kib > 16) { smi_set_error("diag_edge: bad shape"); return SMI_EINVAL; }
  int dev = 0, ncu = 0;
  SMI_HIP(hipGetDevice(&dev));
  SMI_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  if (ncu != 256) { smi_set_error("diag_edge: needs 256 CUs, device has %d", ncu); return SMI_EINVAL; }
  if (dma_kib > 0 && (!big || big_bytes < ((size_t)1 << 20))) { smi_set_error("diag_edge: stream buffer too small"); return SMI_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  EdgeArg a;
  memset(&a, 0, sizeof(a));
  const size_t gbytes = (size_t)2 * kGranPerBuf * 8, sbytes = (size_t)2 * layers * 8 * 8;
  unsigned char* mem = nullptr;
  SMI_HIP(hipMalloc((void**)&mem, gbytes + 256 + sbytes));
  SMI_HIP(hipMemsetAsync(mem, 0, gbytes + 256 + sbytes, st));
  a.gran = (smi_u64*)mem; a.serial = (unsigned*)(mem + gbytes); a.err = a.serial + 16; a.bad = a.serial + 32;
  a.stamps = stamps_out ? (unsigned long long*)(mem + gbytes + 256) : nullptr;
  a.big = (const unsigned char*)big; a.big_bytes = big_bytes; a.layers = layers; a.dma_kib = dma_kib; a.ncu = ncu;
  a.timeout_ticks = (unsigned)(timeout_ms * 1e5f);
  const size_t lds = (size_t)144 * 1024;   // one workgroup per CU, as in the engine (__syncthreads_and keeps 256 B of static LDS)
  static bool attr = false;
  if (!attr) { SMI_HIP(hipFuncSetAttribute((const void*)ub_edge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  hipEvent_t e0, e1;
  SMI_HIP(hipEventCreate(&e0)); SMI_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(ub_edge, dim3(ncu), dim3(kThreads), lds, st, a);
  SMI_LAUNCH_CHECK();
  SMI_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(ub_edge, dim3(ncu), dim3(kThreads), lds, st, a);
  SMI_LAUNCH_CHECK();
  SMI_HIP(hipEventRecord(e1, st));
  SMI_HIP(hipEventSynchronize(e1));
  float ms = 0;
  SMI_HIP(hipEventElapsedTime(&ms, e0, e1));
  *us_per_launch = ms * 1e3f / (float)launches;
  unsigned e[4] = {0, 0, 0, 0}, bad = 0;
  SMI_HIP(hipMemcpy(e, a.err, 16, hipMemcpyDeviceToHost));
  SMI_HIP(hipMemcpy(&bad, a.bad, 4, hipMemcpyDeviceToHost));
  err_out[0] = e[0]; err_out[1] = e[1]; err_out[2] = e[2]; err_out[3] = bad;
  if (stamps_out) SMI_HIP(hipMemcpy(stamps_out, a.stamps, sbytes, hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(mem);
  return SMI_OK;
}
