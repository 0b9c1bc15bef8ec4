// diag_ubench.hip -- launch-floor microbenchmarks (diagnostics only; not part of the product path).
// Times hipGraph replays of a chain of N dependent kernels of one kind, to price the fixed cost of a
// kernel boundary on this machine separately from the work inside the kernels.
#include "../smi_common.h"
#include <vector>

namespace {
struct BigArg { const float* in; float* out; int n; int pad[45]; };   // ~200 bytes like GemmP

__global__ void ub_empty() {}
__global__ void ub_touch(BigArg a) {
  if (blockIdx.x == 0 && threadIdx.x == 0) a.out[0] = a.in[0] + 1.0f;
}
__global__ void ub_lds(BigArg a) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = a.in[threadIdx.x & 63];
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) a.out[0] = sm[1] + 1.0f;
}
// every block streams `n` uint4 per thread from a private region, then one dependent write
__global__ void ub_stream(BigArg a) {
  const uint4* p = (const uint4*)a.in + ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
#pragma unroll 8
  for (int i = 0; i < a.n; ++i) { uint4 v = p[i * stride]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) a.out[1] = 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) a.out[0] = a.out[0] + 1.0f;
}
// XCD-affine streaming: a block learns its XCD from HW_REG_XCC_ID and takes the next chunk of that
// XCD's partition (per-XCD atomic counter).  Work blocks stream `cur`; helper blocks stream `nxt`,
// so the next kernel's work blocks on the same XCD find their partition in that XCD's L2.
struct XArg { const uint4* cur; const uint4* nxt; int work_blocks, helper_blocks, loads; unsigned* ctr_cur; unsigned* ctr_nxt; float* out; };
__global__ void ub_xcc(XArg a) {
  __shared__ unsigned slot_s;
  const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 7u;   // HW_REG_XCC_ID[3:0]
  const bool work = (int)blockIdx.x < a.work_blocks;
  if (threadIdx.x == 0) slot_s = atomicAdd(work ? &a.ctr_cur[xcc] : &a.ctr_nxt[xcc], 1u);
  __syncthreads();
  const unsigned slot = slot_s;
  uint32_t acc = 0;
  if (work) {
    const unsigned per = a.work_blocks / 8;
    if (slot < per) {
      const uint4* p = a.cur + ((size_t)(xcc * per + slot) * blockDim.x * a.loads) + threadIdx.x;
#pragma unroll 8
      for (int i = 0; i < a.loads; ++i) { uint4 v = p[(size_t)i * blockDim.x]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
  } else if (a.helper_blocks > 0) {
    const unsigned hper = a.helper_blocks / 8, per = a.work_blocks / 8;
    if (slot < hper) {
      // this XCD's partition of nxt: per * blockDim * loads uint4, split over hper helpers
      const size_t part = (size_t)per * blockDim.x * a.loads;
      const uint4* p = a.nxt + (size_t)xcc * part;
      for (size_t i = (size_t)slot * blockDim.x + threadIdx.x; i < part; i += (size_t)hper * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y; }
    }
  }
  if (acc == 0x12345678u) a.out[1] = 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) a.out[0] = a.out[0] + 1.0f;
}
}  // namespace

// chain of n kernels; kernel i streams region i (cold), and with helpers > 0 also prefetches region i+1
extern "C" int smi_ubench_xcc(int work_blocks, int helper_blocks, int block, int loads, int n_kernels, int iters,
                              const void* buf, size_t buf_bytes, unsigned* ctrs /* [n_kernels+1][8] */, float* scratch,
                              float* us_per_kernel, void* stream) {
  const size_t region = (size_t)work_blocks * block * loads * 16;
  if (work_blocks % 8 || helper_blocks % 8 || region * 2 > buf_bytes) { smi_set_error("ubench_xcc: bad sizes"); return SMI_EINVAL; }
  const size_t nreg = buf_bytes / region;
  hipStream_t st = (hipStream_t)stream, cs;
  SMI_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
  SMI_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
  SMI_HIP(hipMemsetAsync(ctrs, 0, (size_t)(2 * n_kernels + 2) * 8 * 4, cs));
  for (int i = 0; i < n_kernels; ++i) {
    XArg a;
    a.cur = (const uint4*)((const char*)buf + ((size_t)i % nreg) * region);
    a.nxt = (const uint4*)((const char*)buf + ((size_t)(i + 1) % nreg) * region);
    a.work_blocks = work_blocks; a.helper_blocks = helper_blocks; a.loads = loads;
    a.ctr_cur = ctrs + (size_t)i * 8; a.ctr_nxt = ctrs + (size_t)(n_kernels + 1 + i) * 8; a.out = scratch;
    hipLaunchKernelGGL(ub_xcc, dim3(work_blocks + helper_blocks), dim3(block), 0, cs, a);
  }
  SMI_HIP(hipStreamEndCapture(cs, &g));
  SMI_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  SMI_HIP(hipEventCreate(&e0)); SMI_HIP(hipEventCreate(&e1));
  SMI_HIP(hipGraphLaunch(ge, st));
  SMI_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) SMI_HIP(hipGraphLaunch(ge, st));
  SMI_HIP(hipEventRecord(e1, st));
  SMI_HIP(hipEventSynchronize(e1));
  float ms = 0;
  SMI_HIP(hipEventElapsedTime(&ms, e0, e1));
  *us_per_kernel = ms * 1e3f / ((float)iters * n_kernels);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(cs);
  return SMI_OK;
}

// Records HW_REG_XCC_ID of every block of a chain of kernels with the given grid sizes (graph replay),
// to see how the dispatcher maps blocks to XCDs from one kernel to the next.
__global__ void ub_xccmap(unsigned char* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned char)(__builtin_amdgcn_s_getreg(6164) & 7u);
}
extern "C" int smi_ubench_xccmap(const int* grids, const int* blocks, int n, int reps, unsigned char* out_dev, int stride, void* stream) {
  hipStream_t st = (hipStream_t)stream, cs;
  SMI_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
  SMI_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
  for (int r = 0; r < reps; ++r)
    for (int i = 0; i < n; ++i)
      hipLaunchKernelGGL(ub_xccmap, dim3(grids[i]), dim3(blocks[i]), 0, cs, out_dev + (size_t)(r * n + i) * stride);
  SMI_HIP(hipStreamEndCapture(cs, &g));
  SMI_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  SMI_HIP(hipGraphLaunch(ge, st));
  SMI_HIP(hipGraphLaunch(ge, st));
  SMI_HIP(hipStreamSynchronize(st));
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(cs);
  return SMI_OK;
}

// rotate_bytes > 0: kernel i of the chain reads its own region (i * rotate_bytes) mod buf_bytes, so with
// buf_bytes well above the 256 MB Infinity Cache every kernel streams cold HBM data.
extern "C" int smi_ubench_chain2(int kind, int grid, int block, int lds_bytes, int loads_per_thread, int n_kernels, int iters,
                                 const void* buf, size_t buf_bytes, size_t rotate_bytes, float* scratch, float* us_per_kernel,
                                 void* stream) {
  const size_t need = (size_t)grid * block * (loads_per_thread > 0 ? loads_per_thread : 1) * 16;
  if (kind >= 3 && (need > buf_bytes || (rotate_bytes && rotate_bytes < need))) { smi_set_error("ubench: region too small"); return SMI_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  hipStream_t cs;
  SMI_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  BigArg a;
  a.out = scratch; a.n = loads_per_thread;
  const size_t nreg = rotate_bytes ? buf_bytes / rotate_bytes : 1;
  SMI_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n_kernels; ++i) {
    a.in = (const float*)((const char*)buf + (rotate_bytes ? ((size_t)i % nreg) * rotate_bytes : 0));
    switch (kind) {
      case 0: hipLaunchKernelGGL(ub_empty, dim3(grid), dim3(block), 0, cs); break;
      case 1: hipLaunchKernelGGL(ub_touch, dim3(grid), dim3(block), 0, cs, a); break;
      case 2: hipLaunchKernelGGL(ub_lds, dim3(grid), dim3(block), lds_bytes, cs, a); break;
      default: hipLaunchKernelGGL(ub_stream, dim3(grid), dim3(block), 0, cs, a); break;
    }
  }
  SMI_HIP(hipStreamEndCapture(cs, &g));
  SMI_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  SMI_HIP(hipEventCreate(&e0));
  SMI_HIP(hipEventCreate(&e1));
  SMI_HIP(hipGraphLaunch(ge, st));
  SMI_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) SMI_HIP(hipGraphLaunch(ge, st));
  SMI_HIP(hipEventRecord(e1, st));
  SMI_HIP(hipEventSynchronize(e1));
  float ms = 0;
  SMI_HIP(hipEventElapsedTime(&ms, e0, e1));
  *us_per_kernel = ms * 1e3f / ((float)iters * n_kernels);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(cs);
  return SMI_OK;
}

extern "C" int smi_ubench_chain(int kind, int grid, int block, int lds_bytes, int loads_per_thread, int n_kernels, int iters,
                                const void* buf, float* scratch, float* us_per_kernel, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipStream_t cs;
  SMI_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  BigArg a;
  a.in = (const float*)buf; a.out = scratch; a.n = loads_per_thread;
  SMI_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n_kernels; ++i) {
    switch (kind) {
      case 0: hipLaunchKernelGGL(ub_empty, dim3(grid), dim3(block), 0, cs); break;
      case 1: hipLaunchKernelGGL(ub_touch, dim3(grid), dim3(block), 0, cs, a); break;
      case 2: hipLaunchKernelGGL(ub_lds, dim3(grid), dim3(block), lds_bytes, cs, a); break;
      default: hipLaunchKernelGGL(ub_stream, dim3(grid), dim3(block), 0, cs, a); break;
    }
  }
  SMI_HIP(hipStreamEndCapture(cs, &g));
  SMI_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  SMI_HIP(hipEventCreate(&e0));
  SMI_HIP(hipEventCreate(&e1));
  SMI_HIP(hipGraphLaunch(ge, st));
  SMI_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) SMI_HIP(hipGraphLaunch(ge, st));
  SMI_HIP(hipEventRecord(e1, st));
  SMI_HIP(hipEventSynchronize(e1));
  float ms = 0;
  SMI_HIP(hipEventElapsedTime(&ms, e0, e1));
  *us_per_kernel = ms * 1e3f / ((float)iters * n_kernels);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(cs);
  return SMI_OK;
}
