// smi_ubench.hip -- launch-floor microbenchmarks (diagnostics only; not part of the product path).
// Times hipGraph replays of a chain of N dependent kernels of one kind, to price the fixed cost of a
// kernel boundary on this machine separately from the work inside the kernels.
#include "smi_common.h"
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
}  // namespace

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
