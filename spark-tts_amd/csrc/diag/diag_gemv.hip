// diag_gemv.hip -- one-row GEMV on the VALU against the matrix pipe (diagnostics only; not part of the product path).
//
// The A/B round 2's review asked for: at M = 1, y[n] = sum_k W[n][k] x[k] with
//   variant 0  MFMA   the product's arithmetic: x as exact bf16 triples, hi / mid / lo in three columns of ONE
//                     v_mfma_f32_16x16x32_bf16 per weight tile, chains k tile -> wave (kt mod NW), (lo + mid) + hi
//   variant 1  VALU   fp32 FMAs on bf16-expanded weights, x in fp32 (no triples): 8 unpacks + 8 v_fma_f32 per 16-byte piece,
//                     products consumed as the pieces land
//   variant 2  DOT2   v_dot2_f32_bf16 on x rounded to bf16 (NOT exact: the cheapest the VALU route can be; timing bound only)
//   variant 3  MFMA4  (down_proj) the matrix pipe with full load instructions: four chains per wave and MFMA (ub_gemv4 below)
// on the product's weight layouts, grids and wave counts:
//   shape 0  down_proj  N = 896, K = 4864: 56 tiles x 4 row parts = 224 blocks of 16 waves, W_down stored row-part-major
//            (a part's 16 pieces of a k tile are 256 contiguous bytes), so a VALU wave covers FOUR k tiles per load
//            instruction with all 64 lanes busy, where the MFMA wave feeds zeros to 48 lanes;
//   shape 1  QKV        N = 1152, K = 896: 72 blocks of 16 waves, one tile (1 KiB) per wave-instruction either way.
// Each launch reads its own copy of the weights (`copies` of them, walked round-robin: more than the 256 MB Infinity
// Cache holds), eager back-to-back launches like smi_llm_time_kernel's stand-alone probe.
#include "../smi_common.h"
#include <string.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 gbf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 gbf16x2;
typedef __attribute__((ext_vector_type(4))) float gf32x4;

struct GemvArg {
  const uint4* W;        // [NT][KT][64] 16-byte pieces
  const unsigned char* XS;   // variant 0: triples [KT][3][4][16 B]
  const float* X;        // variants 1, 2: fp32 x [K]
  float* Y;              // [N]
  int NT, KT;
};

__device__ __forceinline__ uint4 ldw(const uint4* q) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)q);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float lo16(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float hi16(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }

// PARTS = 4: blockIdx.x = part * NT + tile (row-part-major weights); PARTS = 1: blockIdx.x = tile.  TPW = k tiles per wave (upper bound)
template <int VAR, int PARTS, int NW, int TPW>
__global__ __launch_bounds__(NW * 64) void ub_gemv(GemvArg a) {
  __shared__ float red[NW * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KT = a.KT;
  const int nt = PARTS > 1 ? (int)blockIdx.x % a.NT : (int)blockIdx.x;
  const int part = PARTS > 1 ? (int)blockIdx.x / a.NT : 0;
  const int k8 = VAR == 0 ? lane >> 4 : (PARTS > 1 ? (lane >> 2) & 3 : lane >> 4);
  const uint4* wt = a.W + (size_t)nt * KT * 64;
  if constexpr (VAR == 0) {
    // ---- the product's form: one MFMA per tile, columns 0 / 1 / 2 = hi / mid / lo
    const bool wact = PARTS == 1 || ((lane & 15) >> 2) == part;
    const int wl = PARTS > 1 ? ((lane & 12) << 2) + ((lane >> 4) << 2) + (lane & 3) : lane;
    const int col = (lane & 15) < 3 ? (lane & 15) : 0;
    uint4 w[TPW];
    gbf16x8 b[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      int t = wave + u * NW;
      t = t < KT ? t : KT - 1;
      w[u] = wact ? ldw(wt + (size_t)t * 64 + wl) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      int t = wave + u * NW;
      t = t < KT ? t : KT - 1;
      b[u] = *(const gbf16x8*)(a.XS + ((size_t)(t * 3 + col) * 4 + k8) * 16);
    }
    gf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (wave + u * NW < KT) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gbf16x8, w[u]), b[u], acc, 0, 0, 0);
    float4 t;
    t.x = (smi_dpp<0x102>(acc[0]) + smi_dpp<0x101>(acc[0])) + acc[0];
    t.y = (smi_dpp<0x102>(acc[1]) + smi_dpp<0x101>(acc[1])) + acc[1];
    t.z = (smi_dpp<0x102>(acc[2]) + smi_dpp<0x101>(acc[2])) + acc[2];
    t.w = (smi_dpp<0x102>(acc[3]) + smi_dpp<0x101>(acc[3])) + acc[3];
    ((float4*)red)[wave * 64 + lane] = t;
    __syncthreads();
    if (wave == 0) {
      float4 s = ((float4*)red)[lane];
#pragma unroll
      for (int wv = 1; wv < NW; ++wv) {
        const float4 q = ((float4*)red)[wv * 64 + lane];
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
      }
      // D layout: lane (col = lane & 15, row group lane >> 4) holds rows 4 * (lane >> 4) .. + 3 of column col; column 0 is the row
      const bool ract = PARTS == 1 || (lane >> 4) == part;
      if ((lane & 15) == 0 && ract) *(float4*)(a.Y + nt * 16 + 4 * (lane >> 4)) = s;
    }
  } else {
    // ---- VALU: lane = one 16-byte piece (8 consecutive k of one weight row)
    constexpr int TPI = PARTS > 1 ? 4 : 1;            // k tiles one wave-instruction covers
    constexpr int G = (TPW + TPI - 1) / TPI;          // load groups per wave
    const int j = PARTS > 1 ? lane >> 4 : 0;          // which of the instruction's tiles
    const int piece = PARTS > 1 ? part * 16 + (lane & 15) : lane;   // row-part-major: piece = part * 16 + k8 * 4 + r
    uint4 w[G];
    float4 x0[G], x1[G];
    bool ok[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      int t = wave + NW * (g * TPI + j);
      ok[g] = t < KT;
      t = ok[g] ? t : KT - 1;
      w[g] = ldw(wt + (size_t)t * 64 + piece);
      if (!ok[g]) w[g] = make_uint4(0u, 0u, 0u, 0u);   // (a tile past the end contributes zeros: no branch around the arithmetic)
      x0[g] = *(const float4*)(a.X + t * 32 + k8 * 8);
      x1[g] = *(const float4*)(a.X + t * 32 + k8 * 8 + 4);
    }
    float acc0 = 0.f, acc1 = 0.f;
    asm volatile("" ::: "memory");                    // every load above is issued before the first use below
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if constexpr (VAR == 1) {
        acc0 = fmaf(lo16(w[g].x), x0[g].x, acc0); acc1 = fmaf(hi16(w[g].x), x0[g].y, acc1);
        acc0 = fmaf(lo16(w[g].y), x0[g].z, acc0); acc1 = fmaf(hi16(w[g].y), x0[g].w, acc1);
        acc0 = fmaf(lo16(w[g].z), x1[g].x, acc0); acc1 = fmaf(hi16(w[g].z), x1[g].y, acc1);
        acc0 = fmaf(lo16(w[g].w), x1[g].z, acc0); acc1 = fmaf(hi16(w[g].w), x1[g].w, acc1);
      } else {
        auto pk = [](float lo, float hi) { return __builtin_bit_cast(gbf16x2, (smi_f32_to_bf16(lo)) | (smi_f32_to_bf16(hi) << 16)); };
        acc0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gbf16x2, w[g].x), pk(x0[g].x, x0[g].y), acc0, false);
        acc1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gbf16x2, w[g].y), pk(x0[g].z, x0[g].w), acc1, false);
        acc0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gbf16x2, w[g].z), pk(x1[g].x, x1[g].y), acc0, false);
        acc1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gbf16x2, w[g].w), pk(x1[g].z, x1[g].w), acc1, false);
      }
    }
    float s = acc0 + acc1;
    if constexpr (PARTS > 1) {   // lanes of one row r = lane & 3: sum over k8 (bits 2, 3) and the instruction's tiles (bits 4, 5)
      s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      if (lane < 4) red[wave * 4 + lane] = s;
      __syncthreads();
      if (tid < 4) {
        float y = red[tid];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) y += red[wv * 4 + tid];
        a.Y[nt * 16 + part * 4 + tid] = y;
      }
    } else {                     // row n = lane & 15: sum over k8 (bits 4, 5)
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      if (lane < 16) red[wave * 16 + lane] = s;
      __syncthreads();
      if (tid < 16) {
        float y = red[tid];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) y += red[wv * 16 + tid];
        a.Y[nt * 16 + tid] = y;
      }
    }
  }
}


// variant 3 (down_proj only): the matrix pipe with FULL load instructions -- a wave owns four chains (k tile -> chain kt mod 16,
// as the product) and advances all four with ONE MFMA: A row 4s + r = weight row r of the part, k tile of chain 4w + s; B column
// 3s + c = split c of that tile's x; the four 4 x 3 diagonal blocks of D are the chains' accumulators (the other blocks mix
// tiles and are never read).  A chain's sum and the in-order sum over the 16 chains are the product's, bit for bit; what changes
// is that one load instruction carries 1 KiB (four tiles' parts) instead of 256 B with 48 idle lanes.  NWV waves per block,
// 16 / NWV ... (4 waves: four chains each).
template <int TPC, bool WPERM>
__global__ __launch_bounds__(256) void ub_gemv4(GemvArg a) {
  __shared__ float red[16 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KT = a.KT;
  const int nt = (int)blockIdx.x % a.NT, part = (int)blockIdx.x / a.NT;
  const uint4* wt = a.W + (size_t)nt * KT * 64;
  const int row = lane & 15, k8 = lane >> 4;
  const int sa = row >> 2, r = row & 3;               // A: chain slot and weight row of this lane's row
  const int col = row, sb = col < 12 ? col / 3 : 0, c = col < 12 ? col % 3 : 0;   // B: chain slot and split of this lane's column
  uint4 w[TPC];
  gbf16x8 b[TPC];
#pragma unroll
  for (int u = 0; u < TPC; ++u) {
    int t = 4 * wave + sa + 16 * u;
    const bool ok = t < KT;
    t = ok ? t : KT - 1;
    w[u] = ldw(wt + (size_t)t * 64 + (WPERM ? part * 16 + k8 * 4 + r : k8 * 16 + part * 4 + r));
    if (!ok) w[u] = make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int u = 0; u < TPC; ++u) {
    int t = 4 * wave + sb + 16 * u;
    t = t < KT ? t : KT - 1;
    b[u] = *(const gbf16x8*)(a.XS + ((size_t)(t * 3 + c) * 4 + k8) * 16);
  }
  gf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < TPC; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gbf16x8, w[u]), b[u], acc, 0, 0, 0);
  // D: lane (col, row group g = lane >> 4) holds rows 4g .. 4g + 3 of column col; chain slot s lives in row group s, columns 3s .. 3s + 2
  float t4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) t4[e] = (smi_dpp<0x102>(acc[e]) + smi_dpp<0x101>(acc[e])) + acc[e];
  if (row == 3 * k8)   // column 3s of row group s: chain 4 * wave + s, rows r = 0..3 in t4
    *(float4*)(red + (4 * wave + k8) * 4) = make_float4(t4[0], t4[1], t4[2], t4[3]);
  __syncthreads();
  if (tid < 4) {
    float y = red[tid];
#pragma unroll
    for (int ch = 1; ch < 16; ++ch) y += red[ch * 4 + tid];
    a.Y[nt * 16 + part * 4 + tid] = y;
  }
}

template <int VAR>
void launch(int shape, const GemvArg& a, hipStream_t st) {
  if (shape == 0) hipLaunchKernelGGL((ub_gemv<VAR, 4, 16, 10>), dim3(a.NT * 4), dim3(1024), 0, st, a);
  else hipLaunchKernelGGL((ub_gemv<VAR, 1, 16, 2>), dim3(a.NT), dim3(1024), 0, st, a);
}

}  // namespace

// W: `copies` weight copies of NT * KT KiB each, back to back; returns microseconds per launch (eager, back to back) and the
// last launch's y (so that the caller can check variant 0 against variant 1 on its own inputs).
extern "C" int smi_diag_gemv(int variant, int shape, const void* W, int copies, const void* XS, const float* X, float* Y, int iters,
                             float* us_per_launch, void* stream) {
  if (variant < 0 || variant > 3 || shape < 0 || shape > 1 || copies < 1 || iters < 1 || !W || !XS || !X || !Y || !us_per_launch) {
    smi_set_error("diag_gemv: bad argument");
    return SMI_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  GemvArg a;
  memset(&a, 0, sizeof(a));
  a.NT = shape == 0 ? 56 : 72; a.KT = shape == 0 ? 152 : 28;
  a.XS = (const unsigned char*)XS; a.X = X; a.Y = Y;
  const size_t per = (size_t)a.NT * a.KT * 64;   // uint4 per copy
  hipEvent_t e0, e1;
  SMI_HIP(hipEventCreate(&e0)); SMI_HIP(hipEventCreate(&e1));
  auto go = [&](int i) {
    a.W = (const uint4*)W + (size_t)(i % copies) * per;
    if (variant == 0) launch<0>(shape, a, st); else if (variant == 1) launch<1>(shape, a, st); else if (variant == 2) launch<2>(shape, a, st);
    else if (shape == 0) hipLaunchKernelGGL((ub_gemv4<10, true>), dim3(a.NT * 4), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((ub_gemv4<2, false>), dim3(a.NT * 4), dim3(256), 0, st, a);   // QKV: plain tile order, four 64-byte segments per part
  };
  for (int i = 0; i < copies; ++i) go(i);
  SMI_LAUNCH_CHECK();
  SMI_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) go(i);
  SMI_LAUNCH_CHECK();
  SMI_HIP(hipEventRecord(e1, st));
  SMI_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  SMI_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *us_per_launch = ms * 1e3f / iters;
  return SMI_OK;
}
