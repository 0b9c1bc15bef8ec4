// smi_net.h -- kernels and launch builders shared by the BiCodec vocoder (smi_voc.hip) and the
// prompt encoder (smi_enc.hip): the implicit-GEMM conv / linear kernel on the exact-fp32 matrix pipe,
// depthwise-conv + LayerNorm, the per-utterance GEMV, and the arena entry / conv geometry helpers.
// Everything is in an anonymous namespace: each translation unit gets its own copy of the device code.
#pragma once
#include "smi_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <string>
#include <vector>

namespace {

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_TANH = 2, ACT_RELU = 3, ACT_SIGMOID = 4 };
enum { PACK_RAW = 0, PACK_CONV = 1, PACK_CONVT = 2, PACK_CONV_B = 3, PACK_CONVT_B = 4 };   // _B: two bf16 planes (hi, mid) for k_convb
constexpr int kMaxTaps = 8;     // taps per phase
constexpr int kMaxPhases = 8;   // ConvTranspose1d stride
constexpr int kChunk = 32;      // input channels staged per LDS chunk

struct ConvP {
  const float* X;       // [B][Cin][xstride]
  const float* X2;      // second input added to X while staging (same strides; Res2Net branch sum) or null
  const float* W;       // packed, phase-major
  const float* bias;    // [Cout] or null
  const float* bbias;   // [B][Cout] per-utterance bias or null
  const float* gamma;   // [Cout] layer scale or null
  const float* beta;    // [Cout] shift after gamma (BatchNorm in eval mode: y = gamma * act(conv) + beta) or null
  const float* R;       // residual [B][Cout][ystride] or null (may alias Y)
  const float* alpha;   // [Cout] Snake alpha for Ys or null
  float* Y;             // raw output or null
  float* Ys;            // snake(Y, alpha) or null
  const int* lens;      // [B] valid INPUT lengths
  const int* olens;     // [B] valid OUTPUT positions per phase when they differ from lens (strided conv), or null
  int Cin, CinP, Cout, S, act;
  int xstride, ystride;
  long long xb, yb;     // batch strides (floats)
  int halo_l, xw;       // left halo, staged row width (floats)
  int istr;             // input stride of a strided Conv1d (1 otherwise): output q reads x[q * istr + tap]
  float out_scale;
  int fast_sin;         // Snake with the hardware sine (layers on the bf16-split pipe)
  int ntaps[kMaxPhases];
  int off[kMaxPhases][kMaxTaps];
  long long wphase[kMaxPhases];  // float offset of each phase's weights
};

// fast: the hardware sine (v_sin_f32 on x / 2 pi, ~1e-6 absolute) -- the layers of the bf16-split pipe, whose products already
// carry 2^-17 relative error; the library sine costs ~2500 of the ~4000 vector instructions a thread of a conv block issues
// (32 Snake values per thread), which is what bounded the one-tap layers and a third of the 7-tap ones.
__device__ __forceinline__ float snake_f(float x, float a, bool fast) {
  const float s = fast ? __sinf(a * x) : sinf(a * x);
  return x + (1.0f / (a + 1e-9f)) * (s * s);
}
__device__ __forceinline__ float gelu_f(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }

// Shared tail of the conv kernels: (KS) fixed-order reduction of the four waves' partial tiles through LDS, then bias,
// per-utterance bias, activation, layer scale, residual, tanh, the 3x of SamplingBlock(ratio 1), and the two outputs
// (raw and Snake'd).  acc follows the 32x32 MFMA D layout: register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5),
// column l & 31.
template <int QB, bool KS>
__device__ __forceinline__ void conv_finish(const ConvP& p, f32x16 (&acc)[QB], float* lds, int ct, bool live, int b, int phase,
                                            int q0, int olen, int lane, int wave) {
  if (KS) {
    // fixed-order reduction of the four waves' partial tiles, then each wave finishes 4 of the 16 rows-groups
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int r = 0; r < 16; ++r) lds[((wave * QB + qb) * 16 + r) * 64 + lane] = acc[qb][r];
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = wave * 4 + rr;
        float s = lds[((0 * QB + qb) * 16 + r) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) s += lds[((w * QB + qb) * 16 + r) * 64 + lane];
        acc[qb][rr] = s;
      }
  }
  if (!live) return;

  // Epilogue loads first, arithmetic second.  Written element by element (`if (p.bias) y += p.bias[co]; ... if (p.R) y = p.R[o] + y;`)
  // hipcc branches around every load and waits for it on the spot: up to six dependent round trips per element, ~190
  // `s_waitcnt vmcnt(0)` per block -- the one-tap layers then ran at 1.3 TB/s of traffic with blocks alive for ~50 us.  Here
  // every load of eight output rows is unconditional (an absent operand reads a valid dummy address and is dropped by a
  // select), so they are all in flight together; the arithmetic and its order are unchanged.
  const long long yboff = (long long)b * p.yb;
  constexpr int NREG = KS ? 4 : 16, GRP = KS ? 4 : 8;
  const bool hb = p.bias != nullptr, hbb = p.bbias != nullptr, hg = p.gamma != nullptr, hbe = p.beta != nullptr, hr = p.R != nullptr,
             hs = p.Ys != nullptr;
  const bool fsin = p.fast_sin != 0;
  const float* dummy = p.X;                       // any readable floats: index < Cout <= the first activation row's size
  const float* pb = hb ? p.bias : dummy;
  const float* pbb = hbb ? p.bbias + (long long)b * p.Cout : dummy;
  const float* pg = hg ? p.gamma : dummy;
  const float* pbe = hbe ? p.beta : dummy;
  const float* pa = hs ? p.alpha : dummy;
  bool qok[QB];
  int tq[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q0 + qb * 32 + (lane & 31);
    qok[qb] = q < olen;
    tq[qb] = (qok[qb] ? q : olen - 1) * p.S + phase;   // (q0 < olen: the block has at least one live column)
  }
#pragma unroll
  for (int r0 = 0; r0 < NREG; r0 += GRP) {
    int co[GRP];
    bool cok[GRP];
    float bv[GRP], bbv[GRP], gv[GRP], bev[GRP], av[GRP], rv[QB][GRP];
#pragma unroll
    for (int i = 0; i < GRP; ++i) {
      const int r = KS ? wave * 4 + r0 + i : r0 + i;
      co[i] = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      cok[i] = co[i] < p.Cout;
      const int cc = cok[i] ? co[i] : p.Cout - 1;
      bv[i] = pb[cc]; bbv[i] = pbb[cc]; gv[i] = pg[cc]; bev[i] = pbe[cc]; av[i] = pa[cc];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float* rp = hr ? p.R + (yboff + (long long)cc * p.ystride + tq[qb]) : dummy;
        rv[qb][i] = *rp;
      }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
      for (int i = 0; i < GRP; ++i) {
        float y = acc[qb][r0 + i];
        y = hb ? y + bv[i] : y;
        y = hbb ? y + bbv[i] : y;
        if (p.act == ACT_GELU) y = gelu_f(y);
        if (p.act == ACT_RELU) y = fmaxf(y, 0.f);
        y = hg ? gv[i] * y : y;
        y = hbe ? y + bev[i] : y;
        y = hr ? rv[qb][i] + y : y;
        if (p.act == ACT_TANH) y = tanhf(y);
        if (p.out_scale != 1.0f) y = (y + y) + y;  // SamplingBlock(ratio 1): x + x + x
        if (qok[qb] && cok[i]) {
          const long long o = yboff + (long long)co[i] * p.ystride + tq[qb];
          if (p.Y) p.Y[o] = y;
          if (hs) p.Ys[o] = snake_f(y, av[i], fsin);
        }
      }
    }
  }
}

// QB: 32-wide time sub-tiles per wave.  KS: waves split the input channels of ONE 32-row output
// tile (large C, short T) instead of owning a 32-row output tile each.  CHG: a staged chunk holds
// 32*CHG input channels (1-tap layers use 128 so a chunk carries enough MFMAs per barrier).  NC:
// the staged row (tile * input stride + halo) is up to 64 * NC columns wide.  NWV: waves per block (3 when the number of
// 32-row output tiles is a multiple of 3 but not of 4 -- C = 96, 192: with four waves a quarter of every block, and so one
// SIMD of every CU, would idle).
template <int QB, bool KS, int CHG, int NC, bool WPF = false, int NWV = 4>
__global__ __launch_bounds__(256) void k_conv(ConvP p) {
  static_assert(NWV == 4 || !KS, "the channel-split mode reduces over four waves");
  constexpr int kCh = 8 * NWV * CHG;   // channels per staged chunk: eight rows per wave and CHG
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int QT = QB * 32;
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int b = bz / p.S, phase = bz - b * p.S;
  const int q0 = bx * QT;
  const int len = p.lens[b];
  const int olen = p.olens ? p.olens[b] : len;
  if (q0 >= olen) return;
  const int ct = KS ? by : by * NWV + wave;
  const bool live = ct * 32 < p.Cout;
  const int ntap = p.ntaps[phase];
  const int groups = p.CinP >> 3;
  const float* Xb = p.X + (long long)b * p.xb;
  const float4* Wp = (const float4*)(p.W + p.wphase[phase]) + (long long)ct * ntap * groups * 64;

  f32x16 acc[QB];
#pragma unroll
  for (int i = 0; i < QB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int xw = p.xw;
  // Staging: wave w owns rows 8w..8w+7 of the 32-channel chunk, lanes run along time (coalesced rows).
  // The next chunk's rows are requested before the MFMAs of the current one and written to LDS after
  // them, so global latency hides behind the matrix pipe.  With KS each wave multiplies exactly the
  // rows it staged, so only the wave itself has to see its LDS writes (no block barrier).
  constexpr int RW = 8 * CHG;   // rows staged per wave
  float sreg[RW][NC];
  auto stage_load = [&](int c0) {
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int ci = c0 + wave * RW + r;
      const float* xr = Xb + (long long)ci * p.xstride;
      const float* x2r = p.X2 ? p.X2 + (long long)b * p.xb + (long long)ci * p.xstride : nullptr;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const int col = lane + 64 * k, t = q0 * p.istr - p.halo_l + col;
        const bool ok = ci < p.Cin && col < xw && t >= 0 && t < len;
        float v = ok ? xr[t] : 0.f;
        if (x2r && ok) v += x2r[t];
        sreg[r][k] = v;
      }
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      float* lr = lds + (wave * RW + r) * xw;
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (lane + 64 * k < xw) lr[lane + 64 * k] = sreg[r][k];
    }
  };
  if constexpr (KS && CHG == 4 && WPF) {
    // One-tap layers with the channels split over the waves, launched with so few blocks that a SIMD holds one wave
    // (short sequences: encoder FFNs, point-wise convs; with more blocks occupancy hides the latency and the extra
    // registers only cost).  A
    // wave's weights for a chunk are CHG float4 per lane; they are requested one chunk ahead together with the
    // staged rows, so the L2 latency of both hides behind the previous chunk's MFMAs instead of sitting in front of
    // every group of four.  Channels past CinP multiply zero weights (same accumulation order, same bits).
    float4 wn[CHG];
    auto wload = [&](int c0) {
#pragma unroll
      for (int g = 0; g < CHG; ++g) {
        const int gi = (c0 >> 3) + wave * CHG + g;
        wn[g] = (live && gi < groups) ? Wp[(long long)gi * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    stage_load(0);
    wload(0);
    const int col0 = p.halo_l + p.off[phase][0] + (lane & 31) * p.istr;
    for (int c0 = 0; c0 < p.CinP; c0 += kCh) {
      if (c0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stage_store();
      float4 wc[CHG];
#pragma unroll
      for (int g = 0; g < CHG; ++g) wc[g] = wn[g];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (c0 + kCh < p.CinP) { stage_load(c0 + kCh); wload(c0 + kCh); }
#pragma unroll
      for (int g = 0; g < CHG; ++g) {
        const float wa[4] = {wc[g].x, wc[g].y, wc[g].z, wc[g].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* xr = lds + ((wave * CHG + g) * 8 + j * 2 + (lane >> 5)) * xw + col0;
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], xr[qb * 32 * p.istr], acc[qb], 0, 0, 0);
        }
      }
    }
  } else {
  stage_load(0);
  for (int c0 = 0; c0 < p.CinP; c0 += kCh) {
    if (c0) { if (KS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else __syncthreads(); }   // previous chunk fully read
    stage_store();
    if (KS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else __syncthreads();
    if (c0 + kCh < p.CinP) stage_load(c0 + kCh);
    if (live) {
      // (tap, 8-channel group) steps in order over this wave's groups of the chunk (all of them, or with KS its own CHG),
      // the NEXT step's packed weights requested before this step's MFMAs and fenced there (the plain loop loads a float4,
      // waits an L2 round trip, issues 4 * QB MFMAs, and leaves hiding that latency to the other waves of the SIMD)
      typedef float f32x4v __attribute__((ext_vector_type(4)));
      const int gbase = KS ? wave * CHG : 0, gcap = KS ? CHG : NWV * CHG;
      int gl = (p.CinP - c0 + 7) / 8 - gbase;   // live groups of this chunk from gbase on
      gl = gl < gcap ? gl : gcap;
      const int nstep = gl > 0 ? ntap * gl : 0;
      const f32x4v* Wq = (const f32x4v*)Wp + ((long long)(c0 >> 3) + gbase) * 64 + lane;
      // two weight registers used in turn (see k_convb's step loop: one register rotated with `wv = wn` makes hipcc wait for the
      // weights it has just requested in the middle of the current step)
      auto advance = [&](int& t, int& gg) { if (++gg == gl) { gg = 0; ++t; } };
      auto wptr = [&](int t, int gg) { return Wq + ((long long)(t < ntap ? t : ntap - 1) * groups + gg) * 64; };
      auto step = [&](const f32x4v& wv, int t, int gg) {
        const int col0 = p.halo_l + p.off[phase][t] + (lane & 31) * p.istr;
        const float wa[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* xr = lds + ((gbase + gg) * 8 + j * 2 + (lane >> 5)) * xw + col0;
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], xr[qb * 32 * p.istr], acc[qb], 0, 0, 0);
        }
      };
      if (nstep > 0) {
        int ta = 0, ga = 0;
        f32x4v wa_ = Wq[0];
        for (int st = 0; st < nstep; st += 2) {
          int tb = ta, gb = ga;
          advance(tb, gb);
          const f32x4v wb_ = *wptr(tb, gb);
          __builtin_amdgcn_sched_barrier(0);
          step(wa_, ta, ga);
          ta = tb; ga = gb;
          advance(ta, ga);
          wa_ = *wptr(ta, ga);
          __builtin_amdgcn_sched_barrier(0);
          if (st + 1 < nstep) step(wb_, tb, gb);
        }
      }
    }
  }

  }
  conv_finish<QB, KS>(p, acc, lds, ct, live, b, phase, q0, olen, lane, wave);
}


// ------------------------------------------------------------------------------------------
// k_convb: the same implicit GEMM on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the rate of the exact-fp32
// MFMA) with fp32 accuracy kept by a 2-plane split of BOTH operands: x = x_hi + x_mid (+ 2^-16 |x|), w likewise, and
//   w x ~= w_hi x_hi + w_hi x_mid + w_mid x_hi          (three products; the dropped terms are ~2^-17 relative)
// accumulated in fp32.  On the 0.5B vocoder this moves the waveform by 5e-5 max-abs against the fp32 path (CPU emulation
// and GPU test; north_star allows 1e-3).  Weights are split at pack time (two bf16 planes in A-operand order:
// [ct][tap][16-channel step][plane][lane][8 bf16], lane l = row l & 31, channels 8 (l >> 5) ..+7 of the step);
// activations are split while a chunk is staged: the thread that loaded eight consecutive channels of one time column packs
// them into one 16-byte B-operand piece per plane, LDS image [plane][octet][column][8 bf16].
// QB / KS / CHG / NC as k_conv (four waves; KS needs CHG >= 2 so that a wave owns whole 16-channel steps).
// ------------------------------------------------------------------------------------------
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {   // round-to-nearest-even pair -> packed bf16 (v_cvt_pk_bf16_f32)
  const bf16x2v h = __builtin_convertvector((f32x2v){a, b}, bf16x2v);
  return __builtin_bit_cast(uint32_t, h);
}
// eight fp32 -> hi plane and mid plane (x - hi, rounded again), 8 bf16 each
__device__ __forceinline__ void split2x8(const float (&v)[8], uint4& hi, uint4& mid) {
  uint32_t h[4], m[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = pk_bf16(v[2 * i], v[2 * i + 1]);
    const float ra = v[2 * i] - __uint_as_float(h[i] << 16), rb = v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u);
    m[i] = pk_bf16(ra, rb);
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  mid = make_uint4(m[0], m[1], m[2], m[3]);
}

// four waves per SIMD (<= 128 VGPRs; the kernels needed 112-134: no spill at 4, 100-220 bytes of scratch at 5).  PMC said the waves
// of these kernels are parked at s_waitcnt / s_barrier half of their cycles; a fourth resident wave covers more of that than deeper
// software pipelining did (A/B as separate builds, one box: vocoder batch 32 31.0 -> 29.8 ms, 8 prompt encodes 27.6 -> 27.2 ms).
#ifndef SMI_CB_OCC
#define SMI_CB_OCC 4
#endif
template <int QB, bool KS, int CHG, int NC, bool WPF = false, bool WALL = false>
__global__ __launch_bounds__(256, WALL ? 2 : SMI_CB_OCC) void k_convb(ConvP p) {
  static_assert(!KS || CHG % 2 == 0, "channel-split waves own whole 16-channel steps");
  static_assert(!WPF || (KS && CHG == 4), "WPF: one-tap layers with the channels split over the waves, 128-channel chunks");
  static_assert(!WALL || (KS && CHG == 2 && !WPF), "WALL: several taps, the channels split over the waves, one 16-channel step per wave and chunk");
  constexpr int kCh = 32 * CHG;        // channels per staged chunk (four waves x CHG octets)
  constexpr int NOCT = 4 * CHG;        // octets per chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];
  uint4* lds16 = (uint4*)lds;          // [2 planes][NOCT][xw]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int QT = QB * 32;
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int b = bz / p.S, phase = bz - b * p.S;
  const int q0 = bx * QT;
  const int len = p.lens[b];
  const int olen = p.olens ? p.olens[b] : len;
  if (q0 >= olen) return;
  const int ct = KS ? by : by * 4 + wave;
  const bool live = ct * 32 < p.Cout;
  const int ntap = p.ntaps[phase];
  const int ksteps = (p.Cin + 15) >> 4;                 // 16-channel steps of the packed weights
  const float* Xb = p.X + (long long)b * p.xb;
  const uint4* Wp = (const uint4*)(p.W + p.wphase[phase]) + (long long)ct * ntap * ksteps * 128;

  f32x16 acc[QB];
#pragma unroll
  for (int i = 0; i < QB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int xw = p.xw;
  constexpr int RW = 8 * CHG;   // rows staged per wave = CHG octets
  float sreg[RW][NC];
  auto stage_load = [&](int c0) {
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int ci = c0 + wave * RW + r;
      const float* xr = Xb + (long long)ci * p.xstride;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const int col = lane + 64 * k, t = q0 - p.halo_l + col;
        const bool ok = ci < p.Cin && col < xw && t >= 0 && t < len;
        sreg[r][k] = ok ? xr[t] : 0.f;
      }
    }
  };
  auto stage_store = [&]() {   // split + pack: one 16-byte piece per (octet, column, plane)
#pragma unroll
    for (int g = 0; g < CHG; ++g)
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = sreg[g * 8 + r][k];
        uint4 hi, mid;
        split2x8(v, hi, mid);
        if (lane + 64 * k < xw) {
          lds16[(size_t)(wave * CHG + g) * xw + lane + 64 * k] = hi;
          lds16[(size_t)(NOCT + wave * CHG + g) * xw + lane + 64 * k] = mid;
        }
      }
  };
  if constexpr (WPF) {
    // One-tap layers with the channels split over the waves, launched with so few blocks that a SIMD holds one or two waves
    // (one utterance: the prenet's point-wise convs, 80 blocks of 12 chunks at 150 frames).  A wave's work per chunk is 3 * QB * 2
    // MFMAs, so the loop runs at the pace of its memory round trips: with the weights requested one step ahead INSIDE a chunk
    // there were three of them per chunk (staged rows, first step's weights, second step's weights: ~3 us per chunk, 35 us per
    // layer).  Here the chunk's weights -- two steps x two planes, 16 registers -- are requested one chunk ahead together with
    // its staged rows: one round trip per chunk, hidden behind the previous chunk.  Same steps in the same order: same bits.
    if constexpr (QB == 1) {
      // One 32-column tile (short sequences): lanes 32-63 would idle in a row load, so a load instruction takes TWO rows (lanes 0-31 row j,
      // lanes 32-63 row 16 + j of the wave's 32): 16 + 4 loads per chunk -- and with 20 instead of 36 loads per chunk TWO chunks fit the
      // 6-bit vmcnt: two register sets, A and B, used in turn, every request unconditional (past the end: clamped addresses, masked
      // values, skipped steps) so that hipcc can count what is in flight: a chunk's round trip hides behind the chunk before it AND the
      // one being multiplied.  (With 36 loads per set the counter overflowed and the waits drained to vmcnt(0): measured, no gain.)
      const int rh = lane >> 5, colq = lane & 31;
      auto loadx = [&](float (&sr)[16], uint4 (&wq)[4], int c0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int ci = c0 + wave * RW + rh * 16 + j;
          const float* xr = Xb + (long long)(ci < p.Cin ? ci : p.Cin - 1) * p.xstride;
          const int t = q0 - p.halo_l + colq;
          sr[j] = xr[t < 0 ? 0 : (t < len ? t : len - 1)];
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          int st = (c0 >> 4) + wave * 2 + g;
          st = st < ksteps ? st : ksteps - 1;
          wq[2 * g] = Wp[(long long)st * 128 + lane];          // (channel-split mode: every block's output tile is live)
          wq[2 * g + 1] = Wp[(long long)st * 128 + 64 + lane];
        }
      };
      const int col0 = p.halo_l + p.off[phase][0] + (lane & 31);
      auto chunk = [&](float (&sr)[16], uint4 (&wq)[4], int c0, bool first) {
        if (!first) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous chunk's LDS reads are done
        const int t = q0 - p.halo_l + colq;
        const bool okc = colq < xw && t >= 0 && t < len;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = (okc && c0 + wave * RW + rh * 16 + o * 8 + r < p.Cin) ? sr[o * 8 + r] : 0.f;
          uint4 hi, mid;
          split2x8(v, hi, mid);
          if (colq < xw) {
            lds16[(size_t)(wave * CHG + 2 * rh + o) * xw + colq] = hi;
            lds16[(size_t)(NOCT + wave * CHG + 2 * rh + o) * xw + colq] = mid;
          }
        }
        uint4 wc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wc[i] = wq[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        loadx(sr, wq, c0 + 2 * kCh);                                       // this set's next chunk: two chunks ahead, unconditionally
        if (live) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            if ((c0 >> 4) + wave * 2 + g < ksteps) {   // wave-uniform
              const uint4* bp = lds16 + (size_t)(2 * (wave * 2 + g) + (lane >> 5)) * xw + col0;
              const bf16x8 ah = __builtin_bit_cast(bf16x8, wc[2 * g]), am = __builtin_bit_cast(bf16x8, wc[2 * g + 1]);
              const bf16x8 bh = __builtin_bit_cast(bf16x8, bp[0]);
              const bf16x8 bm = __builtin_bit_cast(bf16x8, bp[(size_t)NOCT * xw]);
              acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[0], 0, 0, 0);
              acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[0], 0, 0, 0);
              acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[0], 0, 0, 0);
            }
          }
        }
      };
      float sa[16], sb[16];
      uint4 wa[4], wb[4];
      loadx(sa, wa, 0);
      loadx(sb, wb, kCh);
      for (int c0 = 0; c0 < p.CinP; c0 += 2 * kCh) {   // chunks in pairs (an odd count ends with a chunk of masked rows and skipped steps)
        chunk(sa, wa, c0, c0 == 0);
        chunk(sb, wb, c0 + kCh, false);
      }
      conv_finish<QB, KS>(p, acc, lds, ct, live, b, phase, q0, olen, lane, wave);
      return;
    }
    uint4 wn[4];
    const int ksl = ksteps;   // (steps past the last one multiply zero rows: their weights are clamped loads, never used)
    auto wload = [&](int c0) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        int st = (c0 >> 4) + wave * 2 + g;
        st = st < ksl ? st : ksl - 1;
        wn[2 * g] = live ? Wp[(long long)st * 128 + lane] : make_uint4(0u, 0u, 0u, 0u);
        wn[2 * g + 1] = live ? Wp[(long long)st * 128 + 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
      }
    };
    stage_load(0);
    wload(0);
    const int col0 = p.halo_l + p.off[phase][0] + (lane & 31);
    for (int c0 = 0; c0 < p.CinP; c0 += kCh) {
      if (c0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stage_store();
      uint4 wc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wc[i] = wn[i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (c0 + kCh < p.CinP) { stage_load(c0 + kCh); wload(c0 + kCh); }
      if (live) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          if ((c0 >> 4) + wave * 2 + g < ksl) {   // wave-uniform
            const uint4* bp = lds16 + (size_t)(2 * (wave * 2 + g) + (lane >> 5)) * xw + col0;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, wc[2 * g]), am = __builtin_bit_cast(bf16x8, wc[2 * g + 1]);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
              const bf16x8 bh = __builtin_bit_cast(bf16x8, bp[qb * 32]);
              const bf16x8 bm = __builtin_bit_cast(bf16x8, bp[(size_t)NOCT * xw + qb * 32]);
              acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[qb], 0, 0, 0);
              acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[qb], 0, 0, 0);
              acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[qb], 0, 0, 0);
            }
          }
        }
      }
    }
    conv_finish<QB, KS>(p, acc, lds, ct, live, b, phase, q0, olen, lane, wave);
    return;
  }
  if constexpr (WALL) {
    // Several taps, the channels split over the waves, launched with about a block per CU (one utterance: the 7-tap convs at C = 768 /
    // 384, conv_in, the prenet's embed convs, the transposed convs of the first two decoder blocks).  A wave's chunk is ONE 16-channel
    // step x taps; with the next tap's weights requested one step ahead the taps were a chain of dependent round trips to cold
    // weights -- 7 x ~0.8 us per chunk, 72 us for the 7-tap conv at C = 768 where its MFMAs need 7.  Here ALL taps' weight planes of the
    // chunk (<= 8 x 2 KiB per wave: 64 registers, hence two waves per SIMD) are requested together at the top of the chunk, behind the
    // staged rows requested one chunk earlier: one round trip per chunk.  Same taps in the same order: same bits.
    stage_load(0);
    for (int c0 = 0; c0 < p.CinP; c0 += kCh) {
      const int stp = (c0 >> 4) + wave;                 // this wave's 16-channel step of the chunk
      const bool has = live && stp < ksteps;             // wave-uniform
      uint4 wt[kMaxTaps][2];
      {
        const uint4* Wq = Wp + (long long)(stp < ksteps ? stp : ksteps - 1) * 128 + lane;
#pragma unroll
        for (int t = 0; t < kMaxTaps; ++t) {
          const int tc = t < ntap ? t : ntap - 1;
          wt[t][0] = live ? Wq[(long long)tc * ksteps * 128] : make_uint4(0u, 0u, 0u, 0u);
          wt[t][1] = live ? Wq[(long long)tc * ksteps * 128 + 64] : make_uint4(0u, 0u, 0u, 0u);
        }
      }
      if (c0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stage_store();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (c0 + kCh < p.CinP) stage_load(c0 + kCh);
      if (has) {
#pragma unroll
        for (int t = 0; t < kMaxTaps; ++t) {
          if (t < ntap) {   // uniform
            const int col0 = p.halo_l + p.off[phase][t] + (lane & 31);
            const uint4* bp = lds16 + (size_t)(2 * wave + (lane >> 5)) * xw + col0;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, wt[t][0]), am = __builtin_bit_cast(bf16x8, wt[t][1]);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
              const bf16x8 bh = __builtin_bit_cast(bf16x8, bp[qb * 32]);
              const bf16x8 bm = __builtin_bit_cast(bf16x8, bp[(size_t)NOCT * xw + qb * 32]);
              acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[qb], 0, 0, 0);
              acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[qb], 0, 0, 0);
              acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[qb], 0, 0, 0);
            }
          }
        }
      }
    }
    conv_finish<QB, KS>(p, acc, lds, ct, live, b, phase, q0, olen, lane, wave);
    return;
  }
  stage_load(0);
  for (int c0 = 0; c0 < p.CinP; c0 += kCh) {
    if (c0) { if (KS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else __syncthreads(); }   // previous chunk fully read
    stage_store();
    if (KS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else __syncthreads();
    if (c0 + kCh < p.CinP) stage_load(c0 + kCh);
    if (live) {
      // (tap, 16-channel step) steps in order over this wave's steps of the chunk (all 2 CHG of them, or with KS its own
      // CHG / 2); the NEXT step's two weight planes are requested before this step's MFMAs and fenced there
      const int sbase = KS ? wave * (CHG / 2) : 0, scap = KS ? CHG / 2 : 2 * CHG;
      int sl = ksteps - (c0 >> 4) - sbase;            // live steps of this chunk from sbase on
      sl = sl < scap ? sl : scap;
      const int nstep = sl > 0 ? ntap * sl : 0;
      const uint4* Wq = Wp + ((long long)(c0 >> 4) + sbase) * 128 + lane;
      // Two weight sets, A and B, used in turn (the loop advances two steps per trip): written as ONE set with `w = w_next` at the end
      // of a step, hipcc copies the registers there and waits vmcnt(0) for the just-requested next weights in the MIDDLE of the
      // current step's MFMAs (round-4 ISA of the 7-tap conv: every step exposed a whole L2 round trip; the prefetch hid nothing).
      // A set that is loaded straight into its own registers after its last use needs no copy, and the wait in front of a step's
      // MFMAs leaves the other set's request in flight.
      auto advance = [&](int& t, int& gg) { if (++gg == sl) { gg = 0; ++t; } };
      auto wptr = [&](int t, int gg) { return Wq + ((long long)(t < ntap ? t : ntap - 1) * ksteps + gg) * 128; };   // (past the end: a valid address, never used)
      auto step = [&](const uint4& wh, const uint4& wm, int t, int gg) {
        const int col0 = p.halo_l + p.off[phase][t] + (lane & 31);
        const uint4* bp = lds16 + (size_t)(2 * (sbase + gg) + (lane >> 5)) * xw + col0;
        const bf16x8 ah = __builtin_bit_cast(bf16x8, wh), am = __builtin_bit_cast(bf16x8, wm);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const bf16x8 bh = __builtin_bit_cast(bf16x8, bp[qb * 32]);
          const bf16x8 bm = __builtin_bit_cast(bf16x8, bp[(size_t)NOCT * xw + qb * 32]);
          acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[qb], 0, 0, 0);
          acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[qb], 0, 0, 0);
          acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[qb], 0, 0, 0);
        }
      };
      if (nstep > 0) {
        int ta = 0, ga = 0;                      // set A's step
        uint4 wah = Wq[0], wam = Wq[64];
        for (int st = 0; st < nstep; st += 2) {
          int tb = ta, gb = ga;
          advance(tb, gb);                       // set B's step = A's successor
          const uint4* pb = wptr(tb, gb);
          const uint4 wbh = pb[0], wbm = pb[64];
          __builtin_amdgcn_sched_barrier(0);
          step(wah, wam, ta, ga);
          ta = tb; ga = gb;
          advance(ta, ga);                       // A's next step = B's successor
          const uint4* pa = wptr(ta, ga);
          wah = pa[0]; wam = pa[64];
          __builtin_amdgcn_sched_barrier(0);
          if (st + 1 < nstep) step(wbh, wbm, tb, gb);
        }
      }
    }
  }
  conv_finish<QB, KS>(p, acc, lds, ct, live, b, phase, q0, olen, lane, wave);
}

// k_convbT (round 4): ConvTranspose1d on the bf16-split pipe with PH output PHASES per block.  As a polyphase conv on k_convb
// every phase r is its own set of blocks (grid.z = B * S): the input tile is staged S times, and -- what costs -- output element
// (q, r) lives at t = q * S + r, so a phase's stores put 4 bytes per lane at a stride of 4 S bytes: 64 store instructions per wave
// (16 rows x 2 column tiles x 2 outputs), each spread over 64 different 32-byte sectors -- about half of the kernel's time
// (the transposed convs ran at 118-154 TFLOP/s fp32-equivalent beside 280-320 for the 7-tap convs of the same shapes).  Here a
// block keeps PH accumulator sets (PH * 16 registers) for ONE 32-column tile: the chunk is staged once for the PH phases -- 32 + halo
// <= 64 columns, one 64-lane pass per row -- and in the epilogue a lane holds the PH consecutive outputs t = q S + r0 .. + PH - 1 of
// each of its rows: one 16-byte store (PH = 4) or PH dword stores per row.  Measured (profiles/r04_convT_multi_phase.txt): what pays
// is the STAGING -- a 2-tap phase carries 24 MFMAs per 32 staged channels against ~150 staging instructions (loads, the fp32 -> 2 x
// bf16 split, LDS writes), PH phases share them; the stores alone (first build: two passes per row as in k_convb) changed nothing.
// A phase's taps and 16-channel steps run in k_convb's order (64-channel chunks), one accumulator per phase: the bits of k_convb
// with CHG = 2.  Epilogue: bias, raw output and / or the next layer's Snake -- what the WaveGenerator's transposed convs need (the
// launch builder checks that nothing else is asked for).
template <int PH>
__global__ __launch_bounds__(256, 3) void k_convbT(ConvP p) {
  constexpr int QB = 1, NC = 1, CHG = 2, NOCT = 4 * CHG, kCh = 32 * CHG, RW = 8 * CHG;   // 32 + halo <= 64 staged columns: ONE 64-lane pass per row
  extern __shared__ __attribute__((aligned(16))) float lds[];
  uint4* lds16 = (uint4*)lds;          // [2 planes][NOCT][xw]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int QT = QB * 32;
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int ngrp = p.S / PH;
  const int b = bz / ngrp, phase0 = (bz - b * ngrp) * PH;
  const int q0 = bx * QT;
  const int len = p.lens[b];
  if (q0 >= len) return;
  const int ct = by * 4 + wave;
  const bool live = ct * 32 < p.Cout;
  const int ksteps = (p.Cin + 15) >> 4;
  const float* Xb = p.X + (long long)b * p.xb;
  f32x16 acc[PH][QB];
#pragma unroll
  for (int h = 0; h < PH; ++h)
#pragma unroll
    for (int i = 0; i < QB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][i][r] = 0.f;
  const int xw = p.xw;
  float sreg[RW][NC];
  auto stage_load = [&](int c0) {
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int ci = c0 + wave * RW + r;
      const float* xr = Xb + (long long)(ci < p.Cin ? ci : p.Cin - 1) * p.xstride;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const int col = lane + 64 * k, t = q0 - p.halo_l + col;
        const bool ok = ci < p.Cin && col < xw && t >= 0 && t < len;
        const float v = xr[t < 0 ? 0 : (t < len ? t : len - 1)];   // unconditional load from a clamped address, masked by value
        sreg[r][k] = ok ? v : 0.f;
      }
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int g = 0; g < CHG; ++g)
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = sreg[g * 8 + r][k];
        uint4 hi, mid;
        split2x8(v, hi, mid);
        if (lane + 64 * k < xw) {
          lds16[(size_t)(wave * CHG + g) * xw + lane + 64 * k] = hi;
          lds16[(size_t)(NOCT + wave * CHG + g) * xw + lane + 64 * k] = mid;
        }
      }
  };
  // weights of (phase, tap, step): two 1-KiB planes at wbase(phase) + ((tap * ksteps + step) * 128 + lane [+ 64])
  auto wbase = [&](int phase) { return (const uint4*)(p.W + p.wphase[phase]) + (long long)ct * p.ntaps[phase] * ksteps * 128 + lane; };
  stage_load(0);
  for (int c0 = 0; c0 < p.CinP; c0 += kCh) {
    if (c0) __syncthreads();   // previous chunk fully read
    stage_store();
    __syncthreads();
    if (c0 + kCh < p.CinP) stage_load(c0 + kCh);
    if (live) {
      int sl = ksteps - (c0 >> 4);
      sl = sl < 2 * CHG ? sl : 2 * CHG;            // live 16-channel steps of this chunk (>= 1)
      const long long so = (long long)(c0 >> 4) * 128;
      // two weight sets used in turn, as in k_convb (a single set rotated with `w = w_next` makes hipcc wait for the just-requested
      // weights in the middle of the current step); a phase's last step requests the NEXT phase's first weights
      uint4 wah, wam;
      { const uint4* w0 = wbase(phase0) + so; wah = w0[0]; wam = w0[64]; }
#pragma unroll
      for (int h = 0; h < PH; ++h) {
        const int phase = phase0 + h, ntap = p.ntaps[phase];
        const uint4* Wq = wbase(phase) + so;
        const uint4* Wnext = wbase(h + 1 < PH ? phase + 1 : phase) + so;   // the next phase's first step (last phase: any valid address)
        const int nstep = ntap * sl;
        auto advance = [&](int& t, int& gg) { if (++gg == sl) { gg = 0; ++t; } };
        auto wptr = [&](int t, int gg) { return t >= ntap ? Wnext : Wq + ((long long)t * ksteps + gg) * 128; };
        auto step = [&](const uint4& wh, const uint4& wm, int t, int gg) {
          const int col0 = p.halo_l + p.off[phase][t] + (lane & 31);
          const uint4* bp = lds16 + (size_t)(2 * gg + (lane >> 5)) * xw + col0;
          const bf16x8 ah = __builtin_bit_cast(bf16x8, wh), am = __builtin_bit_cast(bf16x8, wm);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, bp[qb * 32]);
            const bf16x8 bm = __builtin_bit_cast(bf16x8, bp[(size_t)NOCT * xw + qb * 32]);
            acc[h][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[h][qb], 0, 0, 0);
            acc[h][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[h][qb], 0, 0, 0);
            acc[h][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[h][qb], 0, 0, 0);
          }
        };
        int ta = 0, ga = 0;
        for (int st = 0; st < nstep; st += 2) {
          int tb = ta, gb = ga;
          advance(tb, gb);
          const uint4* pb = wptr(tb, gb);                 // (an odd step count: the phase's successor, i.e. the next phase's first weights)
          uint4 wbh = pb[0], wbm = pb[64];
          __builtin_amdgcn_sched_barrier(0);
          step(wah, wam, ta, ga);
          if (st + 1 < nstep) {
            ta = tb; ga = gb;
            advance(ta, ga);
            const uint4* pa = wptr(ta, ga);
            wah = pa[0]; wam = pa[64];
            __builtin_amdgcn_sched_barrier(0);
            step(wbh, wbm, tb, gb);
          } else {
            wah = wbh; wam = wbm;                          // odd count: B already holds the next phase's first weights
          }
        }
      }
    }
  }
  if (!live) return;
  // epilogue: register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31 (the 32x32 MFMA D layout)
  const long long yboff = (long long)b * p.yb;
  const bool hb = p.bias != nullptr, hs = p.Ys != nullptr, hy = p.Y != nullptr, fsin = p.fast_sin != 0;
  const float* dummy = p.X;
#pragma unroll
  for (int r0 = 0; r0 < 16; r0 += 8) {
    float bv[8], av[8];
    int co[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + i;
      co[i] = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int cc = co[i] < p.Cout ? co[i] : p.Cout - 1;
      bv[i] = (hb ? p.bias : dummy)[cc];
      av[i] = (hs ? p.alpha : dummy)[cc];
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int q = q0 + qb * 32 + (lane & 31);
      if (q >= len) continue;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (co[i] >= p.Cout) continue;
        float y[PH], ys[PH];
#pragma unroll
        for (int h = 0; h < PH; ++h) {
          y[h] = acc[h][qb][r0 + i];
          y[h] = hb ? y[h] + bv[i] : y[h];
          ys[h] = snake_f(y[h], av[i], fsin);
        }
        const long long o = yboff + (long long)co[i] * p.ystride + (long long)q * p.S + phase0;
        if constexpr (PH == 4) {
          if (hy) *(float4*)(p.Y + o) = make_float4(y[0], y[1], y[2], y[3]);
          if (hs) *(float4*)(p.Ys + o) = make_float4(ys[0], ys[1], ys[2], ys[3]);
        } else {
#pragma unroll
          for (int h = 0; h < PH; ++h) {
            if (hy) p.Y[o + h] = y[h];
            if (hs) p.Ys[o + h] = ys[h];
          }
        }
      }
    }
  }
}

// k_resunit (round 4): a whole ResidualUnit (blocks/layers.py:51-67) in ONE launch where a block can hold every channel of its
// time tile -- C = 96 (three waves) and C = 192 (six waves), the two resolutions at which the unit is bound by memory, not by the
// matrix pipe: y = x + conv1(snake(conv7_dil(snake(x)))).  As two launches the 7-tap conv writes A = snake(conv7 + b) to HBM and the
// 1x1 conv reads it back, stages it and splits it again: six passes over a [C][T] tensor per unit (Xs in, A out | A in, residual in,
// raw out, Snake'd out).  Here wave w owns output tile w in BOTH convs: phase 1 is k_convb's 7-tap loop (48-channel chunks: 8 rows per
// wave x 6 waves, or 16 x 3); its accumulators -- register r of lane l = channel 32 w + (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31:
// four consecutive channels per lane -- get bias and Snake, are split into the two bf16 planes and go straight into the LDS image of
// the 1x1 conv's B operand ([plane][octet][column][8 bf16], over the dead staging rows); phase 2 multiplies W1 (2 KiB per step from
// L2) against that image, adds bias and the residual and writes the unit's two outputs.  Four passes per unit, no second staging.
// Same products and the same (tap, step) order per chunk as k_convb; the 7-tap sum runs over 48- instead of 32-channel chunks.
struct ResP {
  const float* Xs;      // snake(U, alpha0): [B][C][stride]
  const float* U;       // residual (raw), same layout
  const float* W7;      // conv7 weights, bf16 planes (PACK_CONV_B)
  const float* b7;
  const float* alpha2;  // Snake between the convs
  const float* W1;      // 1x1 weights, bf16 planes
  const float* b1;
  const float* alpha_next;   // Snake of the NEXT consumer (Ys) or null
  float* Y;             // raw output (may alias U) or null
  float* Ys;            // snake(Y, alpha_next) or null
  const int* lens;
  int C, stride;
  long long bs;
  int halo_l, xw;
  int off[kMaxTaps];
  int fast_sin;
};

template <int NWV>
__global__ __launch_bounds__(NWV * 64) void k_resunit(ResP p) {
  constexpr int QB = 2, NC = 2, RW = 48 / NWV, OPW = RW / 8;   // 64 columns per block; rows / octets staged per wave per 48-channel chunk
  static_assert(NWV == 3 || NWV == 6, "C = 96 or 192");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  uint4* lds16 = (uint4*)lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, q0 = blockIdx.x * (QB * 32);
  const int len = p.lens[b];
  if (q0 >= len) return;
  const int C = p.C, ksteps = C >> 4, xw = p.xw;
  const float* Xb = p.Xs + (long long)b * p.bs;
  f32x16 acc[QB];
#pragma unroll
  for (int i = 0; i < QB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float sreg[RW][NC];
  auto stage_load = [&](int c0) {
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const float* xr = Xb + (long long)(c0 + wave * RW + r) * p.stride;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const int col = lane + 64 * k, t = q0 - p.halo_l + col;
        const bool ok = col < xw && t >= 0 && t < len;
        const float v = xr[t < 0 ? 0 : (t < len ? t : len - 1)];
        sreg[r][k] = ok ? v : 0.f;
      }
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int g = 0; g < OPW; ++g)
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = sreg[g * 8 + r][k];
        uint4 hi, mid;
        split2x8(v, hi, mid);
        if (lane + 64 * k < xw) {
          lds16[(size_t)(wave * OPW + g) * xw + lane + 64 * k] = hi;
          lds16[(size_t)(6 + wave * OPW + g) * xw + lane + 64 * k] = mid;
        }
      }
  };
  // ---- phase 1: the dilated 7-tap conv, output tile `wave`
  {
    const uint4* Wp = (const uint4*)p.W7 + (long long)wave * 7 * ksteps * 128 + lane;
    stage_load(0);
    for (int c0 = 0; c0 < C; c0 += 48) {
      if (c0) __syncthreads();
      stage_store();
      __syncthreads();
      if (c0 + 48 < C) stage_load(c0 + 48);
      const uint4* Wq = Wp + (long long)(c0 >> 4) * 128;
      // 7 taps x 3 steps, two weight sets used in turn (k_convb's step loop: no register rotation, no wait for the set just requested)
      auto advance = [&](int& t, int& gg) { if (++gg == 3) { gg = 0; ++t; } };
      auto wptr = [&](int t, int gg) { return Wq + ((long long)(t < 7 ? t : 6) * ksteps + gg) * 128; };
      auto step = [&](const uint4& wh, const uint4& wm, int t, int gg) {
        const uint4* bp = lds16 + (size_t)(2 * gg + (lane >> 5)) * xw + p.halo_l + p.off[t] + (lane & 31);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, wh), am = __builtin_bit_cast(bf16x8, wm);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const bf16x8 bh = __builtin_bit_cast(bf16x8, bp[qb * 32]);
          const bf16x8 bm = __builtin_bit_cast(bf16x8, bp[(size_t)6 * xw + qb * 32]);
          acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[qb], 0, 0, 0);
          acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[qb], 0, 0, 0);
          acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[qb], 0, 0, 0);
        }
      };
      int ta = 0, ga = 0;
      uint4 wah = Wq[0], wam = Wq[64];
      for (int st = 0; st < 7 * 3; st += 2) {
        int tb = ta, gb = ga;
        advance(tb, gb);
        const uint4* pb = wptr(tb, gb);
        const uint4 wbh = pb[0], wbm = pb[64];
        __builtin_amdgcn_sched_barrier(0);
        step(wah, wam, ta, ga);
        ta = tb; ga = gb;
        advance(ta, ga);
        const uint4* pa = wptr(ta, ga);
        wah = pa[0]; wam = pa[64];
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < 7 * 3) step(wbh, wbm, tb, gb);
      }
    }
  }
  // ---- A = snake(conv7 + b7, alpha2) -> the 1x1 conv's B operand in LDS: [plane][octet = channel / 8][64 columns][8 bf16]
  const int noct = C >> 3;
  const uint4* W1p = (const uint4*)p.W1 + (long long)wave * ksteps * 128 + lane;
  uint4 wh = W1p[0], wm = W1p[64];         // the first step's 1x1 weights travel under the Snake arithmetic
  __syncthreads();                          // every wave has read the last staged chunk: the image may overwrite it
  {
    const bool fsin = p.fast_sin != 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float bv[4], av[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int c = wave * 32 + 8 * k + 4 * (lane >> 5) + e; bv[e] = p.b7[c]; av[e] = p.alpha2[c]; }
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = snake_f(acc[qb][4 * k + e] + bv[e], av[e], fsin);
        const uint32_t h0 = pk_bf16(a[0], a[1]), h1 = pk_bf16(a[2], a[3]);
        const uint32_t m0 = pk_bf16(a[0] - __uint_as_float(h0 << 16), a[1] - __uint_as_float(h0 & 0xffff0000u));
        const uint32_t m1 = pk_bf16(a[2] - __uint_as_float(h1 << 16), a[3] - __uint_as_float(h1 & 0xffff0000u));
        unsigned char* q = (unsigned char*)lds + ((size_t)(wave * 4 + k) * 64 + qb * 32 + (lane & 31)) * 16 + 8 * (lane >> 5);
        *(uint2*)q = make_uint2(h0, h1);
        *(uint2*)(q + (size_t)noct * 64 * 16) = make_uint2(m0, m1);
      }
    }
  }
  __syncthreads();
  // ---- phase 2: the 1x1 conv on the image, output tile `wave`
#pragma unroll
  for (int i = 0; i < QB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  {
    auto step2 = [&](const uint4& xh, const uint4& xm, int st) {
      const uint4* bp = lds16 + (size_t)(2 * st + (lane >> 5)) * 64 + (lane & 31);
      const bf16x8 ah = __builtin_bit_cast(bf16x8, xh), am = __builtin_bit_cast(bf16x8, xm);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bp[qb * 32]);
        const bf16x8 bm = __builtin_bit_cast(bf16x8, bp[(size_t)noct * 64 + qb * 32]);
        acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[qb], 0, 0, 0);
        acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[qb], 0, 0, 0);
        acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[qb], 0, 0, 0);
      }
    };
    for (int st = 0; st < ksteps; st += 2) {      // ksteps = C / 16 = 6 or 12: even; (wh, wm) is set A, already requested
      const uint4 wbh = W1p[(long long)(st + 1) * 128], wbm = W1p[(long long)(st + 1) * 128 + 64];
      __builtin_amdgcn_sched_barrier(0);
      step2(wh, wm, st);
      const int sn = st + 2 < ksteps ? st + 2 : st;
      wh = W1p[(long long)sn * 128]; wm = W1p[(long long)sn * 128 + 64];
      __builtin_amdgcn_sched_barrier(0);
      step2(wbh, wbm, st + 1);
    }
  }
  // ---- epilogue: + b1 + residual; raw and Snake'd outputs (all loads of eight rows first, as in conv_finish)
  const long long boff = (long long)b * p.bs;
  const bool hs = p.Ys != nullptr, hy = p.Y != nullptr, fsin = p.fast_sin != 0;
  bool qok[QB];
  int tq[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) { const int q = q0 + qb * 32 + (lane & 31); qok[qb] = q < len; tq[qb] = qok[qb] ? q : len - 1; }
#pragma unroll
  for (int r0 = 0; r0 < 16; r0 += 8) {
    float bv[8], av[8], rv[QB][8];
    int co[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + i;
      co[i] = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      bv[i] = p.b1[co[i]];
      av[i] = (hs ? p.alpha_next : p.b1)[co[i]];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) rv[qb][i] = p.U[boff + (long long)co[i] * p.stride + tq[qb]];
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float y = acc[qb][r0 + i];
        y = y + bv[i];
        y = rv[qb][i] + y;
        if (qok[qb]) {
          const long long o = boff + (long long)co[i] * p.stride + tq[qb];
          if (hy) p.Y[o] = y;
          if (hs) p.Ys[o] = snake_f(y, av[i], fsin);
        }
      }
  }
}

// Conv1d with ONE output channel (the vocoder's last layer: C -> 1, 7 taps, tanh): y[b][q] = act(bias + sum_ci sum_tap W[ci][tap]
// x[b][ci][q + off[tap]]).  On the MFMA kernels 31 of a tile's 32 output rows are padding and the layer -- 393 MB of
// activations at batch 32 -- ran at 0.3 TB/s (1.3 ms); here a thread owns one output sample, rows are read coalesced along time
// (a sample is re-read by its 7 taps: L1 hits), weights come from LDS (unpacked from the conv layout: lane l, slot j of group g
// holds W[l & 31][8g + 2j + (l >> 5)]), every load is unconditional (clamped index, masked value).
template <int NTAP>
__global__ __launch_bounds__(256) void k_conv_c1(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];      // [Cin][NTAP]
  const int tid = threadIdx.x, b = blockIdx.y, q = blockIdx.x * 256 + tid;
  const int len = p.lens[b], olen = p.olens ? p.olens[b] : len;
  if ((int)blockIdx.x * 256 >= olen) return;
  const int groups = p.CinP >> 3;
  const float* Wf = p.W + p.wphase[0];
  for (int i = tid; i < p.Cin * NTAP; i += 256) {
    const int ci = i / NTAP, tap = i - ci * NTAP;
    lds[i] = Wf[((size_t)(tap * groups + (ci >> 3)) * 64 + (ci & 1) * 32) * 4 + ((ci & 7) >> 1)];
  }
  __syncthreads();
  if (q >= olen) return;
  int tt[NTAP];
  bool ok[NTAP];
#pragma unroll
  for (int k = 0; k < NTAP; ++k) {
    const int t = q + p.off[0][k];
    ok[k] = t >= 0 && t < len;
    tt[k] = t < 0 ? 0 : (t < len ? t : len - 1);
  }
  const float* Xb = p.X + (long long)b * p.xb;
  float acc = 0.f;
  for (int c0 = 0; c0 < p.Cin; c0 += 4) {
    float v[4][NTAP];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ci = c0 + u < p.Cin ? c0 + u : p.Cin - 1;
      const float* xr = Xb + (long long)ci * p.xstride;
#pragma unroll
      for (int k = 0; k < NTAP; ++k) v[u][k] = xr[tt[k]];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c0 + u < p.Cin) {
#pragma unroll
        for (int k = 0; k < NTAP; ++k) acc = fmaf(lds[(c0 + u) * NTAP + k], ok[k] ? v[u][k] : 0.f, acc);
      }
    }
  }
  float y = acc;
  if (p.bias) y += p.bias[0];
  if (p.act == ACT_GELU) y = gelu_f(y);
  if (p.act == ACT_RELU) y = fmaxf(y, 0.f);
  if (p.act == ACT_TANH) y = tanhf(y);
  p.Y[(long long)b * p.yb + q] = y;
}

// Linear layer on one vector per utterance (d-vector projection, AdaLN parameters): y[b][co] =
// bias[co] + sum_ci W[co][ci] x[b][ci], fp32, reading the same packed conv weights (lane l, slot j of
// group g holds W[32*ct + (l&31)][8g + 2j + (l>>5)]).  One block per 32 outputs, waves split K.
__global__ __launch_bounds__(256) void k_gemv1(ConvP p) {
  __shared__ float part[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ct = blockIdx.x, b = blockIdx.y;
  const int groups = p.CinP >> 3;
  const float4* Wp = (const float4*)p.W + (long long)ct * groups * 64;
  const float* x = p.X + (long long)b * p.xb;           // xstride == 1: x[ci]
  float acc = 0.f;
  for (int g = wave; g < groups; g += 4) {
    const float4 w = Wp[(long long)g * 64 + lane];
    const int ci = g * 8 + (lane >> 5);
    const float x0 = ci < p.Cin ? x[ci] : 0.f, x1 = ci + 2 < p.Cin ? x[ci + 2] : 0.f;
    const float x2 = ci + 4 < p.Cin ? x[ci + 4] : 0.f, x3 = ci + 6 < p.Cin ? x[ci + 6] : 0.f;
    acc += (w.x * x0 + w.y * x1) + (w.z * x2 + w.w * x3);
  }
  acc += __shfl_xor(acc, 32, 64);
  if (lane < 32) part[wave][lane] = acc;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int co = ct * 32 + threadIdx.x;
    if (co < p.Cout) {
      float y = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
      if (p.bias) y += p.bias[co];
      if (p.act == ACT_RELU) y = fmaxf(y, 0.f);
      if (p.act == ACT_SIGMOID) y = 1.0f / (1.0f + expf(-y));
      p.Y[(long long)b * p.yb + co] = y;
    }
  }
}

// depthwise conv7 (optional) + LayerNorm / AdaLayerNorm over channels, eps 1e-6 (vocos.py:65-110)
struct LnP {
  const float* X;      // [B][C][stride]
  float* Y;
  const float* dww;    // [C][7] or null
  const float* dwb;    // [C]
  const float* w;      // LN weight [C] (plain) or null
  const float* bsh;    // LN bias [C]
  const float* ada;    // AdaLN: [B][ada_stride] with scale at +0, shift at +C; or null
  const int* lens;
  int C, stride, ada_stride;
  long long bs;
  int triple;          // 1: write 3x (SamplingBlock ratio 1 after final_layer_norm)
  float eps;           // 1e-6 (vocos), 1e-5 (wav2vec2)
  int gelu;            // 1: GELU(erf) after the affine (wav2vec2 feature-encoder layers)
};

template <int CPT>  // channels per thread (C <= 32*CPT): 8 time steps x 32 channel groups per block
__global__ __launch_bounds__(256) void k_dwln(LnP p) {
  __shared__ float red[32][8];
  const int tt = threadIdx.x & 7, cg = threadIdx.x >> 3;
  const int b = blockIdx.y, t = blockIdx.x * 8 + tt;
  const int len = p.lens[b];
  const bool tv = t < len;
  const float* Xb = p.X + (long long)b * p.bs;
  float v[CPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = cg + 32 * i;
    float x = 0.f;
    if (c < p.C && tv) {
      const float* xr = Xb + (long long)c * p.stride;
      if (p.dww) {
        x = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const int tj = t + j - 3;
          const float xv = (tj >= 0 && tj < len) ? xr[tj] : 0.f;
          x += p.dww[c * 7 + j] * xv;
        }
        x += p.dwb[c];
      } else {
        x = xr[t];
      }
    }
    v[i] = x;
    s += x;
  }
  red[cg][tt] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int g = 0; g < 32; ++g) mean += red[g][tt];
  mean /= (float)p.C;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = cg + 32 * i;
    if (c < p.C) { const float d = v[i] - mean; q += d * d; }
  }
  red[cg][tt] = q;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int g = 0; g < 32; ++g) var += red[g][tt];
  const float rstd = 1.0f / sqrtf(var / (float)p.C + p.eps);
  if (!tv) return;
  float* Yb = p.Y + (long long)b * p.bs;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = cg + 32 * i;
    if (c >= p.C) continue;
    float y = (v[i] - mean) * rstd;
    if (p.ada) y = y * p.ada[(long long)b * p.ada_stride + c] + p.ada[(long long)b * p.ada_stride + p.C + c];
    else y = y * p.w[c] + p.bsh[c];
    if (p.triple) y = (y + y) + y;
    if (p.gelu) y = gelu_f(y);
    Yb[(long long)c * p.stride + t] = y;
  }
}

// codebook lookup: Z[b][d][t] = codebook[sem[b][t]][d]   (factorized_vector_quantize.py:160-167)
__global__ void k_codebook(const int64_t* sem, int semstride, const float* cb, int D, int cbsize,
                           const int* lens, float* Z, int zstride, long long zb) {
  const int b = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lens[b]) return;
  long long id = sem[(long long)b * semstride + t];
  id = id < 0 ? 0 : (id >= cbsize ? cbsize - 1 : id);
  for (int d = 0; d < D; ++d) Z[(long long)b * zb + (long long)d * zstride + t] = cb[id * D + d];
}

// FSQ index -> level codes -> Linear(nd -> latent), written d-major: out[b][d*Ntok + t]
// (finite_scalar_quantization.py:143-162, residual_fsq.py:191-199, speaker_encoder.py:109-110)
struct FsqP {
  const int32_t* glob;  // [B][Ntok]
  const float* W1;      // [latent][nd]
  const float* b1;      // [latent]
  float* out;           // [B][latent*Ntok]
  int Ntok, latent, nd;
  int levels[8];
};
__global__ void k_fsq(FsqP p) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < p.latent * p.Ntok; i += blockDim.x) {
    const int d = i / p.Ntok, t = i - d * p.Ntok;
    int idx = p.glob[b * p.Ntok + t];
    float acc = 0.f;
    int basis = 1;
    for (int j = 0; j < p.nd; ++j) {
      const int L = p.levels[j], half = L / 2;
      const int lvl = (idx / basis) % L;
      const float code = (float)(lvl - half) / (float)half;
      acc += code * p.W1[d * p.nd + j];
      basis *= L;
    }
    p.out[(long long)b * p.latent * p.Ntok + i] = acc + p.b1[d];
  }
}

__global__ void k_zero_tail(float* wav, int stride, const int* lens, int hop) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < stride && t >= lens[b] * hop) wav[(long long)b * stride + t] = 0.f;
}

// ------------------------------------------------------------------------------------------
// arena description
// ------------------------------------------------------------------------------------------
struct Entry {
  std::string name;   // reference state_dict key (after remove_weight_norm); "cat:a|b|c" = rows concatenated
  int kind;           // PACK_*
  int Cout, Cin, K;   // logical dims (K taps)
  int S, pad;         // ConvTranspose1d stride / padding
  size_t offset, bytes;
};

struct ConvGeom {
  int S;
  int ntaps[kMaxPhases];
  int off[kMaxPhases][kMaxTaps];
  int tapk[kMaxPhases][kMaxTaps];   // kernel index j of each (phase, tap)
  long long wphase[kMaxPhases];
  int halo_l, halo_r;
  long long floats;
};

inline int pad8(int c) { return (c + 7) / 8 * 8; }
inline int pad32(int c) { return (c + 31) / 32 * 32; }

// Conv1d (S = 1): tap j reads x[t + j*dil - pad].  ConvTranspose1d (stride S, padding pad):
// out[q*S + r] += W[ci][co][j] * x[ci][(q*S + r + pad - j)/S] for j == (r + pad) mod S.
ConvGeom conv_geom(int Cout, int Cin, int K, int dil, int pad, int S, bool bf = false) {
  ConvGeom g;
  memset(&g, 0, sizeof(g));
  g.S = S;
  long long o = 0;
  int lo = 0, hi = 0;
  // fp32 packing: [ct][tap][8-channel group][lane][4 floats]; bf16-split packing: [ct][tap][16-channel step][2 planes][lane][8 bf16]
  const long long per_tap = bf ? (long long)(pad32(Cout) / 32) * ((Cin + 15) / 16) * 512
                               : (long long)(pad32(Cout) / 32) * (pad8(Cin) / 8) * 256;
  for (int r = 0; r < S; ++r) {
    g.wphase[r] = o;
    int n = 0;
    if (S == 1) {
      for (int j = 0; j < K; ++j) { g.off[r][n] = j * dil - pad; g.tapk[r][n] = j; ++n; }
    } else {
      const int j0 = (r + pad) % S, c = (r + pad) / S;
      for (int i = 0; j0 + S * i < K; ++i) { g.off[r][n] = c - i; g.tapk[r][n] = j0 + S * i; ++n; }
    }
    g.ntaps[r] = n;
    for (int i = 0; i < n; ++i) { lo = g.off[r][i] < lo ? g.off[r][i] : lo; hi = g.off[r][i] > hi ? g.off[r][i] : hi; }
    o += per_tap * n;
  }
  g.halo_l = -lo; g.halo_r = hi; g.floats = o;
  return g;
}

struct Launch {
  int kind;          // 0 conv, 1 dwln, 2 codebook, 3 fsq, 4 zero-tail, 5 fused ResidualUnit, 9 closure (fn)
  std::function<void(hipStream_t)> fn;
  std::string name;
  double flops;
  ConvP cp; int qb; bool ks; int chg; int nwv = 4; bool gemv; bool bf = false; bool c1 = false; int c1_len = 0; dim3 grid; size_t lds;
  int tph = 0;       // k_convbT: output phases per block of a transposed conv (0: k_convb, one phase per block)
  ResP rp; int res_nwv = 0;   // kind 5: a fused ResidualUnit (k_resunit) with res_nwv waves per block
  LnP lp; int cpt;
  // small kernels keep their args here
  const int64_t* sem; int semstride; const float* cb; int D, cbsize; float* Z; int zstride; long long zb;
  FsqP fp; int B;
  float* wav; int wstride, hop, tmaxhop;
  const int* lens;
};

int run_launch(const Launch& L, hipStream_t st) {
  switch (L.kind) {
    case 0:
      if (L.gemv) {
        hipLaunchKernelGGL(k_gemv1, L.grid, dim3(256), 0, st, L.cp);
      } else if (L.c1) {            // one output channel, 7 taps: a thread per output sample (k_conv_c1)
        hipLaunchKernelGGL((k_conv_c1<7>), dim3((L.c1_len + 255) / 256, L.grid.z), dim3(256), (size_t)L.cp.Cin * 7 * 4, st, L.cp);
      } else if (L.bf && L.tph) {   // transposed conv, several output phases per block (k_convbT)
        if (L.tph == 4) hipLaunchKernelGGL((k_convbT<4>), L.grid, dim3(256), L.lds, st, L.cp);
        else hipLaunchKernelGGL((k_convbT<5>), L.grid, dim3(256), L.lds, st, L.cp);
      } else if (L.bf) {            // bf16-split matrix pipe (k_convb): weights packed as two bf16 planes
#define SMI_CB(QB_, KS_, CHG_, NC_) hipLaunchKernelGGL((k_convb<QB_, KS_, CHG_, NC_>), L.grid, dim3(256), L.lds, st, L.cp)
        const bool wide = L.cp.xw > 64;
        if (L.chg == 4) {
          // one tap, channels split over the waves, at most two blocks per CU: the chunk's weights are requested one chunk ahead (WPF)
          const bool wpf = L.ks && L.cp.S == 1 && L.cp.ntaps[0] == 1 && (long long)L.grid.x * L.grid.y * L.grid.z <= 512 && !smi_env("SPARKMI_CB_NOWPF");
          if (wpf) { if (L.qb == 1) hipLaunchKernelGGL((k_convb<1, true, 4, 1, true>), L.grid, dim3(256), L.lds, st, L.cp); else hipLaunchKernelGGL((k_convb<2, true, 4, 1, true>), L.grid, dim3(256), L.lds, st, L.cp); }
          else if (L.ks) { if (L.qb == 1) SMI_CB(1, true, 4, 1); else SMI_CB(2, true, 4, 1); }
          else { if (L.qb == 1) SMI_CB(1, false, 4, 1); else SMI_CB(2, false, 4, 1); }
        } else if (L.ks) {
          // several taps on a grid of about a block per CU: all taps' weights of a chunk requested together (WALL)
          int mt = 0;
          for (int r = 0; r < L.cp.S; ++r) mt = L.cp.ntaps[r] > mt ? L.cp.ntaps[r] : mt;
          // (measured at one / two utterances, 150 frames: 7-tap convs 71 -> 62 us at 456 blocks but 101 -> 115 at 912; the first transposed conv,
          // 2 taps per phase, 85 -> 129 us: stride-1 layers with four taps or more on at most two blocks per CU)
          const bool wall = L.chg == 2 && mt >= 4 && L.cp.S == 1 && (long long)L.grid.x * L.grid.y * L.grid.z <= 512 && !smi_env("SPARKMI_CB_NOWALL");
#define SMI_CBW(QB_, NC_) hipLaunchKernelGGL((k_convb<QB_, true, 2, NC_, false, true>), L.grid, dim3(256), L.lds, st, L.cp)
          if (wall) { if (L.qb == 1) { if (wide) SMI_CBW(1, 2); else SMI_CBW(1, 1); } else SMI_CBW(2, 2); }
          else if (L.qb == 1) { if (wide) SMI_CB(1, true, 2, 2); else SMI_CB(1, true, 2, 1); }
          else SMI_CB(2, true, 2, 2);
#undef SMI_CBW
        } else if (L.chg == 2) {    // few taps per phase, rows wider than 64 columns (transposed convs): 64-channel chunks
          if (L.qb == 1) SMI_CB(1, false, 2, 2); else SMI_CB(2, false, 2, 2);
        } else {
          if (L.qb == 1) { if (wide) SMI_CB(1, false, 1, 2); else SMI_CB(1, false, 1, 1); }
          else SMI_CB(2, false, 1, 2);
        }
#undef SMI_CB
      } else if (L.chg == 4) {      // 1-tap layers: 128-channel chunks, narrow rows
        if (L.ks) {
          // measured (profiles/README.md): pays up to about two blocks per CU, costs beyond (fewer waves fit)
          const bool wpf = (long long)L.grid.x * L.grid.y * L.grid.z <= 512;
          if (L.qb == 1 && wpf) hipLaunchKernelGGL((k_conv<1, true, 4, 1, true>), L.grid, dim3(256), L.lds, st, L.cp);
          else if (L.qb == 1) hipLaunchKernelGGL((k_conv<1, true, 4, 1>), L.grid, dim3(256), L.lds, st, L.cp);
          else if (wpf) hipLaunchKernelGGL((k_conv<2, true, 4, 1, true>), L.grid, dim3(256), L.lds, st, L.cp);
          else hipLaunchKernelGGL((k_conv<2, true, 4, 1>), L.grid, dim3(256), L.lds, st, L.cp);
        } else {
          if (L.qb == 1) hipLaunchKernelGGL((k_conv<1, false, 4, 1>), L.grid, dim3(256), L.lds, st, L.cp);
          else hipLaunchKernelGGL((k_conv<2, false, 4, 1>), L.grid, dim3(256), L.lds, st, L.cp);
        }
      } else if (L.nwv == 3) {      // three output tiles per block (never channel-split)
        if (L.cp.xw > 128) {
          if (L.qb == 1) hipLaunchKernelGGL((k_conv<1, false, 1, 3, false, 3>), L.grid, dim3(192), L.lds, st, L.cp);
          else hipLaunchKernelGGL((k_conv<2, false, 1, 3, false, 3>), L.grid, dim3(192), L.lds, st, L.cp);
        } else {
          if (L.qb == 1) hipLaunchKernelGGL((k_conv<1, false, 1, 2, false, 3>), L.grid, dim3(192), L.lds, st, L.cp);
          else hipLaunchKernelGGL((k_conv<2, false, 1, 2, false, 3>), L.grid, dim3(192), L.lds, st, L.cp);
        }
      } else if (L.cp.xw > 128) {   // strided convs: up to 192 staged columns
        if (L.ks) {
          if (L.qb == 1) hipLaunchKernelGGL((k_conv<1, true, 1, 3>), L.grid, dim3(256), L.lds, st, L.cp);
          else hipLaunchKernelGGL((k_conv<2, true, 1, 3>), L.grid, dim3(256), L.lds, st, L.cp);
        } else {
          if (L.qb == 1) hipLaunchKernelGGL((k_conv<1, false, 1, 3>), L.grid, dim3(256), L.lds, st, L.cp);
          else hipLaunchKernelGGL((k_conv<2, false, 1, 3>), L.grid, dim3(256), L.lds, st, L.cp);
        }
      } else if (L.ks) {
        if (L.qb == 1) hipLaunchKernelGGL((k_conv<1, true, 1, 2>), L.grid, dim3(256), L.lds, st, L.cp);
        else hipLaunchKernelGGL((k_conv<2, true, 1, 2>), L.grid, dim3(256), L.lds, st, L.cp);
      } else {
        if (L.qb == 1) hipLaunchKernelGGL((k_conv<1, false, 1, 2>), L.grid, dim3(256), L.lds, st, L.cp);
        else hipLaunchKernelGGL((k_conv<2, false, 1, 2>), L.grid, dim3(256), L.lds, st, L.cp);
      }
      break;
    case 1:
      if (L.cpt <= 2) hipLaunchKernelGGL(k_dwln<2>, L.grid, dim3(256), 0, st, L.lp);
      else if (L.cpt <= 4) hipLaunchKernelGGL(k_dwln<4>, L.grid, dim3(256), 0, st, L.lp);
      else if (L.cpt <= 12) hipLaunchKernelGGL(k_dwln<12>, L.grid, dim3(256), 0, st, L.lp);
      else if (L.cpt <= 16) hipLaunchKernelGGL(k_dwln<16>, L.grid, dim3(256), 0, st, L.lp);
      else hipLaunchKernelGGL(k_dwln<32>, L.grid, dim3(256), 0, st, L.lp);
      break;
    case 2:
      hipLaunchKernelGGL(k_codebook, L.grid, dim3(128), 0, st, L.sem, L.semstride, L.cb, L.D, L.cbsize, L.lens, L.Z, L.zstride, L.zb);
      break;
    case 3:
      hipLaunchKernelGGL(k_fsq, dim3(L.B), dim3(256), 0, st, L.fp);
      break;
    case 4:
      hipLaunchKernelGGL(k_zero_tail, L.grid, dim3(256), 0, st, L.wav, L.wstride, L.lens, L.hop);
      break;
    case 5:
      if (L.res_nwv == 3) hipLaunchKernelGGL((k_resunit<3>), L.grid, dim3(192), L.lds, st, L.rp);
      else hipLaunchKernelGGL((k_resunit<6>), L.grid, dim3(384), L.lds, st, L.rp);
      break;
    case 9:
      L.fn(st);
      break;
  }
  SMI_LAUNCH_CHECK();
  return SMI_OK;
}

// Build one conv launch.  X/Y strides are in floats; Lmax = padded INPUT length (time units).
Launch make_conv_w(const std::string& name, const float* W, const float* bias,
                   int Cout, int Cin, int K, int dil, int S, int pad, const float* X, int xstride, long long xb,
                 float* Y, float* Ys, const float* alpha, const float* R, int ystride, long long yb,
                 const int* lens, int B, int Lmax, int act, int istr = 1, const int* olens = nullptr, bool bf = false) {
  // Lmax = padded OUTPUT positions per phase (= input length for stride-1 convs and transposed convs)
  // bf: the weights are packed as bf16 planes (PACK_CONV_B / PACK_CONVT_B) and the layer runs on k_convb
  Launch L;
  L.kind = 0; L.name = name; L.bf = bf;
  ConvGeom g = conv_geom(Cout, Cin, K, dil, pad, S, bf);
  ConvP& p = L.cp;
  memset(&p, 0, sizeof(p));
  p.X = X; p.W = W; p.bias = bias;
  p.R = R; p.alpha = alpha; p.Y = Y; p.Ys = Ys; p.lens = lens;
  p.Cin = Cin; p.CinP = pad8(Cin); p.Cout = Cout; p.S = S; p.act = act;
  p.xstride = xstride; p.ystride = ystride; p.xb = xb; p.yb = yb;
  p.out_scale = 1.0f;
  { const char* e = smi_env("SPARKMI_SNAKE_SINF"); p.fast_sin = bf && !(e && e[0] == '1'); }   // SPARKMI_SNAKE_SINF=1: library sine everywhere (A/B)
  p.istr = istr; p.olens = olens;
  for (int r = 0; r < S; ++r) {
    p.ntaps[r] = g.ntaps[r];
    p.wphase[r] = g.wphase[r];
    for (int i = 0; i < g.ntaps[r]; ++i) p.off[r][i] = g.off[r][i];
  }
  const int cot = pad32(Cout) / 32;
  // waves split the input channels when there are few time tiles and many channels
  // 64-column time tiles unless that leaves most CUs idle (short sequences): then 32-column tiles double the blocks
  const long long blocks64 = (long long)((Lmax + 63) / 64) * cot * B * S;
  const int qb = (Lmax <= 32 || (blocks64 < 256 && smi_env("SPARKMI_QB2") == nullptr)) ? 1 : 2;
  const int qt = qb * 32, nq = (Lmax + qt - 1) / qt;
  const long long blocks_cosplit = (long long)nq * ((cot + 3) / 4) * B * S;
  // (a partial last group of output tiles just idles its spare waves).  The choice depends on the call's shape
  // (B, longest row): a row is independent of its neighbours and padding bit for bit, and equal to fp32
  // re-association between calls of different shapes.
  L.ks = blocks_cosplit < 512 && Cin >= 8;
  L.qb = qb;
  p.halo_l = g.halo_l;
  p.xw = (qt - 1) * istr + 1 + g.halo_l + g.halo_r;
  // three-wave blocks where four would leave a wave (and so a SIMD of each CU) without an output tile; measured at
  // 150 frames: 75.0 -> 73.2 ms at B = 32, 21.0 -> 20.5 at B = 8, but 4.66 -> 4.72 at B = 1, hence the grid-size condition
  L.nwv = (!bf && !L.ks && cot % 3 == 0 && cot % 4 != 0 && blocks_cosplit >= 2048) ? 3 : 4;
  L.grid = dim3(nq, L.ks ? cot : (cot + L.nwv - 1) / L.nwv, B * S);
  L.chg = (S == 1 && K == 1 && Cin >= 128 && L.ks) ? 4 : 1;   // K-split 1-tap layers stage 128 channels per chunk
  if (bf && L.ks && L.chg == 1) L.chg = 2;                     // k_convb: a channel-split wave owns whole 16-channel steps
  // k_convb, one output tile per wave, few taps per phase (1x1 convs: 1; transposed convs: 2-3): a 32-channel chunk carries only
  // 6 * taps MFMAs per wave between its two barriers and its staging round trip.  128-channel chunks where the staged row is one
  // 64-column pass (1 tap), 64-channel chunks up to 128 columns (round 4; SPARKMI_CB_CHG=0: 32-channel chunks everywhere, A/B).
  // A 1-tap layer sums its channels in the same order whatever the chunk; a transposed conv's order goes from (32 channels, taps)
  // to (64 channels, taps) -- the choice depends on the layer's shape only, never on the batch.
  if (bf && !L.ks && istr == 1 && Cin >= 128) {
    int maxtap = 0;
    for (int r = 0; r < S; ++r) maxtap = g.ntaps[r] > maxtap ? g.ntaps[r] : maxtap;
    const char* e = smi_env("SPARKMI_CB_CHG");
    if (maxtap <= 3 && !(e && e[0] == '0')) L.chg = p.xw <= 64 ? 4 : (p.xw <= 128 ? 2 : 1);
  }
  // transposed convs with a grid that still fills the chip with PH phases per block: k_convbT (SPARKMI_CBT=0: off, A/B).  Its
  // epilogue knows bias, the raw output and the Snake'd output only: check_launches() rejects a program that set anything else
  // on such a launch after this function returned.
  if (bf && !L.ks && S > 1 && istr == 1 && Cin >= 64 && olens == nullptr) {
    const int ph = S % 4 == 0 ? 4 : (S == 5 ? 5 : 0);   // (S = 2: two phases x two column tiles measured level with k_convb -- it stays there)
    const int qbt = 1;
    const char* e = smi_env("SPARKMI_CBT");
    if (ph && !(e && e[0] == '0')) {
      const int qtt = qbt * 32, nqt = (Lmax + qtt - 1) / qtt;
      const int xwt = qtt + g.halo_l + g.halo_r;
      // (eight blocks per CU and more: at 900 blocks the stride-5 layer of an 8-row batch lost 0.40 -> 0.48 ms, at 3000+ every layer gained)
      if ((long long)nqt * ((cot + 3) / 4) * B * (S / ph) >= 2048 && xwt <= 64 && !R && act == ACT_NONE) {
        L.tph = ph; L.qb = qbt; L.chg = 2;
        p.xw = xwt;
        L.grid = dim3(nqt, (cot + 3) / 4, B * (S / ph));
        L.lds = (size_t)2 * 8 * xwt * 16;
      }
    }
  }
  L.gemv = false;   // set by the caller for the per-utterance vector projections (use_gemv)
  size_t lds = (size_t)8 * L.nwv * L.chg * p.xw * 4;
  const size_t red = L.ks ? (size_t)4 * qb * 16 * 64 * 4 : 0;
  L.lds = lds > red ? lds : red;
  double taps = 0;
  for (int r = 0; r < S; ++r) taps += g.ntaps[r];
  L.flops = 2.0 * Cout * Cin * taps * Lmax * B;   // all phases together cover S*Lmax outputs
  return L;
}

}  // namespace
