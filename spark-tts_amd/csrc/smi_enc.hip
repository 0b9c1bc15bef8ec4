// smi_enc.hip -- voice-clone prompt encoder for gfx950 (MI355X): BiCodecTokenizer.tokenize
// (sparktts/models/audio_tokenizer.py:85-130) behind a flat C ABI (include/sparkmi.h, smi_enc_*).
//
//   wav -> zero-mean / unit-variance (Wav2Vec2FeatureExtractor)                 k_wavnorm
//       -> wav2vec2 feature encoder: 7 x [Conv1d(stride) -> LayerNorm -> GELU]  k_conv0 / k_conv(istr) + k_dwln
//       -> LayerNorm -> Linear 512->1024, + GELU(grouped pos-conv k=128)        k_dwln, k_conv, k_posconv
//       -> 16 pre-LN transformer layers (only hidden states 11/14/16 are used)  k_dwln, k_conv, k_mha
//       -> mean of the three tapped hidden states                               k_tap
//   feat -> BiCodec Encoder (Vocos backbone + 2 x [3x, Vocos(2)] + Linear)      k_conv, k_dwln  (feat_encoder.py:76-87)
//        -> in_project, cosine-VQ arg-max over the codebook = semantic ids      k_conv, k_vq    (factorized_vector_quantize.py:148-187)
//   ref clip -> mel (framed DFT as a GEMM, magnitude, slaney filterbank)        k_frames, k_conv, k_mag
//        -> ECAPA-TDNN up to its latent                                         k_conv(ReLU+BN epilogue), k_rowmean, k_gemv1, k_se
//        -> perceiver resampler (2 x [cross-attn incl. queries, GEGLU FF])      k_conv, k_mha, k_geglu, k_rmsn
//        -> FSQ project_in / bound / round = global ids                         k_fsq_quant     (residual_fsq.py:211-276)
//
// Activations are [C][T] fp32 (time contiguous), one utterance per call like the reference's tokenize().
// Dense contractions run on the exact-fp32 matrix pipe through the vocoder's implicit-GEMM kernel
// (smi_net.h); attention and the 128-tap grouped positional conv are VALU kernels (small: T <= ~1500).
#include "smi_net.h"
#include <math.h>
#include <map>

namespace {

// ------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------

// (x - mean) / sqrt(var + 1e-7), feature_extraction_wav2vec2.py zero_mean_unit_var_norm; one block.
__global__ __launch_bounds__(1024) void k_wavnorm(const float* x, int n, float* y) {
  __shared__ double red[1024];
  const int tid = threadIdx.x;
  double s = 0.0;
  for (int i = tid; i < n; i += 1024) s += (double)x[i];
  red[tid] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  const double mean = red[0] / n;
  __syncthreads();
  double q = 0.0;
  for (int i = tid; i < n; i += 1024) { const double d = (double)x[i] - mean; q += d * d; }
  red[tid] = q;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  const float fm = (float)mean;
  const float rs = 1.0f / sqrtf((float)(red[0] / n) + 1e-7f);
  for (int i = tid; i < n; i += 1024) y[i] = (x[i] - fm) * rs;
}

// first feature-encoder conv: Conv1d(1 -> C, K, stride), no padding.  W [C][K].
__global__ void k_conv0(const float* x, const float* W, const float* bias, int K, int stride, float* Y, int C, int T, int ystride) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (t >= T) return;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc += W[c * K + k] * x[t * stride + k];
  Y[(long long)c * ystride + t] = acc + (bias ? bias[c] : 0.f);
}

// grouped positional conv (Wav2Vec2PositionalConvEmbedding): out[c][t] = x[c][t] + gelu(b[c] +
//   sum_{ci < Cg, k < K} W[c][ci][k] * x[g*Cg + ci][t + k - K/2]);  the SamePad layer drops the extra last frame.
// Block = 16 output channels of one group x 64 frames; wave w computes channels 4w..4w+3 (weights wave-uniform).
__global__ __launch_bounds__(256) void k_posconv(const float* X, const float* W, const float* bias, float* Y, int C, int Cg, int K,
                                                 int T, int stride) {
  extern __shared__ float xs[];            // [Cg][64 + K - 1]
  const int t0 = blockIdx.x * 64, cot = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int co0 = cot * 16 + wave * 4, g = (cot * 16) / Cg;
  const int xw = 64 + K - 1;
  for (int i = threadIdx.x; i < Cg * xw; i += 256) {
    const int ci = i / xw, col = i - ci * xw, t = t0 + col - K / 2;
    xs[i] = (t >= 0 && t < T) ? X[(long long)(g * Cg + ci) * stride + t] : 0.f;
  }
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* w0 = W + (long long)co0 * Cg * K;
  for (int ci = 0; ci < Cg; ++ci) {
    const float* xr = xs + ci * xw + lane;
    for (int k = 0; k < K; ++k) {
      const float xv = xr[k];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] += w0[((long long)c * Cg + ci) * K + k] * xv;
    }
  }
  const int t = t0 + lane;
  if (t < T) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long long o = (long long)(co0 + c) * stride + t;
      Y[o] = X[o] + gelu_f(acc[c] + bias[co0 + c]);
    }
  }
}

// Multi-head attention, head_dim 64, no mask: O[h*64+d][q] = sum_j softmax_j(scale * Q[:,q].K[:,j]) V[h*64+d][j].
// Q/K/V/O are channel-major [rows][stride].  Block = (8 queries, head); scores of the 8 rows live in LDS.
struct MhaP {
  const float *Q, *K, *V;
  float* O;
  int qs, ks, vs, os;   // row strides
  int Tq, Tk;
  float scale;
};
__global__ __launch_bounds__(256) void k_mha(MhaP p) {
  extern __shared__ float sm[];
  float* qs = sm;                       // [64][8]
  float* S = sm + 512;                  // [8][Tk]
  float* Vs = S + 8 * p.Tk;             // [64][65]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.x * 8, h = blockIdx.y;
  const float* Qh = p.Q + (long long)h * 64 * p.qs;
  const float* Kh = p.K + (long long)h * 64 * p.ks;
  const float* Vh = p.V + (long long)h * 64 * p.vs;
  for (int i = tid; i < 512; i += 256) {
    const int d = i >> 3, qi = i & 7;
    qs[i] = (q0 + qi < p.Tq) ? Qh[(long long)d * p.qs + q0 + qi] : 0.f;
  }
  __syncthreads();
  for (int j = tid; j < p.Tk; j += 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < 64; ++d) {
      const float kv = Kh[(long long)d * p.ks + j];
#pragma unroll
      for (int qi = 0; qi < 8; ++qi) acc[qi] += qs[d * 8 + qi] * kv;
    }
#pragma unroll
    for (int qi = 0; qi < 8; ++qi) S[qi * p.Tk + j] = acc[qi] * p.scale;
  }
  __syncthreads();
  // softmax: wave w owns rows 2w, 2w+1
  for (int r = wave * 2; r < wave * 2 + 2; ++r) {
    float* row = S + r * p.Tk;
    float m = -INFINITY;
    for (int j = lane; j < p.Tk; j += 64) m = fmaxf(m, row[j]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
    for (int j = lane; j < p.Tk; j += 64) { const float e = expf(row[j] - m); row[j] = e; s += e; }
    s = smi_wave_sum(s);
    const float inv = 1.0f / s;
    for (int j = lane; j < p.Tk; j += 64) row[j] *= inv;
  }
  __syncthreads();
  // O = P V: thread (d = lane, rows 2*wave, 2*wave+1); V staged 64 keys at a time
  float o0 = 0.f, o1 = 0.f;
  const float* P0 = S + (wave * 2) * p.Tk;
  const float* P1 = P0 + p.Tk;
  for (int j0 = 0; j0 < p.Tk; j0 += 64) {
    for (int i = tid; i < 64 * 64; i += 256) {
      const int d = i >> 6, jj = i & 63;
      Vs[d * 65 + jj] = (j0 + jj < p.Tk) ? Vh[(long long)d * p.vs + j0 + jj] : 0.f;
    }
    __syncthreads();
    const int n = p.Tk - j0 < 64 ? p.Tk - j0 : 64;
    for (int jj = 0; jj < n; ++jj) {
      const float v = Vs[lane * 65 + jj];
      o0 += P0[j0 + jj] * v;
      o1 += P1[j0 + jj] * v;
    }
    __syncthreads();
  }
  float* Oh = p.O + (long long)h * 64 * p.os;
  const int qa = q0 + wave * 2;
  if (qa < p.Tq) Oh[(long long)lane * p.os + qa] = o0;
  if (qa + 1 < p.Tq) Oh[(long long)lane * p.os + qa + 1] = o1;
}

// hidden-state taps: mode 0: acc = h; 1: acc = acc + h; 2: out = (acc + h) / 3   (audio_tokenizer.py:96-98)
__global__ void k_tap(const float* h, float* acc, float* out, long long n, int mode) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 0) acc[i] = h[i];
  else if (mode == 1) acc[i] = acc[i] + h[i];
  else out[i] = (acc[i] + h[i]) / 3.0f;
}

// framed, reflect-padded reference clip: F[k][t] = xpad[t*hop + k], xpad = reflect pad of n_fft/2 (torch.stft center=True)
__global__ void k_frames(const float* x, int n, int n_fft, int hop, float* F, int T, int stride) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (t >= T) return;
  int i = t * hop + k - n_fft / 2;
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  F[(long long)k * stride + t] = x[i];
}

// |re + i im| of the DFT rows: D [2*nf][stride] (re rows then im rows) -> M [nf][stride]
__global__ void k_mag(const float* D, int nf, int T, int stride, float* M) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
  if (t >= T) return;
  const float re = D[(long long)f * stride + t], im = D[(long long)(nf + f) * stride + t];
  M[(long long)f * stride + t] = sqrtf(re * re + im * im);
}

// mean over time of every channel (SE_Connect, ecapa_tdnn.py:104): one wave per channel
__global__ __launch_bounds__(256) void k_rowmean(const float* X, int C, int T, int stride, float* out) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  float s = 0.f;
  for (int t = lane; t < T; t += 64) s += X[(long long)c * stride + t];
  s = smi_wave_sum(s);
  if (lane == 0) out[c] = s / (float)T;
}

// SE_Res2Block tail: out[c][t] = xin[c][t] + y[c][t] * s[c]   (ecapa_tdnn.py:107,133)
__global__ void k_se(const float* xin, const float* y, const float* s, float* out, int T, int istride, int ostride) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (t >= T) return;
  out[(long long)c * ostride + t] = xin[(long long)c * istride + t] + y[(long long)c * istride + t] * s[c];
}

// GEGLU (perceiver_encoder.py:213-216): Y[c][t] = gelu(X[inner + c][t]) * X[c][t]
__global__ void k_geglu(const float* X, int inner, int T, int stride, float* Y) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (t >= T) return;
  Y[(long long)c * stride + t] = gelu_f(X[(long long)(inner + c) * stride + t]) * X[(long long)c * stride + t];
}

// perceiver RMSNorm over channels of T columns: x / max(||x||, 1e-12) * sqrt(C) * gamma (perceiver_encoder.py:180-191)
__global__ void k_rmsn(const float* X, int C, int T, int stride, const float* gamma, float* Y, int ystride) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  float ss = 0.f;
  for (int c = 0; c < C; ++c) { const float v = X[(long long)c * stride + t]; ss += v * v; }
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  const float sc = sqrtf((float)C);
  for (int c = 0; c < C; ++c) Y[(long long)c * ystride + t] = X[(long long)c * stride + t] / nrm * sc * gamma[c];
}

// codebook rows L2-normalised once at create (F.normalize(codebook), factorized_vector_quantize.py:176)
__global__ void k_cbnorm(const float* cb, int n, int D, float* out, float* c2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float ss = 0.f;
  for (int d = 0; d < D; ++d) ss += cb[i * D + d] * cb[i * D + d];
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  float s2 = 0.f;
  for (int d = 0; d < D; ++d) { const float v = cb[i * D + d] / nrm; out[i * D + d] = v; s2 += v * v; }
  c2[i] = s2;
}

// cosine-VQ arg-max of one frame per block: -(|e|^2 - 2 e.c + |c|^2), lowest index on ties (torch .max(1)[1])
__global__ __launch_bounds__(256) void k_vq(const float* Ze, int D, int T, int stride, const float* cbn, const float* c2, int ncode,
                                            int64_t* sem) {
  __shared__ float bv[256];
  __shared__ int bi[256];
  const int t = blockIdx.x, tid = threadIdx.x;
  float e[16];
  float ss = 0.f;
  for (int d = 0; d < D; ++d) { e[d] = Ze[(long long)d * stride + t]; ss += e[d] * e[d]; }
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  float e2 = 0.f;
  for (int d = 0; d < D; ++d) { e[d] = e[d] / nrm; e2 += e[d] * e[d]; }
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int i = tid; i < ncode; i += 256) {
    float dot = 0.f;
    for (int d = 0; d < D; ++d) dot += e[d] * cbn[i * D + d];
    const float v = -((e2 - 2.0f * dot) + c2[i]);
    if (v > best) { best = v; besti = i; }
  }
  bv[tid] = best; bi[tid] = besti;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      if (bv[tid + o] > bv[tid] || (bv[tid + o] == bv[tid] && bi[tid + o] < bi[tid])) { bv[tid] = bv[tid + o]; bi[tid] = bi[tid + o]; }
    }
    __syncthreads();
  }
  if (tid == 0) sem[t] = bi[0];
}

// FSQ: z = W x + b; bounded = tanh(z + shift) * half_l - offset; round (half to even, torch.round); index
// (finite_scalar_quantization.py:113-137, residual_fsq.py:211-276 with one quantizer)
struct FsqQP {
  const float* X;     // [latent][stride]
  const float* W;     // [nd][latent]
  const float* b;     // [nd]
  int32_t* out;       // [Ntok]
  float* bounded;     // [Ntok][nd] (debug) or null
  int latent, stride, Ntok, nd;
  int levels[8];
};
__global__ void k_fsq_quant(FsqQP p) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.Ntok) return;
  int idx = 0, basis = 1;
  for (int j = 0; j < p.nd; ++j) {
    float z = 0.f;
    for (int d = 0; d < p.latent; ++d) z += p.W[j * p.latent + d] * p.X[(long long)d * p.stride + t];
    z += p.b[j];
    const int L = p.levels[j];
    const float half_l = (float)(L - 1) * (1.0f + 1e-3f) / 2.0f;
    const float offset = (L % 2 == 0) ? 0.5f : 0.0f;
    const float shift = atanhf(offset / half_l);
    const float bd = tanhf(z + shift) * half_l - offset;
    if (p.bounded) p.bounded[t * p.nd + j] = bd;
    const float q = rintf(bd);
    const int half_w = L / 2;
    idx += ((int)q + half_w) * basis;
    basis *= L;
  }
  p.out[t] = idx;
}

__global__ void k_copy2d(const float* src, int sstride, float* dst, int dstride, int rows, int cols) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (t < cols && r < rows) dst[(long long)r * dstride + t] = src[(long long)r * sstride + t];
}

// ------------------------------------------------------------------------------------------
// layout
// ------------------------------------------------------------------------------------------
struct EncLayout {
  std::vector<Entry> e;
  size_t total;
};

bool enc_cfg_ok(const smi_enc_cfg* c) {
  if (!c) return false;
  if (c->w2v_nconv < 2 || c->w2v_nconv > 8 || c->w2v_conv_dim < 32 || c->w2v_conv_dim % 32 || c->w2v_conv_dim > 1024) return false;
  for (int i = 0; i < c->w2v_nconv; ++i)
    if (c->w2v_kernel[i] < 1 || c->w2v_stride[i] < 1 || (i > 0 && (c->w2v_kernel[i] > kMaxTaps || c->w2v_stride[i] > 2))) return false;
  if (c->w2v_hidden % 64 || c->w2v_hidden != c->w2v_heads * 64 || c->w2v_hidden > 1024 || c->w2v_layers < 1) return false;
  if (c->w2v_pos_k < 2 || c->w2v_pos_k % 2 || c->w2v_pos_groups < 1 || c->w2v_hidden % c->w2v_pos_groups) return false;
  if ((c->w2v_hidden / c->w2v_pos_groups) % 16) return false;
  for (int i = 0; i < 3; ++i) if (c->w2v_taps[i] < 0 || c->w2v_taps[i] > c->w2v_layers) return false;
  if (c->enc_in != c->w2v_hidden || c->enc_dim < 1 || c->enc_dim > 1024 || c->enc_layers < 1 || c->enc_num_down < 0) return false;
  if (c->codebook_dim < 1 || c->codebook_dim > 16 || c->codebook_size < 1) return false;
  if (c->n_fft < 16 || c->n_fft % 2 || c->win_length > c->n_fft || c->hop_length < 1 || c->num_mels < 1) return false;
  if (c->ecapa_channels % 64 || c->ecapa_channels < 64 || c->spk_latent < 1 || c->spk_tokens < 1) return false;
  if (c->fsq_dims < 1 || c->fsq_dims > 8 || c->perc_depth < 1 || c->perc_heads < 1 || c->perc_ff_inner < 1) return false;
  if (c->max_samples < 400 || c->max_ref_samples < c->n_fft) return false;
  return true;
}

EncLayout enc_layout(const smi_enc_cfg* c) {
  EncLayout L;
  size_t o = 0;
  auto add = [&](const std::string& name, int kind, int Cout, int Cin, int K, size_t floats) {
    Entry e{name, kind, Cout, Cin, K, 1, 0, o, floats * 4};
    L.e.push_back(e);
    o += smi_align_up(floats * 4, 256);
  };
  auto raw = [&](const std::string& name, size_t n) { add(name, PACK_RAW, 0, 0, 0, n); };
  auto conv = [&](const std::string& name, int Cout, int Cin, int K) {
    add(name, PACK_CONV, Cout, Cin, K, (size_t)conv_geom(Cout, Cin, K, 1, 0, 1).floats);
  };
  // convb: the dense stride-1 layers that carry the FLOPs (wav2vec2's transformer projections, the BiCodec encoder's
  // ConvNeXt stack) -- on the bf16-split matrix pipe (k_convb, smi_net.h) unless the handle asks for exact fp32
  auto convb = [&](const std::string& name, int Cout, int Cin, int K) {
    const bool bf = !c->exact_fp32 && Cin >= 32 && Cout >= 32;
    add(name, bf ? PACK_CONV_B : PACK_CONV, Cout, Cin, K, (size_t)conv_geom(Cout, Cin, K, 1, 0, 1, bf).floats);
  };
  // ---- wav2vec2
  const int CD = c->w2v_conv_dim, H = c->w2v_hidden, I = c->w2v_inter;
  for (int i = 0; i < c->w2v_nconv; ++i) {
    const std::string p = "w2v.feature_extractor.conv_layers." + std::to_string(i);
    if (i == 0) raw(p + ".conv.weight", (size_t)CD * c->w2v_kernel[0]);
    else conv(p + ".conv.weight", CD, CD, c->w2v_kernel[i]);
    raw(p + ".conv.bias", CD);
    raw(p + ".layer_norm.weight", CD);
    raw(p + ".layer_norm.bias", CD);
  }
  raw("w2v.feature_projection.layer_norm.weight", CD);
  raw("w2v.feature_projection.layer_norm.bias", CD);
  conv("w2v.feature_projection.projection.weight", H, CD, 1);
  raw("w2v.feature_projection.projection.bias", H);
  raw("w2v.encoder.pos_conv_embed.conv.weight", (size_t)H * (H / c->w2v_pos_groups) * c->w2v_pos_k);
  raw("w2v.encoder.pos_conv_embed.conv.bias", H);
  for (int l = 0; l < c->w2v_layers; ++l) {
    const std::string p = "w2v.encoder.layers." + std::to_string(l);
    raw(p + ".layer_norm.weight", H);
    raw(p + ".layer_norm.bias", H);
    convb("cat:" + p + ".attention.q_proj.weight|" + p + ".attention.k_proj.weight|" + p + ".attention.v_proj.weight", 3 * H, H, 1);
    raw("cat:" + p + ".attention.q_proj.bias|" + p + ".attention.k_proj.bias|" + p + ".attention.v_proj.bias", (size_t)3 * H);
    convb(p + ".attention.out_proj.weight", H, H, 1);
    raw(p + ".attention.out_proj.bias", H);
    raw(p + ".final_layer_norm.weight", H);
    raw(p + ".final_layer_norm.bias", H);
    convb(p + ".feed_forward.intermediate_dense.weight", I, H, 1);
    raw(p + ".feed_forward.intermediate_dense.bias", I);
    convb(p + ".feed_forward.output_dense.weight", H, I, 1);
    raw(p + ".feed_forward.output_dense.bias", H);
  }
  // ---- BiCodec encoder + quantizer
  const int D = c->enc_dim, EI = c->enc_inter;
  auto vocos = [&](const std::string& p, int cin, int nl) {
    convb(p + ".embed.weight", D, cin, 7);
    raw(p + ".embed.bias", D);
    raw(p + ".norm.weight", D);
    raw(p + ".norm.bias", D);
    for (int j = 0; j < nl; ++j) {
      const std::string b = p + ".convnext." + std::to_string(j);
      raw(b + ".dwconv.weight", (size_t)D * 7);
      raw(b + ".dwconv.bias", D);
      raw(b + ".norm.weight", D);
      raw(b + ".norm.bias", D);
      convb(b + ".pwconv1.weight", EI, D, 1);
      raw(b + ".pwconv1.bias", EI);
      convb(b + ".pwconv2.weight", D, EI, 1);
      raw(b + ".pwconv2.bias", D);
      raw(b + ".gamma", D);
    }
    raw(p + ".final_layer_norm.weight", D);
    raw(p + ".final_layer_norm.bias", D);
  };
  vocos("encoder.encoder", c->enc_in, c->enc_layers);
  for (int i = 0; i < c->enc_num_down; ++i) vocos("encoder.downsample." + std::to_string(i) + ".1", D, 2);
  conv("encoder.project.weight", c->enc_out, D, 1);
  raw("encoder.project.bias", c->enc_out);
  conv("quantizer.in_project.weight", c->codebook_dim, c->enc_out, 1);
  raw("quantizer.in_project.bias", c->codebook_dim);
  raw("quantizer.codebook.weight", (size_t)c->codebook_size * c->codebook_dim);
  // ---- mel (derived on the host)
  const int nf = c->n_fft / 2 + 1;
  conv("mel.dft", 2 * nf, c->n_fft, 1);
  conv("mel.fb", c->num_mels, nf, 1);
  // ---- ECAPA-TDNN up to `latent`
  const int C = c->ecapa_channels, W = C / 8;
  const std::string se = "speaker_encoder.speaker_encoder";
  auto crb = [&](const std::string& p, int co, int ci, int k) {
    conv(p + ".conv.weight", co, ci, k);
    raw(p + ".conv.bias", co);
    raw("bnscale:" + p + ".bn", co);
    raw("bnshift:" + p + ".bn", co);
  };
  crb(se + ".layer1", C, c->num_mels, 5);
  for (int li = 2; li <= 4; ++li) {
    const std::string b = se + ".layer" + std::to_string(li) + ".se_res2block";
    crb(b + ".0", C, C, 1);
    for (int j = 0; j < 7; ++j) {
      conv(b + ".1.convs." + std::to_string(j) + ".weight", W, W, 3);
      raw(b + ".1.convs." + std::to_string(j) + ".bias", W);
      raw("bnscale:" + b + ".1.bns." + std::to_string(j), W);
      raw("bnshift:" + b + ".1.bns." + std::to_string(j), W);
    }
    crb(b + ".2", C, C, 1);
    conv(b + ".3.linear1.weight", 128, C, 1);
    raw(b + ".3.linear1.bias", 128);
    conv(b + ".3.linear2.weight", C, 128, 1);
    raw(b + ".3.linear2.bias", C);
  }
  conv(se + ".conv.weight", c->ecapa_out, 3 * C, 1);
  raw(se + ".conv.bias", c->ecapa_out);
  // ---- perceiver + FSQ
  const std::string ps = "speaker_encoder.perceiver_sampler";
  const int Ld = c->spk_latent, inner = c->perc_heads * 64, FI = c->perc_ff_inner;
  conv(ps + ".proj_context.weight", Ld, c->ecapa_out, 1);
  raw(ps + ".proj_context.bias", Ld);
  raw("transpose:" + ps + ".latents", (size_t)Ld * c->spk_tokens);
  for (int i = 0; i < c->perc_depth; ++i) {
    const std::string a = ps + ".layers." + std::to_string(i) + ".0", f = ps + ".layers." + std::to_string(i) + ".1";
    conv(a + ".to_q.weight", inner, Ld, 1);
    conv(a + ".to_kv.weight", 2 * inner, Ld, 1);
    conv(a + ".to_out.weight", Ld, inner, 1);
    conv(f + ".0.weight", 2 * FI, Ld, 1);
    raw(f + ".0.bias", (size_t)2 * FI);
    conv(f + ".2.weight", Ld, FI, 1);
    raw(f + ".2.bias", Ld);
  }
  raw(ps + ".norm.gamma", Ld);
  raw("speaker_encoder.quantizer.project_in.weight", (size_t)c->fsq_dims * Ld);
  raw("speaker_encoder.quantizer.project_in.bias", c->fsq_dims);
  L.total = o;
  return L;
}

int conv_out_len(int n, int k, int s) { return n < k ? 0 : (n - k) / s + 1; }

}  // namespace

struct smi_enc {
  smi_enc_cfg cfg;
  EncLayout lay;
  const unsigned char* arena;
  std::map<std::string, float*> buf;        // named device buffers
  std::map<std::string, size_t> buf_floats;
  float *cbn, *c2;                          // normalised codebook, |c|^2
  int* lens_dev;                            // length slots
  std::vector<int32_t> host_lens;
  std::vector<Launch> prog;
  struct Stage { const float* ptr; int rows, cols, stride; };
  std::map<std::string, Stage> stages;      // debug views of the last forward
  int last_frames;
  hipEvent_t ev0, ev1;
  // One hipGraph per (n_samples, n_ref): the ~260 launches of an encode replayed as one graph launch.  The graph reads the
  // prompt from / leaves the ids in handle-owned buffers (in_wav, in_ref, out_sem, out_glob), so it does not depend on the
  // caller's pointers; the length slots are uploaded before every launch (they differ from key to key).
  struct Graph {
    hipGraphExec_t exec = nullptr;
    std::vector<int32_t> lens;
    std::vector<Launch> prog;
    std::map<std::string, Stage> stages;
    int frames = 0;
    unsigned long long used = 0;
    hipStream_t last = nullptr; bool launched = false;   // the stream of its last launch: synchronised before the exec is destroyed
  };
  std::map<std::pair<int, int>, Graph> graphs;
  std::map<std::pair<int, int>, int> seen;   // a shape is captured at its second occurrence: traffic whose shapes never repeat runs eagerly
  std::pair<int, int> prog_key{-1, -1};
  unsigned long long tick = 0;
  bool use_graph = true;
  hipStream_t gstream = nullptr;            // graph launches of callers on the null stream (which cannot be captured) run here
  hipEvent_t gev0 = nullptr, gev1 = nullptr;
};

namespace {

const float* ent(const smi_enc* h, const std::string& name) {
  for (const Entry& e : h->lay.e)
    if (e.name == name) return (const float*)(h->arena + e.offset);
  return nullptr;
}
bool ent_is_bf(const smi_enc* h, const std::string& name) {
  for (const Entry& e : h->lay.e)
    if (e.name == name) return e.kind == PACK_CONV_B || e.kind == PACK_CONVT_B;
  return false;
}

}  // namespace

extern "C" {

int smi_enc_arena_count(const smi_enc_cfg* cfg) {
  if (!enc_cfg_ok(cfg)) { smi_set_error("smi_enc_arena_count: invalid config"); return SMI_EINVAL; }
  return (int)enc_layout(cfg).e.size();
}

size_t smi_enc_arena_bytes(const smi_enc_cfg* cfg) {
  if (!enc_cfg_ok(cfg)) return 0;
  return enc_layout(cfg).total;
}

int smi_enc_arena_entry(const smi_enc_cfg* cfg, int index, char* name, int name_cap, size_t* offset, size_t* bytes, int32_t* info) {
  SMI_REQUIRE(enc_cfg_ok(cfg), "smi_enc_arena_entry: invalid config");
  EncLayout L = enc_layout(cfg);
  SMI_REQUIRE(index >= 0 && index < (int)L.e.size(), "smi_enc_arena_entry: index %d out of range", index);
  const Entry& e = L.e[index];
  if (name && name_cap > 0) {
    SMI_REQUIRE((int)e.name.size() < name_cap, "smi_enc_arena_entry: name buffer too small (%zu needed)", e.name.size() + 1);
    strncpy(name, e.name.c_str(), (size_t)name_cap - 1); name[name_cap - 1] = 0;
  }
  if (offset) *offset = e.offset;
  if (bytes) *bytes = e.bytes;
  if (info) { info[0] = e.kind; info[1] = e.Cout; info[2] = e.Cin; info[3] = e.K; info[4] = e.S; info[5] = e.pad; }
  return SMI_OK;
}

int smi_enc_create(const smi_enc_cfg* cfg, const void* arena_dev, size_t arena_bytes, smi_enc** out) {
  SMI_REQUIRE(out, "smi_enc_create: null out");
  *out = nullptr;
  SMI_REQUIRE(enc_cfg_ok(cfg), "smi_enc_create: invalid or unsupported config");
  SMI_REQUIRE(arena_dev, "smi_enc_create: null arena");
  char arch[128];
  int rc = smi_device_check(arch, sizeof(arch));
  if (rc) return rc;
  smi_enc* h = new smi_enc();
  h->cfg = *cfg;
  h->lay = enc_layout(cfg);
  h->arena = (const unsigned char*)arena_dev;
  h->cbn = h->c2 = nullptr; h->lens_dev = nullptr; h->ev0 = h->ev1 = nullptr; h->last_frames = 0;
  if (arena_bytes < h->lay.total) {
    smi_set_error("smi_enc_create: arena is %zu bytes, layout needs %zu", arena_bytes, h->lay.total);
    delete h;
    return SMI_EINVAL;
  }
  const smi_enc_cfg& c = h->cfg;
  // frame counts at the longest input
  int n = c.max_samples;
  std::vector<int> Ts;
  for (int i = 0; i < c.w2v_nconv; ++i) { n = conv_out_len(n, c.w2v_kernel[i], c.w2v_stride[i]); Ts.push_back(n); }
  const int T0 = Ts[0] + 8, T = Ts.back() + 8;
  const int Tm = c.max_ref_samples / c.hop_length + 1 + 8;
  const int nf = c.n_fft / 2 + 1;
  auto want = [&](const std::string& name, size_t floats) { h->buf_floats[name] = floats; };
  want("wavn", (size_t)c.max_samples + 64);
  want("cf0", (size_t)c.w2v_conv_dim * T0);
  want("cf1", (size_t)c.w2v_conv_dim * T0);
  const int Hh = c.w2v_hidden;
  const int wide = c.w2v_inter > 3 * Hh ? c.w2v_inter : 3 * Hh;
  want("h", (size_t)Hh * T);
  want("x", (size_t)Hh * T);
  want("wide", (size_t)wide * T);
  want("att", (size_t)Hh * T);
  want("acc", (size_t)Hh * T);
  want("feat", (size_t)Hh * T);
  want("dbg_hs0", (size_t)Hh * T);
  const int D = c.enc_dim;
  const int ew = c.enc_inter > c.enc_out ? c.enc_inter : c.enc_out;
  want("e0", (size_t)(D > c.codebook_dim ? D : c.codebook_dim) * T);
  want("e1", (size_t)D * T);
  want("e2", (size_t)D * T);
  want("ew", (size_t)ew * T);
  want("frames", (size_t)c.n_fft * Tm);
  want("dft", (size_t)2 * nf * Tm);
  want("mag", (size_t)nf * Tm);
  want("mel", (size_t)c.num_mels * Tm);
  const int C = c.ecapa_channels;
  want("ec_a", (size_t)C * Tm);
  want("ec_b", (size_t)C * Tm);
  want("ec_c", (size_t)C * Tm);
  want("ec_cat", (size_t)3 * C * Tm);
  want("ec_lat", (size_t)c.ecapa_out * Tm);
  want("ec_vec", (size_t)4 * (C + 128));
  const int Tc = c.spk_tokens + Tm;
  const int inner = c.perc_heads * 64;
  want("pctx", (size_t)c.spk_latent * Tc);
  want("pq", (size_t)inner * c.spk_tokens);
  want("pkv", (size_t)2 * inner * Tc);
  want("po", (size_t)inner * c.spk_tokens);
  want("pff", (size_t)2 * c.perc_ff_inner * c.spk_tokens);
  want("pg", (size_t)c.perc_ff_inner * c.spk_tokens);
  want("pout", (size_t)c.spk_latent * c.spk_tokens);
  want("fsqb", (size_t)c.spk_tokens * 8);
  want("in_wav", (size_t)c.max_samples + 64);
  want("in_ref", (size_t)c.max_ref_samples + 64);
  want("out_sem", (size_t)2 * T);           // int64 ids
  want("out_glob", (size_t)c.spk_tokens + 64);
  {
    const char* e = smi_env("SPARKMI_ENC_GRAPH");
    h->use_graph = !(e && e[0] == '0');
  }
  // more than the default dynamic LDS window for the attention / positional-conv kernels (per device, before any capture)
  (void)hipFuncSetAttribute((const void*)k_mha, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute((const void*)k_posconv, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  bool ok = true;
  for (auto& kv : h->buf_floats) {
    float* p = nullptr;
    if (hipMalloc((void**)&p, kv.second * 4) != hipSuccess) { ok = false; break; }
    h->buf[kv.first] = p;
  }
  ok = ok && hipMalloc((void**)&h->cbn, (size_t)c.codebook_size * c.codebook_dim * 4) == hipSuccess &&
       hipMalloc((void**)&h->c2, (size_t)c.codebook_size * 4) == hipSuccess &&
       hipMalloc((void**)&h->lens_dev, 64 * 4) == hipSuccess &&
       hipEventCreate(&h->ev0) == hipSuccess && hipEventCreate(&h->ev1) == hipSuccess &&
       hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking) == hipSuccess &&
       hipEventCreateWithFlags(&h->gev0, hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&h->gev1, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    smi_set_error("smi_enc_create: device allocation failed");
    smi_enc_destroy(h);
    return SMI_EHIP;
  }
  hipLaunchKernelGGL(k_cbnorm, dim3((c.codebook_size + 255) / 256), dim3(256), 0, 0, ent(h, "quantizer.codebook.weight"),
                     c.codebook_size, c.codebook_dim, h->cbn, h->c2);
  if (hipDeviceSynchronize() != hipSuccess) {
    smi_set_error("smi_enc_create: codebook normalisation failed");
    smi_enc_destroy(h);
    return SMI_EHIP;
  }
  *out = h;
  return SMI_OK;
}

int smi_enc_destroy(smi_enc* h) {
  if (!h) return SMI_OK;
  for (auto& kv : h->buf) if (kv.second) (void)hipFree(kv.second);
  if (h->cbn) (void)hipFree(h->cbn);
  if (h->c2) (void)hipFree(h->c2);
  if (h->lens_dev) (void)hipFree(h->lens_dev);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (auto& kv : h->graphs)
    if (kv.second.exec) {
      if (kv.second.launched) (void)hipStreamSynchronize(kv.second.last);   // an exec is never destroyed while a launch of it may be running
      (void)hipGraphExecDestroy(kv.second.exec);
    }
  if (h->gev0) (void)hipEventDestroy(h->gev0);
  if (h->gev1) (void)hipEventDestroy(h->gev1);
  if (h->gstream) (void)hipStreamDestroy(h->gstream);
  delete h;
  return SMI_OK;
}

}  // extern "C"

namespace {

// The launch sequence of one encode (h->prog), its length slots (h->host_lens) and debug views (h->stages); nothing runs here.
int enc_build(smi_enc* h, const float* wav_dev, int n_samples, const float* ref_dev, int n_ref, int64_t* sem_dev, int32_t* glob_dev,
              int* n_frames) {
  const smi_enc_cfg& c = h->cfg;
  std::vector<Launch>& P = h->prog;
  P.clear();
  h->stages.clear();
  // ---- length slots (device ints the conv / LN kernels read)
  std::vector<int32_t>& hl = h->host_lens;
  hl.clear();
  auto slot = [&](int v) -> const int* {
    for (size_t i = 0; i < hl.size(); ++i) if (hl[i] == v) return h->lens_dev + i;
    hl.push_back(v);
    return h->lens_dev + (hl.size() - 1);
  };
  std::vector<int> Ts;
  {
    int n = n_samples;
    for (int i = 0; i < c.w2v_nconv; ++i) { n = conv_out_len(n, c.w2v_kernel[i], c.w2v_stride[i]); Ts.push_back(n); }
  }
  const int T = Ts.back();
  SMI_REQUIRE(T >= 2, "smi_enc_forward: %d samples give %d frames", n_samples, T);
  SMI_REQUIRE(T <= 2040, "smi_enc_forward: %d frames exceed the attention kernel's 2040-key score buffer", T);
  const int Tm = n_ref / c.hop_length + 1;
  auto B = [&](const char* n) { return h->buf.at(n); };
  auto closure = [&](const std::string& name, double flops, std::function<void(hipStream_t)> fn) {
    Launch L; L.kind = 9; L.name = name; L.flops = flops; L.fn = std::move(fn);
    P.push_back(std::move(L));
  };
  auto stage = [&](const std::string& name, const float* p, int rows, int cols, int stride) { h->stages[name] = {p, rows, cols, stride}; };
  auto lnorm = [&](const std::string& name, const std::string& pfx, const float* dww, const float* dwb, const float* X, float* Y, int C,
                   int Tn, int stride, float eps, int gelu, int triple) {
    Launch L; L.kind = 1; L.name = name; L.flops = (dww ? 14.0 : 0.0) * C * Tn + 8.0 * C * Tn;
    LnP& p = L.lp; memset(&p, 0, sizeof(p));
    p.X = X; p.Y = Y; p.dww = dww; p.dwb = dwb; p.lens = slot(Tn); p.C = C; p.stride = stride; p.bs = 0; p.triple = triple;
    p.w = ent(h, pfx + ".weight"); p.bsh = ent(h, pfx + ".bias"); p.eps = eps; p.gelu = gelu;
    L.cpt = (C + 31) / 32; L.grid = dim3((Tn + 7) / 8, 1);
    P.push_back(L);
  };
  // Linear / Conv1d on [C][T] activations (B = 1).  Returns the launch for epilogue tweaks.
  auto conv = [&](const std::string& name, const std::string& wname, const std::string& bname, int Cout, int Cin, int K, int dil, int pad,
                  const float* X, int xstride, float* Y, const float* R, int ystride, int Tin, int Tout, int act, int istr) -> Launch& {
    P.push_back(make_conv_w(name, ent(h, wname), bname.empty() ? nullptr : ent(h, bname), Cout, Cin, K, dil, 1, pad, X, xstride, 0, Y,
                            nullptr, nullptr, R, ystride, 0, slot(Tin), 1, Tout, act, istr, Tin == Tout ? nullptr : slot(Tout),
                            ent_is_bf(h, wname)));
    return P.back();
  };
  auto mha = [&](const std::string& name, const float* Q, int qs, const float* K, int ks, const float* V, int vs, float* O, int os,
                 int heads, int Tq, int Tk) {
    MhaP m{Q, K, V, O, qs, ks, vs, os, Tq, Tk, 0.125f};
    const size_t lds = (size_t)(512 + 8 * Tk + 64 * 65) * 4;
    const dim3 grid((Tq + 7) / 8, heads);
    closure(name, 4.0 * heads * 64.0 * Tq * Tk, [=](hipStream_t s) {
      hipLaunchKernelGGL(k_mha, grid, dim3(256), lds, s, m);
    });
  };

  // ================= wav2vec2 =================
  float* wavn = B("wavn");
  closure("w2v.normalize", 4.0 * n_samples, [=](hipStream_t s) { hipLaunchKernelGGL(k_wavnorm, dim3(1), dim3(1024), 0, s, wav_dev, n_samples, wavn); });
  stage("input_values", wavn, 1, n_samples, n_samples);
  const int CD = c.w2v_conv_dim;
  float* cf[2] = {B("cf0"), B("cf1")};
  int cur = 0, Tc = Ts[0];
  {
    const float* W0 = ent(h, "w2v.feature_extractor.conv_layers.0.conv.weight");
    const float* b0 = ent(h, "w2v.feature_extractor.conv_layers.0.conv.bias");
    float* Y = cf[0];
    const int K0 = c.w2v_kernel[0], S0 = c.w2v_stride[0], T0 = Ts[0];
    closure("w2v.conv0", 2.0 * CD * K0 * T0, [=](hipStream_t s) {
      hipLaunchKernelGGL(k_conv0, dim3((T0 + 255) / 256, CD), dim3(256), 0, s, wavn, W0, b0, K0, S0, Y, CD, T0, T0);
    });
    lnorm("w2v.conv0.ln+gelu", "w2v.feature_extractor.conv_layers.0.layer_norm", nullptr, nullptr, cf[0], cf[1], CD, T0, T0, 1e-5f, 1, 0);
    cur = 1;
  }
  for (int i = 1; i < c.w2v_nconv; ++i) {
    const std::string p = "w2v.feature_extractor.conv_layers." + std::to_string(i);
    const int Tn = Ts[i];
    // conv reads cf[cur] (stride Tc), writes cf[1-cur] (stride Tn); LN + GELU back into cf[cur] with stride Tn
    conv(p + ".conv", p + ".conv.weight", p + ".conv.bias", CD, CD, c.w2v_kernel[i], 1, 0, cf[cur], Tc, cf[1 - cur], nullptr, Tn, Tc, Tn,
         ACT_NONE, c.w2v_stride[i]);
    lnorm(p + ".ln+gelu", p + ".layer_norm", nullptr, nullptr, cf[1 - cur], cf[cur], CD, Tn, Tn, 1e-5f, 1, 0);
    Tc = Tn;
  }
  stage("conv_feats", cf[cur], CD, T, T);
  const int Hd = c.w2v_hidden, I = c.w2v_inter;
  float *hbuf = B("h"), *x = B("x"), *wide = B("wide"), *att = B("att"), *acc = B("acc"), *feat = B("feat");
  lnorm("w2v.feature_projection.ln", "w2v.feature_projection.layer_norm", nullptr, nullptr, cf[cur], cf[1 - cur], CD, T, T, c.w2v_eps, 0, 0);
  conv("w2v.feature_projection.projection", "w2v.feature_projection.projection.weight", "w2v.feature_projection.projection.bias", Hd, CD, 1, 1,
       0, cf[1 - cur], T, x, nullptr, T, T, T, ACT_NONE, 1);
  {
    const float* Wp = ent(h, "w2v.encoder.pos_conv_embed.conv.weight");
    const float* bp = ent(h, "w2v.encoder.pos_conv_embed.conv.bias");
    const int Cg = Hd / c.w2v_pos_groups, K = c.w2v_pos_k;
    const size_t lds = (size_t)Cg * (64 + K - 1) * 4;
    closure("w2v.pos_conv+gelu+res", 2.0 * Hd * Cg * K * T, [=](hipStream_t s) {
      hipLaunchKernelGGL(k_posconv, dim3((T + 63) / 64, Hd / 16), dim3(256), lds, s, x, Wp, bp, hbuf, Hd, Cg, K, T, T);
    });
  }
  {   // test view of hidden_states[0] (the residual stream is updated in place by the layers)
    float* d0 = B("dbg_hs0");
    closure("w2v.hs0(dbg copy)", 0.0, [=](hipStream_t s) { hipLaunchKernelGGL(k_copy2d, dim3((T + 255) / 256, Hd), dim3(256), 0, s, hbuf, T, d0, T, Hd, T); });
    stage("hs0", d0, Hd, T, T);
  }
  const long long nHT = (long long)Hd * T;
  auto tap = [&](int idx) {
    // hidden_states[idx] is the residual stream before layer idx
    for (int k = 0; k < 3; ++k) {
      if (c.w2v_taps[k] != idx) continue;
      const int mode = k;
      closure("w2v.tap" + std::to_string(idx), 1.0 * nHT, [=](hipStream_t s) {
        hipLaunchKernelGGL(k_tap, dim3((unsigned)((nHT + 255) / 256)), dim3(256), 0, s, hbuf, acc, feat, nHT, mode);
      });
    }
  };
  SMI_REQUIRE(c.w2v_taps[0] < c.w2v_taps[1] && c.w2v_taps[1] < c.w2v_taps[2], "smi_enc_forward: hidden-state taps must be increasing");
  tap(0);
  for (int l = 0; l < c.w2v_layers; ++l) {
    const std::string p = "w2v.encoder.layers." + std::to_string(l);
    lnorm(p + ".ln1", p + ".layer_norm", nullptr, nullptr, hbuf, x, Hd, T, T, c.w2v_eps, 0, 0);
    conv(p + ".qkv", "cat:" + p + ".attention.q_proj.weight|" + p + ".attention.k_proj.weight|" + p + ".attention.v_proj.weight",
         "cat:" + p + ".attention.q_proj.bias|" + p + ".attention.k_proj.bias|" + p + ".attention.v_proj.bias", 3 * Hd, Hd, 1, 1, 0, x, T,
         wide, nullptr, T, T, T, ACT_NONE, 1);
    mha(p + ".attention", wide, T, wide + (size_t)Hd * T, T, wide + (size_t)2 * Hd * T, T, att, T, c.w2v_heads, T, T);
    conv(p + ".out_proj+res", p + ".attention.out_proj.weight", p + ".attention.out_proj.bias", Hd, Hd, 1, 1, 0, att, T, hbuf, hbuf, T, T, T,
         ACT_NONE, 1);
    lnorm(p + ".ln2", p + ".final_layer_norm", nullptr, nullptr, hbuf, x, Hd, T, T, c.w2v_eps, 0, 0);
    conv(p + ".ffn1+gelu", p + ".feed_forward.intermediate_dense.weight", p + ".feed_forward.intermediate_dense.bias", I, Hd, 1, 1, 0, x, T,
         wide, nullptr, T, T, T, ACT_GELU, 1);
    conv(p + ".ffn2+res", p + ".feed_forward.output_dense.weight", p + ".feed_forward.output_dense.bias", Hd, I, 1, 1, 0, wide, T, hbuf, hbuf,
         T, T, T, ACT_NONE, 1);
    tap(l + 1);
  }
  stage("feat", feat, Hd, T, T);

  // ================= BiCodec encoder + VQ (feat_encoder.py:76-87) =================
  {
    const int D = c.enc_dim, EI = c.enc_inter;
    float *e0 = B("e0"), *e1 = B("e1"), *e2 = B("e2"), *ew = B("ew");
    const float* in = feat;
    int cin = Hd;
    // Vocos backbone: embed conv7 -> LN -> nl x ConvNeXt -> final LN (vocos.py:324-335); result in e1 ([D][T])
    auto vocos = [&](const std::string& p, int nl, int triple_out) {
      conv(p + ".embed", p + ".embed.weight", p + ".embed.bias", D, cin, 7, 1, 3, in, T, e0, nullptr, T, T, T, ACT_NONE, 1);
      lnorm(p + ".norm", p + ".norm", nullptr, nullptr, e0, e2, D, T, T, 1e-6f, 0, 0);     // residual stream in e2
      for (int j = 0; j < nl; ++j) {
        const std::string b = p + ".convnext." + std::to_string(j);
        lnorm(b + ".dwconv+norm", b + ".norm", ent(h, b + ".dwconv.weight"), ent(h, b + ".dwconv.bias"), e2, e0, D, T, T, 1e-6f, 0, 0);
        conv(b + ".pwconv1", b + ".pwconv1.weight", b + ".pwconv1.bias", EI, D, 1, 1, 0, e0, T, ew, nullptr, T, T, T, ACT_GELU, 1);
        Launch& L = conv(b + ".pwconv2", b + ".pwconv2.weight", b + ".pwconv2.bias", D, EI, 1, 1, 0, ew, T, e2, e2, T, T, T, ACT_NONE, 1);
        L.cp.gamma = ent(h, b + ".gamma");
      }
      lnorm(p + ".final_layer_norm", p + ".final_layer_norm", nullptr, nullptr, e2, e1, D, T, T, 1e-6f, 0, triple_out);
      in = e1; cin = D;
    };
    vocos("encoder.encoder", c.enc_layers, c.enc_num_down > 0 ? 1 : 0);   // SamplingBlock(ratio 1) = 3x (samper.py:79-100)
    for (int i = 0; i < c.enc_num_down; ++i) vocos("encoder.downsample." + std::to_string(i) + ".1", 2, (i + 1 < c.enc_num_down) ? 1 : 0);
    conv("encoder.project", "encoder.project.weight", "encoder.project.bias", c.enc_out, D, 1, 1, 0, e1, T, ew, nullptr, T, T, T, ACT_NONE, 1);
    stage("z", ew, c.enc_out, T, T);
    conv("quantizer.in_project", "quantizer.in_project.weight", "quantizer.in_project.bias", c.codebook_dim, c.enc_out, 1, 1, 0, ew, T, e0,
         nullptr, T, T, T, ACT_NONE, 1);
    const float *cbn = h->cbn, *c2 = h->c2;
    const int Dc = c.codebook_dim, nc = c.codebook_size;
    closure("quantizer.argmax", 2.0 * T * nc * Dc, [=](hipStream_t s) { hipLaunchKernelGGL(k_vq, dim3(T), dim3(256), 0, s, e0, Dc, T, T, cbn, c2, nc, sem_dev); });
  }

  // ================= mel + ECAPA-TDNN latent + perceiver + FSQ =================
  {
    const int nf = c.n_fft / 2 + 1, nfft = c.n_fft, hop = c.hop_length;
    float *fr = B("frames"), *dft = B("dft"), *mag = B("mag"), *mel = B("mel");
    closure("mel.frames", 1.0 * nfft * Tm, [=](hipStream_t s) {
      hipLaunchKernelGGL(k_frames, dim3((Tm + 255) / 256, nfft), dim3(256), 0, s, ref_dev, n_ref, nfft, hop, fr, Tm, Tm);
    });
    conv("mel.dft", "mel.dft", "", 2 * nf, nfft, 1, 1, 0, fr, Tm, dft, nullptr, Tm, Tm, Tm, ACT_NONE, 1);
    closure("mel.magnitude", 4.0 * nf * Tm, [=](hipStream_t s) { hipLaunchKernelGGL(k_mag, dim3((Tm + 255) / 256, nf), dim3(256), 0, s, dft, nf, Tm, Tm, mag); });
    conv("mel.filterbank", "mel.fb", "", c.num_mels, nf, 1, 1, 0, mag, Tm, mel, nullptr, Tm, Tm, Tm, ACT_NONE, 1);
    stage("mel", mel, c.num_mels, Tm, Tm);
    // ---- ECAPA-TDNN (ecapa_tdnn.py:186-197): bn(relu(conv(x))) fused as ReLU + affine in the conv epilogue
    const int C = c.ecapa_channels, W = C / 8;
    const std::string se = "speaker_encoder.speaker_encoder";
    float *ea = B("ec_a"), *eb = B("ec_b"), *ec = B("ec_c"), *ecat = B("ec_cat"), *elat = B("ec_lat"), *evec = B("ec_vec");
    auto crb = [&](const std::string& name, const std::string& p, int co, int ci, int k, int dil, int pad, const float* X, const float* X2,
                   float* Y) {
      Launch& L = conv(name, p + ".conv.weight", p + ".conv.bias", co, ci, k, dil, pad, X, Tm, Y, nullptr, Tm, Tm, Tm, ACT_RELU, 1);
      L.cp.gamma = ent(h, "bnscale:" + p + ".bn"); L.cp.beta = ent(h, "bnshift:" + p + ".bn"); L.cp.X2 = X2;
    };
    crb(se + ".layer1", se + ".layer1", C, c.num_mels, 5, 1, 2, mel, nullptr, ea);
    const float* xin = ea;
    for (int li = 2; li <= 4; ++li) {
      const std::string b = se + ".layer" + std::to_string(li) + ".se_res2block";
      const int dil = li;
      float* y0 = eb;      // after the first 1x1
      float* y1 = ec;      // Res2 output (cat of the 8 branches)
      crb(b + ".0", b + ".0", C, C, 1, 1, 0, xin, nullptr, y0);
      for (int j = 0; j < 7; ++j) {
        const std::string cj = b + ".1.convs." + std::to_string(j);
        // sp = conv(out_{j-1} + spx[j]) -> relu -> bn  (ecapa_tdnn.py:50-58); branch j reads slice j of y0, writes slice j of y1
        Launch& L = conv(cj, cj + ".weight", cj + ".bias", W, W, 3, dil, dil, y0 + (size_t)j * W * Tm, Tm, y1 + (size_t)j * W * Tm, nullptr, Tm, Tm,
                         Tm, ACT_RELU, 1);
        L.cp.gamma = ent(h, "bnscale:" + b + ".1.bns." + std::to_string(j)); L.cp.beta = ent(h, "bnshift:" + b + ".1.bns." + std::to_string(j));
        if (j >= 1) L.cp.X2 = y1 + (size_t)(j - 1) * W * Tm;
      }
      {
        const float* src = y0 + (size_t)7 * W * Tm;
        float* dst = y1 + (size_t)7 * W * Tm;
        closure(b + ".1.passthrough", 0.0, [=](hipStream_t s) { hipLaunchKernelGGL(k_copy2d, dim3((Tm + 255) / 256, W), dim3(256), 0, s, src, Tm, dst, Tm, W, Tm); });
      }
      crb(b + ".2", b + ".2", C, C, 1, 1, 0, y1, nullptr, y0);
      float *mean = evec, *s1 = evec + C, *s2 = evec + C + 128;
      closure(b + ".3.mean", 1.0 * C * Tm, [=](hipStream_t s) { hipLaunchKernelGGL(k_rowmean, dim3((C + 3) / 4), dim3(256), 0, s, y0, C, Tm, Tm, mean); });
      const int* len1 = slot(1);
      P.push_back(make_conv_w(b + ".3.linear1", ent(h, b + ".3.linear1.weight"), ent(h, b + ".3.linear1.bias"), 128, C, 1, 1, 1, 0, mean, 1, C, s1,
                              nullptr, nullptr, nullptr, 1, 128, len1, 1, 1, ACT_RELU));
      P.back().gemv = true; P.back().grid = dim3(4, 1);
      P.push_back(make_conv_w(b + ".3.linear2", ent(h, b + ".3.linear2.weight"), ent(h, b + ".3.linear2.bias"), C, 128, 1, 1, 1, 0, s1, 1, 128, s2,
                              nullptr, nullptr, nullptr, 1, C, len1, 1, 1, ACT_SIGMOID));
      P.back().gemv = true; P.back().grid = dim3((C + 31) / 32, 1);
      float* outl = ecat + (size_t)(li - 2) * C * Tm;
      const float* xi = xin;
      closure(b + ".3.scale+res", 2.0 * C * Tm, [=](hipStream_t s) { hipLaunchKernelGGL(k_se, dim3((Tm + 255) / 256, C), dim3(256), 0, s, xi, y0, s2, outl, Tm, Tm, Tm); });
      xin = outl;
    }
    conv(se + ".conv+relu", se + ".conv.weight", se + ".conv.bias", c.ecapa_out, 3 * C, 1, 1, 0, ecat, Tm, elat, nullptr, Tm, Tm, Tm, ACT_RELU, 1);
    stage("ecapa_latent", elat, c.ecapa_out, Tm, Tm);
    // ---- perceiver resampler (perceiver_encoder.py:297-350): ctx buffer = [latents | projected context] along time
    const std::string ps = "speaker_encoder.perceiver_sampler";
    const int Ld = c.spk_latent, Nt = c.spk_tokens, Tk = Nt + Tm, inner = c.perc_heads * 64, FI = c.perc_ff_inner;
    float *ctx = B("pctx"), *pq = B("pq"), *pkv = B("pkv"), *po = B("po"), *pff = B("pff"), *pg = B("pg"), *pout = B("pout");
    {
      const float* lt = ent(h, "transpose:" + ps + ".latents");
      closure(ps + ".latents", 0.0, [=](hipStream_t s) { hipLaunchKernelGGL(k_copy2d, dim3(1, Ld), dim3(256), 0, s, lt, Nt, ctx, Tk, Ld, Nt); });
    }
    conv(ps + ".proj_context", ps + ".proj_context.weight", ps + ".proj_context.bias", Ld, c.ecapa_out, 1, 1, 0, elat, Tm, ctx + Nt, nullptr, Tk,
         Tm, Tm, ACT_NONE, 1);
    for (int i = 0; i < c.perc_depth; ++i) {
      const std::string a = ps + ".layers." + std::to_string(i) + ".0", f = ps + ".layers." + std::to_string(i) + ".1";
      conv(a + ".to_q", a + ".to_q.weight", "", inner, Ld, 1, 1, 0, ctx, Tk, pq, nullptr, Nt, Nt, Nt, ACT_NONE, 1);
      conv(a + ".to_kv", a + ".to_kv.weight", "", 2 * inner, Ld, 1, 1, 0, ctx, Tk, pkv, nullptr, Tk, Tk, Tk, ACT_NONE, 1);
      mha(a + ".attend", pq, Nt, pkv, Tk, pkv + (size_t)inner * Tk, Tk, po, Nt, c.perc_heads, Nt, Tk);
      conv(a + ".to_out+res", a + ".to_out.weight", "", Ld, inner, 1, 1, 0, po, Nt, ctx, ctx, Tk, Nt, Nt, ACT_NONE, 1);
      conv(f + ".0", f + ".0.weight", f + ".0.bias", 2 * FI, Ld, 1, 1, 0, ctx, Tk, pff, nullptr, Nt, Nt, Nt, ACT_NONE, 1);
      closure(f + ".geglu", 10.0 * FI * Nt, [=](hipStream_t s) { hipLaunchKernelGGL(k_geglu, dim3(1, FI), dim3(64 * ((Nt + 63) / 64)), 0, s, pff, FI, Nt, Nt, pg); });
      conv(f + ".2+res", f + ".2.weight", f + ".2.bias", Ld, FI, 1, 1, 0, pg, Nt, ctx, ctx, Tk, Nt, Nt, ACT_NONE, 1);
    }
    {
      const float* gm = ent(h, ps + ".norm.gamma");
      closure(ps + ".norm", 4.0 * Ld * Nt, [=](hipStream_t s) { hipLaunchKernelGGL(k_rmsn, dim3((Nt + 63) / 64), dim3(64), 0, s, ctx, Ld, Nt, Tk, gm, pout, Nt); });
    }
    stage("perceiver", pout, Ld, Nt, Nt);
    FsqQP q;
    memset(&q, 0, sizeof(q));
    q.X = pout; q.W = ent(h, "speaker_encoder.quantizer.project_in.weight"); q.b = ent(h, "speaker_encoder.quantizer.project_in.bias");
    q.out = glob_dev; q.bounded = B("fsqb"); q.latent = Ld; q.stride = Nt; q.Ntok = Nt; q.nd = c.fsq_dims;
    for (int j = 0; j < 8; ++j) q.levels[j] = j < c.fsq_dims ? c.fsq_levels[j] : 1;
    closure("speaker_encoder.quantizer", 2.0 * Nt * Ld * c.fsq_dims, [=](hipStream_t s) { hipLaunchKernelGGL(k_fsq_quant, dim3((Nt + 63) / 64), dim3(64), 0, s, q); });
    stage("fsq_bounded", B("fsqb"), Nt, c.fsq_dims, c.fsq_dims);
  }

  SMI_REQUIRE(hl.size() <= 64, "smi_enc_forward: too many distinct lengths");
  for (const Launch& L : P) {
    if (L.kind == 0) {
      SMI_REQUIRE(L.cp.W, "smi_enc_forward: arena entry for %s not found", L.name.c_str());
      SMI_REQUIRE(L.lds <= 64 * 1024, "smi_enc_forward: %s needs %zu bytes of LDS", L.name.c_str(), L.lds);
      SMI_REQUIRE(L.cp.xw <= 192 && (L.chg == 1 || L.cp.xw <= 64), "smi_enc_forward: %s stages %d columns", L.name.c_str(), L.cp.xw);
    }
    if (L.kind == 1) SMI_REQUIRE(L.lp.w && L.lp.bsh && L.cpt <= 32, "smi_enc_forward: LayerNorm %s: missing weights or too many channels", L.name.c_str());
  }
  *n_frames = T;
  return SMI_OK;
}

int enc_run(const std::vector<Launch>& P, hipStream_t st) {
  for (const Launch& L : P) {
    int rc = run_launch(L, st);
    if (rc) return rc;
  }
  return SMI_OK;
}

}  // namespace

extern "C" {

int smi_enc_forward(smi_enc* h, const float* wav_dev, int n_samples, const float* ref_dev, int n_ref, int64_t* sem_dev,
                    int32_t* glob_dev, int* n_frames, void* stream) {
  SMI_REQUIRE(h && wav_dev && ref_dev && sem_dev && glob_dev && n_frames, "smi_enc_forward: null argument");
  const smi_enc_cfg& c = h->cfg;
  SMI_REQUIRE(n_samples >= 400 && n_samples <= c.max_samples, "smi_enc_forward: n_samples=%d outside 400..%d", n_samples, c.max_samples);
  SMI_REQUIRE(n_ref > c.n_fft / 2 && n_ref <= c.max_ref_samples, "smi_enc_forward: n_ref=%d outside %d..%d", n_ref, c.n_fft / 2 + 1,
              c.max_ref_samples);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const std::pair<int, int> key(n_samples, n_ref);
  bool eager = !h->use_graph;
  if (!eager && h->graphs.find(key) == h->graphs.end()) {
    if (h->seen.size() > 4096) h->seen.clear();
    eager = h->seen[key]++ == 0;
  }
  if (eager) {
    if ((rc = enc_build(h, wav_dev, n_samples, ref_dev, n_ref, sem_dev, glob_dev, n_frames))) return rc;
    h->prog_key = {-1, -1};
    SMI_HIP(hipMemcpyAsync(h->lens_dev, h->host_lens.data(), h->host_lens.size() * 4, hipMemcpyHostToDevice, st));
    if ((rc = enc_run(h->prog, st))) return rc;
    h->last_frames = *n_frames;
    return SMI_OK;
  }
  float *in_wav = h->buf.at("in_wav"), *in_ref = h->buf.at("in_ref");
  int64_t* out_sem = (int64_t*)h->buf.at("out_sem");
  int32_t* out_glob = (int32_t*)h->buf.at("out_glob");
  // the null stream cannot be captured: such callers' encodes run on the handle's own stream, fenced by events on both sides
  hipStream_t run = st ? st : h->gstream;
  if (!st) {
    SMI_HIP(hipEventRecord(h->gev0, st));
    SMI_HIP(hipStreamWaitEvent(run, h->gev0, 0));
  }
  SMI_HIP(hipMemcpyAsync(in_wav, wav_dev, (size_t)n_samples * 4, hipMemcpyDeviceToDevice, run));
  SMI_HIP(hipMemcpyAsync(in_ref, ref_dev, (size_t)n_ref * 4, hipMemcpyDeviceToDevice, run));
  auto it = h->graphs.find(key);
  if (it == h->graphs.end()) {
    if (h->graphs.size() >= 8) {   // keep the eight most recently used shapes
      auto old = h->graphs.begin();
      for (auto j = h->graphs.begin(); j != h->graphs.end(); ++j) if (j->second.used < old->second.used) old = j;
      SMI_HIP(hipStreamSynchronize(run));
      if (old->second.launched && old->second.last != run) SMI_HIP(hipStreamSynchronize(old->second.last));   // it may have last run on another caller's stream
      if (old->second.exec) (void)hipGraphExecDestroy(old->second.exec);
      h->graphs.erase(old);
    }
    int T = 0;
    if ((rc = enc_build(h, in_wav, n_samples, in_ref, n_ref, out_sem, out_glob, &T))) return rc;
    smi_enc::Graph g;
    g.lens = h->host_lens; g.prog = h->prog; g.stages = h->stages; g.frames = T;
    SMI_HIP(hipMemcpyAsync(h->lens_dev, g.lens.data(), g.lens.size() * 4, hipMemcpyHostToDevice, run));
    SMI_HIP(hipStreamSynchronize(run));      // (the pageable source above must be read before g.lens moves into the map)
    SMI_HIP(hipStreamBeginCapture(run, hipStreamCaptureModeRelaxed));
    rc = enc_run(g.prog, run);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(run, &graph);
    if (rc || ec != hipSuccess || !graph) {
      if (graph) (void)hipGraphDestroy(graph);
      if (!rc) { smi_set_error("smi_enc_forward: stream capture failed: %s", hipGetErrorString(ec)); rc = SMI_EHIP; }
      return rc;
    }
    const hipError_t ei = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { smi_set_error("smi_enc_forward: hipGraphInstantiate: %s", hipGetErrorString(ei)); return SMI_EHIP; }
    h->prog_key = key;
    it = h->graphs.emplace(key, std::move(g)).first;
  } else {
    if (h->prog_key != key) { h->prog = it->second.prog; h->prog_key = key; }
    h->stages = it->second.stages;
    SMI_HIP(hipMemcpyAsync(h->lens_dev, it->second.lens.data(), it->second.lens.size() * 4, hipMemcpyHostToDevice, run));
  }
  it->second.used = ++h->tick;
  SMI_HIP(hipGraphLaunch(it->second.exec, run));
  it->second.last = run; it->second.launched = true;
  const int T = it->second.frames;
  SMI_HIP(hipMemcpyAsync(sem_dev, out_sem, (size_t)T * 8, hipMemcpyDeviceToDevice, run));
  SMI_HIP(hipMemcpyAsync(glob_dev, out_glob, (size_t)c.spk_tokens * 4, hipMemcpyDeviceToDevice, run));
  if (!st) {
    SMI_HIP(hipEventRecord(h->gev1, run));
    SMI_HIP(hipStreamWaitEvent(st, h->gev1, 0));
  }
  h->last_frames = T;
  *n_frames = T;
  return SMI_OK;
}

int smi_enc_debug_stage(smi_enc* h, const char* name, float* out_dev, size_t max_floats, int32_t* dims, void* stream) {
  SMI_REQUIRE(h && name && out_dev && dims, "smi_enc_debug_stage: null argument");
  SMI_REQUIRE(h->last_frames > 0, "smi_enc_debug_stage: no forward has run");
  auto it = h->stages.find(name);
  SMI_REQUIRE(it != h->stages.end(), "smi_enc_debug_stage: unknown stage '%s'", name);
  const smi_enc::Stage& s = it->second;
  SMI_REQUIRE((size_t)s.rows * s.cols <= max_floats, "smi_enc_debug_stage: output buffer too small");
  SMI_HIP(hipMemcpy2DAsync(out_dev, (size_t)s.cols * 4, s.ptr, (size_t)s.stride * 4, (size_t)s.cols * 4, s.rows, hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
  dims[0] = s.rows; dims[1] = s.cols;
  return SMI_OK;
}

int smi_enc_num_launches(smi_enc* h) { return h ? (int)h->prog.size() : 0; }

int smi_enc_time_launch(smi_enc* h, int index, int iters, float* ms_avg, double* flops, char* name, int name_cap, void* stream) {
  SMI_REQUIRE(h && ms_avg && iters > 0, "smi_enc_time_launch: bad argument");
  SMI_REQUIRE(index >= 0 && index < (int)h->prog.size(), "smi_enc_time_launch: index %d out of range", index);
  hipStream_t st = (hipStream_t)stream;
  const Launch& L = h->prog[index];
  int rc = run_launch(L, st);
  if (rc) return rc;
  SMI_HIP(hipEventRecord(h->ev0, st));
  for (int i = 0; i < iters; ++i)
    if ((rc = run_launch(L, st))) return rc;
  SMI_HIP(hipEventRecord(h->ev1, st));
  SMI_HIP(hipEventSynchronize(h->ev1));
  float ms = 0.f;
  SMI_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_avg = ms / iters;
  if (flops) *flops = L.flops;
  if (name && name_cap > 0) { strncpy(name, L.name.c_str(), (size_t)name_cap - 1); name[name_cap - 1] = 0; }
  return SMI_OK;
}

}  // extern "C"
