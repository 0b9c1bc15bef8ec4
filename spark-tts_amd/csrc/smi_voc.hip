// smi_voc.hip -- BiCodec vocoder (detokenize path) for gfx950 (MI355X).
//
// Stands behind BiCodec.detokenize (sparktts/models/bicodec.py:171-189):
//   quantizer.detokenize        sparktts/modules/vq/factorized_vector_quantize.py:154-167
//   speaker_encoder.detokenize  sparktts/modules/speaker/speaker_encoder.py:107-112,
//                               fsq/residual_fsq.py:112-199, fsq/finite_scalar_quantization.py:143-162
//   prenet (Decoder)            sparktts/modules/encoder_decoder/feat_decoder.py:78-94,
//                               blocks/vocos.py:65-110,324-335, blocks/samper.py:79-100 (ratio 1 => 3x)
//   decoder (WaveGenerator)     sparktts/modules/encoder_decoder/wave_generator.py:29-88,
//                               blocks/layers.py:33-67 (snake, ResidualUnit)
//
// Every dense contraction (Linear, Conv1d incl. dilated, ConvTranspose1d as polyphase taps) runs
// through ONE implicit-GEMM kernel on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32: the
// reference computes in fp32 and the parity bar is 1e-3 on the waveform).  Activations are
// [B][C][T] fp32 with time contiguous; input tiles (+halo) are staged through LDS once per
// 32-channel chunk and re-read per tap; weights are pre-packed in MFMA A-operand order so each
// lane streams float4s.  Elementwise work is fused into the producing kernel's epilogue: bias,
// per-utterance bias (d-vector), layer-scale, residual add, GELU / tanh, and the *next* layer's
// Snake activation (written as a second output), so no standalone elementwise kernel exists.
// Ragged batches: every kernel masks loads beyond the row's own length, so each row equals an
// un-padded B=1 run of that row.
#include "smi_net.h"

namespace {

struct VocLayout {
  std::vector<Entry> e;
  size_t total;
};

bool voc_cfg_ok(const smi_voc_cfg* c) {
  if (!c) return false;
  if (c->vq_input_dim <= 0 || c->codebook_size <= 0 || c->codebook_dim <= 0) return false;
  if (c->fsq_dims < 1 || c->fsq_dims > 8 || c->spk_token_num < 1 || c->spk_latent_dim < 1) return false;
  if (c->pre_dim < 1 || c->pre_dim > 512 || c->pre_layers < 1 || c->pre_num_down < 0 || c->pre_num_down > 4) return false;
  if (c->pre_out_channels != c->dec_in || c->spk_out_dim != c->dec_in || c->pre_input_channels != c->vq_input_dim) return false;
  if (c->pre_cond_dim != c->spk_out_dim) return false;
  if (c->dec_nblocks < 1 || c->dec_nblocks > 8 || c->dec_channels >> c->dec_nblocks < 1) return false;
  for (int i = 0; i < c->dec_nblocks; ++i) {
    const int k = c->dec_ksizes[i], s = c->dec_rates[i];
    if (s < 1 || s > kMaxPhases || k < s || (k - s) % 2 || (k + s - 1) / s > kMaxTaps) return false;
  }
  if (c->max_batch < 1 || c->max_frames < 1) return false;
  return true;
}

VocLayout voc_layout(const smi_voc_cfg* c) {
  VocLayout L;
  size_t o = 0;
  auto add = [&](const std::string& name, int kind, int Cout, int Cin, int K, int S, int pad, size_t floats) {
    Entry e{name, kind, Cout, Cin, K, S, pad, o, floats * 4};
    L.e.push_back(e);
    o += smi_align_up(floats * 4, 256);
  };
  auto raw = [&](const std::string& name, size_t n) { add(name, PACK_RAW, 0, 0, 0, 1, 0, n); };
  // conv: layers of the dense stack -- on the bf16-split matrix pipe (k_convb, two bf16 weight planes) unless the handle is
  // created with exact_fp32, or the layer is too thin to fill a 16-channel step / 32-row tile; convf: always exact fp32
  // (the per-utterance GEMVs read the fp32 packing; the 8-channel codebook projection; the one-channel output conv)
  auto convf = [&](const std::string& name, int Cout, int Cin, int K) {
    add(name, PACK_CONV, Cout, Cin, K, 1, 0, (size_t)conv_geom(Cout, Cin, K, 1, 0, 1).floats);
  };
  auto use_bf = [&](int Cout, int Cin) { return !c->exact_fp32 && Cin >= 32 && Cout >= 32; };
  auto conv = [&](const std::string& name, int Cout, int Cin, int K) {
    const bool bf = use_bf(Cout, Cin);
    add(name, bf ? PACK_CONV_B : PACK_CONV, Cout, Cin, K, 1, 0, (size_t)conv_geom(Cout, Cin, K, 1, 0, 1, bf).floats);
  };
  const int D = c->pre_dim, I = c->pre_inter, Cc = c->pre_cond_dim;
  raw("quantizer.codebook.weight", (size_t)c->codebook_size * c->codebook_dim);
  convf("quantizer.out_project.weight", c->vq_input_dim, c->codebook_dim, 1);
  raw("quantizer.out_project.bias", c->vq_input_dim);
  raw("speaker_encoder.quantizer.project_out.weight", (size_t)c->spk_latent_dim * c->fsq_dims);
  raw("speaker_encoder.quantizer.project_out.bias", c->spk_latent_dim);
  convf("speaker_encoder.project.weight", c->spk_out_dim, c->spk_latent_dim * c->spk_token_num, 1);
  raw("speaker_encoder.project.bias", c->spk_out_dim);
  conv("prenet.linear_pre.weight", D, c->pre_input_channels, 1);
  raw("prenet.linear_pre.bias", D);
  std::string adaw = "cat:", adab = "cat:";
  auto vocos = [&](const std::string& p, int nl, bool ada) {
    conv(p + ".embed.weight", D, D, 7);
    raw(p + ".embed.bias", D);
    auto norm = [&](const std::string& n) {
      if (ada) {
        adaw += (adaw.size() > 4 ? "|" : "") + n + ".scale.weight|" + n + ".shift.weight";
        adab += (adab.size() > 4 ? "|" : "") + n + ".scale.bias|" + n + ".shift.bias";
      } else {
        raw(n + ".weight", D);
        raw(n + ".bias", D);
      }
    };
    norm(p + ".norm");
    for (int j = 0; j < nl; ++j) {
      const std::string b = p + ".convnext." + std::to_string(j);
      raw(b + ".dwconv.weight", (size_t)D * 7);
      raw(b + ".dwconv.bias", D);
      norm(b + ".norm");
      conv(b + ".pwconv1.weight", I, D, 1);
      raw(b + ".pwconv1.bias", I);
      conv(b + ".pwconv2.weight", D, I, 1);
      raw(b + ".pwconv2.bias", D);
      raw(b + ".gamma", D);
    }
    raw(p + ".final_layer_norm.weight", D);
    raw(p + ".final_layer_norm.bias", D);
  };
  for (int i = 0; i < c->pre_num_down; ++i) vocos("prenet.downsample." + std::to_string(i) + ".1", 2, false);
  vocos("prenet.vocos_backbone", c->pre_layers, true);
  const int nada = 1 + c->pre_layers;
  convf(adaw, nada * 2 * D, Cc, 1);
  raw(adab, (size_t)nada * 2 * D);
  conv("prenet.linear.weight", c->pre_out_channels, D, 1);
  raw("prenet.linear.bias", c->pre_out_channels);
  int ch = c->dec_channels;
  conv("decoder.model.0.weight", ch, c->dec_in, 7);
  raw("decoder.model.0.bias", ch);
  for (int i = 0; i < c->dec_nblocks; ++i) {
    const int cin = ch >> i, cout = ch >> (i + 1), k = c->dec_ksizes[i], s = c->dec_rates[i];
    const std::string b = "decoder.model." + std::to_string(i + 1) + ".block";
    raw(b + ".0.alpha", cin);
    add(b + ".1.weight", use_bf(cout, cin) ? PACK_CONVT_B : PACK_CONVT, cout, cin, k, s, (k - s) / 2,
        (size_t)conv_geom(cout, cin, k, 1, (k - s) / 2, s, use_bf(cout, cin)).floats);
    raw(b + ".1.bias", cout);
    for (int r = 0; r < 3; ++r) {
      const std::string u = b + "." + std::to_string(r + 2) + ".block";
      raw(u + ".0.alpha", cout);
      conv(u + ".1.weight", cout, cout, 7);
      raw(u + ".1.bias", cout);
      raw(u + ".2.alpha", cout);
      conv(u + ".3.weight", cout, cout, 1);
      raw(u + ".3.bias", cout);
    }
  }
  const int clast = ch >> c->dec_nblocks;
  raw("decoder.model." + std::to_string(c->dec_nblocks + 1) + ".alpha", clast);
  convf("decoder.model." + std::to_string(c->dec_nblocks + 2) + ".weight", 1, clast, 7);
  raw("decoder.model." + std::to_string(c->dec_nblocks + 2) + ".bias", 1);
  L.total = o;
  return L;
}

}  // namespace

struct smi_voc {
  smi_voc_cfg cfg;
  VocLayout lay;
  const unsigned char* arena;
  float* buf[4];        // activation ping-pong buffers [max_batch][maxC x maxL]
  size_t buf_floats;    // per batch item
  float* small;         // d-vector path + AdaLN params
  int* lens_dev;        // [8][max_batch] valid lengths per resolution
  float* dbg[8]; size_t dbg_floats[8]; int debug;
  std::vector<Launch> prog;
  std::vector<int32_t> host_lens;   // staging for the async H2D copy (must outlive the call)
  int lastB, lastT;
  hipEvent_t ev0, ev1;
};

namespace {

// Where a group of launches finds its packed weights: a table of entries (the vocoder's arena, or one block's -- smi_voc_block_*)
struct WSrc {
  const std::vector<Entry>* e;
  const unsigned char* arena;
  const float* get(const std::string& name) const {
    for (const Entry& x : *e)
      if (x.name == name) return (const float*)(arena + x.offset);
    return nullptr;
  }
  bool is_bf(const std::string& name) const {
    for (const Entry& x : *e)
      if (x.name == name) return x.kind == PACK_CONV_B || x.kind == PACK_CONVT_B;
    return false;
  }
};
WSrc wsrc(const smi_voc* h) { return WSrc{&h->lay.e, h->arena}; }
const float* ent(const smi_voc* h, const std::string& name) { return wsrc(h).get(name); }

Launch make_conv(const WSrc& w, const std::string& name, const std::string& wname, const char* bname,
                 int Cout, int Cin, int K, int dil, int S, int pad, const float* X, int xstride, long long xb,
                 float* Y, float* Ys, const float* alpha, const float* R, int ystride, long long yb,
                 const int* lens, int B, int Lmax, int act) {
  return make_conv_w(name, w.get(wname), bname ? w.get(bname) : nullptr, Cout, Cin, K, dil, S, pad, X, xstride, xb, Y, Ys,
                     alpha, R, ystride, yb, lens, B, Lmax, act, 1, nullptr, w.is_bf(wname));
}
Launch make_conv(const smi_voc* h, const std::string& name, const std::string& wname, const char* bname,
                 int Cout, int Cin, int K, int dil, int S, int pad, const float* X, int xstride, long long xb,
                 float* Y, float* Ys, const float* alpha, const float* R, int ystride, long long yb,
                 const int* lens, int B, int Lmax, int act) {
  return make_conv(wsrc(h), name, wname, bname, Cout, Cin, K, dil, S, pad, X, xstride, xb, Y, Ys, alpha, R, ystride, yb, lens, B, Lmax, act);
}

// ---- launch builders shared by smi_voc_forward and the one-block entry points (smi_voc_block_run): the same kernels,
//      the same launch geometry, the same fusions

// ResidualUnit (blocks/layers.py:51-67): y = x + conv1(snake(conv7_dil(snake(x)))).  US holds snake(U, <u>.0.alpha) -- the
// producer's second output; conv7's epilogue applies the unit's second Snake (-> A); the 1x1 adds the residual U and writes
// the new U to `rawout` (may be null) and, if `want_us`, snake(new U, next_alpha) over US (in place: the 7-tap conv has finished
// with it).  The fused form (k_resunit, below) cannot write over US -- a block's neighbours still read their halo columns from it --
// and leaves the Snake'd output in A, which it does not otherwise need.  Returns the buffer that holds it (null: none asked for).
float* add_res_unit(std::vector<Launch>& P, const WSrc& w, const std::string& u, int C, int dil, const float* U, float* US,
                    float* A, float* rawout, bool want_us, const float* next_alpha, int L, long long bs, const int* lens, int B) {
  float* US_out = want_us ? US : nullptr;
  P.push_back(make_conv(w, u + ".conv7", u + ".1.weight", (u + ".1.bias").c_str(), C, C, 7, dil, 1, 3 * dil, US, L, bs,
                        nullptr, A, w.get(u + ".2.alpha"), nullptr, L, bs, lens, B, L, ACT_NONE));
  P.push_back(make_conv(w, u + ".conv1+res", u + ".3.weight", (u + ".3.bias").c_str(), C, C, 1, 1, 1, 0, A, L, bs,
                        rawout, US_out, next_alpha, U, L, bs, lens, B, L, ACT_NONE));
  // C = 96 / 192 on the bf16-split pipe with one output tile per wave (not the channel-split mode of short sequences): the two
  // launches become one k_resunit (SPARKMI_RESFUSE=0: two launches, A/B).  The choice depends on the layer and on whether the
  // grid fills the chip, as every launch plan does -- never on a row's neighbours.
  const Launch& c7 = P[P.size() - 2];
  const Launch& c1 = P[P.size() - 1];
  const char* e = smi_env("SPARKMI_RESFUSE");
  if ((C == 96 || C == 192) && c7.bf && c1.bf && !c7.ks && !c1.ks && c7.cp.W && c1.cp.W && c7.cp.bias && c1.cp.bias && c7.cp.alpha &&
      c7.cp.xw <= 128 && !(e && e[0] == '0')) {
    Launch F; F.kind = 5; F.name = u + ".conv7+conv1+res"; F.flops = c7.flops + c1.flops; F.res_nwv = C / 32;
    ResP& r = F.rp; memset(&r, 0, sizeof(r));
    r.Xs = US; r.U = U; r.W7 = c7.cp.W; r.b7 = c7.cp.bias; r.alpha2 = c7.cp.alpha; r.W1 = c1.cp.W; r.b1 = c1.cp.bias;
    r.alpha_next = want_us ? next_alpha : nullptr; r.Y = rawout; r.Ys = want_us ? A : nullptr; r.lens = lens; r.C = C; r.stride = L; r.bs = bs;
    r.halo_l = c7.cp.halo_l; r.xw = c7.cp.xw; r.fast_sin = c7.cp.fast_sin;
    for (int i = 0; i < 7; ++i) r.off[i] = c7.cp.off[0][i];
    F.grid = dim3((L + 63) / 64, 1, B);
    const size_t stage = (size_t)2 * 6 * r.xw * 16, image = (size_t)2 * (C / 8) * 64 * 16;
    F.lds = stage > image ? stage : image;
    P.pop_back(); P.pop_back();
    P.push_back(F);
    return F.rp.Ys;
  }
  return US_out;
}

// DecoderBlock (encoder_decoder/wave_generator.py:29-53) on s_in = snake(x, <b>.0.alpha) (left by the producer):
// ConvTranspose1d (polyphase) -> U raw, US = snake(U, unit 2's first alpha); three ResidualUnits (dilation 1, 3, 9).
// The last unit writes its raw output to `raw_final` (may be null) and snake(., next_alpha) (next_alpha may be null) to US or A:
// the function returns which (a fused unit flips the two; see add_res_unit).
float* add_dec_block(std::vector<Launch>& P, const WSrc& w, const std::string& b, int cin, int cout, int k, int s, const float* s_in,
                   int Lin, long long bs_in, float* U, float* US, float* A, float* raw_final, const float* next_alpha, long long bs,
                   const int* lens_in, const int* lens_out, int B) {
  const int Lout = Lin * s;
  P.push_back(make_conv(w, b + ".convT", b + ".1.weight", (b + ".1.bias").c_str(), cout, cin, k, 1, s, (k - s) / 2, s_in, Lin, bs_in,
                        U, US, w.get(b + ".2.block.0.alpha"), nullptr, Lout, bs, lens_in, B, Lin, ACT_NONE));
  for (int r = 0; r < 3; ++r) {
    const std::string u = b + "." + std::to_string(r + 2) + ".block";
    const int dil = r == 0 ? 1 : (r == 1 ? 3 : 9);
    const bool last = r == 2;
    const float* na = last ? next_alpha : w.get(b + "." + std::to_string(r + 3) + ".block.0.alpha");
    float* out = add_res_unit(P, w, u, cout, dil, U, US, A, last ? raw_final : U, !(last && !next_alpha), na, Lout, bs, lens_out, B);
    if (out && out != US) { A = US; US = out; }   // the Snake'd stream now lives in the other buffer; the old one is the next unit's scratch
  }
  return US;
}

// LayerNorm / AdaLayerNorm over channels, optionally behind the depthwise conv7 of a ConvNeXt block (k_dwln)
void add_lnorm(std::vector<Launch>& P, const WSrc& w, const std::string& name, const std::string& pfx, const float* ada, int ada_stride,
               const float* dww, const float* dwb, const float* X, float* Y, int triple, int D, int T, long long bs, const int* lens, int B) {
  Launch L; L.kind = 1; L.name = name; L.flops = (dww ? 14.0 : 0.0) * D * T * B + 8.0 * D * T * B;
  LnP& p = L.lp; memset(&p, 0, sizeof(p));
  p.X = X; p.Y = Y; p.dww = dww; p.dwb = dwb; p.lens = lens; p.C = D; p.stride = T; p.bs = bs; p.triple = triple;
  p.eps = 1e-6f;
  if (ada) { p.ada = ada; p.ada_stride = ada_stride; }
  else { p.w = w.get(pfx + ".weight"); p.bsh = w.get(pfx + ".bias"); }
  L.cpt = (D + 31) / 32; L.grid = dim3((T + 7) / 8, B);
  P.push_back(L);
}

// ConvNeXtBlock (blocks/vocos.py:26-62): x += gamma * pwconv2(GELU(pwconv1(norm(dwconv7(x))))); n, m: scratch
void add_convnext(std::vector<Launch>& P, const WSrc& w, const std::string& b, int D, int I, float* x, float* n, float* m,
                  const float* ada, int ada_stride, int T, long long bs, const int* lens, int B) {
  add_lnorm(P, w, b + ".dwconv+norm", b + ".norm", ada, ada_stride, w.get(b + ".dwconv.weight"), w.get(b + ".dwconv.bias"), x, n, 0, D, T, bs, lens, B);
  P.push_back(make_conv(w, b + ".pwconv1", b + ".pwconv1.weight", (b + ".pwconv1.bias").c_str(), I, D, 1, 1, 1, 0, n, T, bs, m,
                        nullptr, nullptr, nullptr, T, bs, lens, B, T, ACT_GELU));
  P.push_back(make_conv(w, b + ".pwconv2", b + ".pwconv2.weight", (b + ".pwconv2.bias").c_str(), D, I, 1, 1, 1, 0, m, T, bs, x,
                        nullptr, nullptr, x, T, bs, lens, B, T, ACT_NONE));
  P.back().cp.gamma = w.get(b + ".gamma");
}

// what every conv launch must satisfy before it runs (the kernels' staging limits)
int check_launches(const std::vector<Launch>& P, const char* who) {
  for (const Launch& L : P) {
    if (L.kind == 0) {
      SMI_REQUIRE(L.cp.W, "%s: arena entry for %s not found", who, L.name.c_str());
      SMI_REQUIRE(L.lds <= 64 * 1024, "%s: %s needs %zu bytes of LDS", who, L.name.c_str(), L.lds);
      SMI_REQUIRE(L.cp.xw <= 128 && (L.chg != 4 || L.cp.xw <= 64), "%s: %s stages %d columns", who, L.name.c_str(), L.cp.xw);
      if (L.tph) SMI_REQUIRE(!L.cp.bbias && !L.cp.gamma && !L.cp.beta && !L.cp.R && !L.cp.X2 && L.cp.out_scale == 1.0f && L.cp.act == ACT_NONE,
                             "%s: %s: the multi-phase transposed conv has a bias / Snake epilogue only", who, L.name.c_str());
    }
  }
  return SMI_OK;
}

// ---- one reference block as its own little arena (smi_voc_block_*: op-level tests on the reference's layer classes)
bool block_cfg_ok(const smi_voc_block_cfg* c) {
  if (!c) return false;
  switch (c->kind) {
    case SMI_VOC_BLOCK_RESUNIT: return c->C >= 1 && c->C <= 4096 && (c->dil == 1 || c->dil == 3 || c->dil == 9);
    case SMI_VOC_BLOCK_DECBLOCK:
      return c->C >= 1 && c->Cout >= 1 && c->C <= 4096 && c->Cout <= 4096 && c->S >= 1 && c->S <= kMaxPhases && c->K >= c->S &&
             (c->K - c->S) % 2 == 0 && (c->K + c->S - 1) / c->S <= kMaxTaps;
    case SMI_VOC_BLOCK_CONVNEXT: return c->C >= 1 && c->C <= 512 && c->I >= 1 && c->I <= 8192 && c->cond_dim >= 0 && c->cond_dim <= 4096;
  }
  return false;
}

VocLayout block_layout(const smi_voc_block_cfg* c) {
  VocLayout L;
  size_t o = 0;
  auto add = [&](const std::string& name, int kind, int Cout, int Cin, int K, int S, int pad, size_t floats) {
    Entry e{name, kind, Cout, Cin, K, S, pad, o, floats * 4};
    L.e.push_back(e);
    o += smi_align_up(floats * 4, 256);
  };
  auto raw = [&](const std::string& name, size_t n) { add(name, PACK_RAW, 0, 0, 0, 1, 0, n); };
  auto use_bf = [&](int Cout, int Cin) { return !c->exact_fp32 && Cin >= 32 && Cout >= 32; };   // the vocoder's own rule (voc_layout)
  auto conv = [&](const std::string& name, int Cout, int Cin, int K) {
    const bool bf = use_bf(Cout, Cin);
    add(name, bf ? PACK_CONV_B : PACK_CONV, Cout, Cin, K, 1, 0, (size_t)conv_geom(Cout, Cin, K, 1, 0, 1, bf).floats);
  };
  auto unit = [&](const std::string& u, int C) {
    raw(u + ".0.alpha", C);
    conv(u + ".1.weight", C, C, 7);
    raw(u + ".1.bias", C);
    raw(u + ".2.alpha", C);
    conv(u + ".3.weight", C, C, 1);
    raw(u + ".3.bias", C);
  };
  if (c->kind == SMI_VOC_BLOCK_RESUNIT) {
    unit("L.block", c->C);
  } else if (c->kind == SMI_VOC_BLOCK_DECBLOCK) {
    raw("L.block.0.alpha", c->C);
    const bool bf = use_bf(c->Cout, c->C);
    add("L.block.1.weight", bf ? PACK_CONVT_B : PACK_CONVT, c->Cout, c->C, c->K, c->S, (c->K - c->S) / 2,
        (size_t)conv_geom(c->Cout, c->C, c->K, 1, (c->K - c->S) / 2, c->S, bf).floats);
    raw("L.block.1.bias", c->Cout);
    for (int r = 0; r < 3; ++r) unit("L.block." + std::to_string(r + 2) + ".block", c->Cout);
  } else {
    const int D = c->C;
    raw("L.dwconv.weight", (size_t)D * 7);
    raw("L.dwconv.bias", D);
    if (c->cond_dim > 0) {
      add("cat:L.norm.scale.weight|L.norm.shift.weight", PACK_CONV, 2 * D, c->cond_dim, 1, 1, 0, (size_t)conv_geom(2 * D, c->cond_dim, 1, 1, 0, 1).floats);
      raw("cat:L.norm.scale.bias|L.norm.shift.bias", (size_t)2 * D);
    } else {
      raw("L.norm.weight", D);
      raw("L.norm.bias", D);
    }
    conv("L.pwconv1.weight", c->I, D, 1);
    raw("L.pwconv1.bias", c->I);
    conv("L.pwconv2.weight", D, c->I, 1);
    raw("L.pwconv2.bias", D);
    raw("L.gamma", D);
  }
  L.total = o;
  return L;
}

int layout_entry(const VocLayout& L, int index, char* name, int name_cap, size_t* offset, size_t* bytes, int32_t* info, const char* who) {
  SMI_REQUIRE(index >= 0 && index < (int)L.e.size(), "%s: index %d out of range", who, index);
  const Entry& e = L.e[index];
  SMI_REQUIRE(name && name_cap > (int)e.name.size(), "%s: name buffer too small (%zu needed)", who, e.name.size() + 1);
  strcpy(name, e.name.c_str());
  if (offset) *offset = e.offset;
  if (bytes) *bytes = e.bytes;
  if (info) { info[0] = e.kind; info[1] = e.Cout; info[2] = e.Cin; info[3] = e.K; info[4] = e.S; info[5] = e.pad; }
  return SMI_OK;
}

}  // namespace

extern "C" {

int smi_voc_arena_count(const smi_voc_cfg* cfg) {
  if (!voc_cfg_ok(cfg)) return 0;
  return (int)voc_layout(cfg).e.size();
}

size_t smi_voc_arena_bytes(const smi_voc_cfg* cfg) {
  if (!voc_cfg_ok(cfg)) return 0;
  return voc_layout(cfg).total;
}

int smi_voc_arena_entry(const smi_voc_cfg* cfg, int index, char* name, int name_cap, size_t* offset, size_t* bytes,
                        int32_t* info) {
  SMI_REQUIRE(voc_cfg_ok(cfg), "smi_voc_arena_entry: config outside the kernel contract");
  return layout_entry(voc_layout(cfg), index, name, name_cap, offset, bytes, info, "smi_voc_arena_entry");
}

int smi_voc_create(const smi_voc_cfg* cfg, const void* arena_dev, size_t arena_bytes, smi_voc** out) {
  SMI_REQUIRE(out, "smi_voc_create: out is null");
  *out = nullptr;
  SMI_REQUIRE(voc_cfg_ok(cfg), "smi_voc_create: config outside the kernel contract");
  VocLayout lay = voc_layout(cfg);
  SMI_REQUIRE(arena_dev && arena_bytes >= lay.total, "smi_voc_create: arena too small (%zu < %zu)", arena_bytes, lay.total);
  SMI_REQUIRE(((uintptr_t)arena_dev & 255) == 0, "smi_voc_create: arena must be 256-byte aligned");
  smi_voc* h = new smi_voc();
  h->cfg = *cfg; h->lay = lay; h->arena = (const unsigned char*)arena_dev;
  h->lastB = h->lastT = 0; h->small = nullptr; h->lens_dev = nullptr; h->ev0 = h->ev1 = nullptr;
  for (int i = 0; i < 4; ++i) h->buf[i] = nullptr;
  for (int i = 0; i < 8; ++i) { h->dbg[i] = nullptr; h->dbg_floats[i] = 0; }
  const char* dbg = smi_env("SPARKMI_VOC_DEBUG");
  h->debug = dbg && dbg[0] == '1';
  // largest activation: channels x length over all stages
  size_t mx = (size_t)cfg->vq_input_dim * cfg->max_frames;
  mx = std::max(mx, (size_t)cfg->pre_inter * cfg->max_frames);
  mx = std::max(mx, (size_t)cfg->dec_channels * cfg->max_frames);
  size_t len = cfg->max_frames;
  for (int i = 0; i < cfg->dec_nblocks; ++i) {
    len *= cfg->dec_rates[i];
    mx = std::max(mx, (size_t)(cfg->dec_channels >> (i + 1)) * len);
  }
  h->buf_floats = smi_align_up(mx, 64);
  const size_t small_floats = (size_t)cfg->max_batch *
      ((size_t)cfg->spk_latent_dim * cfg->spk_token_num + cfg->spk_out_dim + (size_t)(1 + cfg->pre_layers) * 2 * cfg->pre_dim + 64);
  bool ok = true;
  for (int i = 0; i < 4 && ok; ++i) ok = hipMalloc((void**)&h->buf[i], h->buf_floats * cfg->max_batch * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->small, small_floats * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->lens_dev, 8 * cfg->max_batch * 4 + 64) == hipSuccess;
  if (ok && h->debug) {
    for (int i = 0; i < 8 && ok; ++i) {
      h->dbg_floats[i] = h->buf_floats * cfg->max_batch;
      ok = hipMalloc((void**)&h->dbg[i], h->dbg_floats[i] * 4) == hipSuccess;
    }
  }
  ok = ok && hipEventCreate(&h->ev0) == hipSuccess && hipEventCreate(&h->ev1) == hipSuccess;
  if (!ok) {
    smi_set_error("smi_voc_create: device allocation failed (%zu floats per activation buffer x %d)", h->buf_floats, cfg->max_batch);
    smi_voc_destroy(h);
    return SMI_ENOMEM;
  }
  *out = h;
  return SMI_OK;
}

int smi_voc_destroy(smi_voc* h) {
  if (!h) return SMI_OK;
  for (int i = 0; i < 4; ++i) if (h->buf[i]) (void)hipFree(h->buf[i]);
  for (int i = 0; i < 8; ++i) if (h->dbg[i]) (void)hipFree(h->dbg[i]);
  if (h->small) (void)hipFree(h->small);
  if (h->lens_dev) (void)hipFree(h->lens_dev);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
  return SMI_OK;
}

int smi_voc_forward(smi_voc* h, const int64_t* sem_dev, const int32_t* lens_host, const int32_t* glob_dev, int B, int T,
                    float* wav_dev, void* stream) {
  SMI_REQUIRE(h && sem_dev && lens_host && glob_dev && wav_dev, "smi_voc_forward: null argument");
  const smi_voc_cfg& c = h->cfg;
  SMI_REQUIRE(B >= 1 && B <= c.max_batch, "smi_voc_forward: B=%d outside 1..%d", B, c.max_batch);
  SMI_REQUIRE(T >= 1 && T <= c.max_frames, "smi_voc_forward: T_max=%d outside 1..%d", T, c.max_frames);
  hipStream_t st = (hipStream_t)stream;
  // valid lengths at every resolution: lens_dev[s][b] = lens[b] * prod(rates[0..s))
  std::vector<int32_t>& hl = h->host_lens;
  hl.assign((size_t)8 * c.max_batch, 0);
  for (int b = 0; b < B; ++b) {
    SMI_REQUIRE(lens_host[b] >= 1 && lens_host[b] <= T, "smi_voc_forward: lens[%d]=%d outside 1..%d", b, lens_host[b], T);
    int up = 1;
    for (int s = 0; s <= c.dec_nblocks; ++s) {
      hl[(size_t)s * c.max_batch + b] = lens_host[b] * up;
      if (s < c.dec_nblocks) up *= c.dec_rates[s];
    }
    hl[(size_t)7 * c.max_batch + b] = 1;   // length-1 "sequences" for the d-vector GEMVs
  }
  SMI_HIP(hipMemcpyAsync(h->lens_dev, hl.data(), hl.size() * 4, hipMemcpyHostToDevice, st));
  auto lens_at = [&](int s) { return (const int*)(h->lens_dev + (size_t)s * c.max_batch); };
  const int* len1 = lens_at(7);
  const int* len0 = lens_at(0);

  std::vector<Launch>& P = h->prog;
  P.clear();
  const long long bs = (long long)h->buf_floats;
  float* bufs[4] = {h->buf[0], h->buf[1], h->buf[2], h->buf[3]};
  // a buffer that none of the listed live activations occupies
  auto other = [&](std::initializer_list<const float*> used) -> float* {
    for (float* f : bufs) {
      bool u = false;
      for (const float* q : used) u |= (q == f);
      if (!u) return f;
    }
    return nullptr;
  };
  const int D = c.pre_dim, I = c.pre_inter;
  const int nada = 1 + c.pre_layers;
  // small scratch: [lat: B x (latent*Ntok)] [dvec: B x out] [ada: B x nada*2*D]
  const int latn = c.spk_latent_dim * c.spk_token_num;
  float* lat = h->small;
  float* dvec = lat + (size_t)c.max_batch * latn;
  float* ada = dvec + (size_t)c.max_batch * c.spk_out_dim;
  const int ada_stride = nada * 2 * D;

  // --- d-vector: FSQ codes -> latent -> project (speaker_encoder.py:107-112)
  {
    Launch L; L.kind = 3; L.name = "fsq_latent"; L.flops = 2.0 * B * latn * c.fsq_dims; L.B = B;
    L.fp.glob = glob_dev; L.fp.W1 = ent(h, "speaker_encoder.quantizer.project_out.weight");
    L.fp.b1 = ent(h, "speaker_encoder.quantizer.project_out.bias"); L.fp.out = lat;
    L.fp.Ntok = c.spk_token_num; L.fp.latent = c.spk_latent_dim; L.fp.nd = c.fsq_dims;
    for (int j = 0; j < 8; ++j) L.fp.levels[j] = j < c.fsq_dims ? c.fsq_levels[j] : 1;
    P.push_back(L);
  }
  P.push_back(make_conv(h, "spk_project", "speaker_encoder.project.weight", "speaker_encoder.project.bias",
                        c.spk_out_dim, latn, 1, 1, 1, 0, lat, 1, latn, dvec, nullptr, nullptr, nullptr, 1, c.spk_out_dim,
                        len1, B, 1, ACT_NONE));
  auto use_gemv = [&](Launch& L) { L.gemv = true; L.grid = dim3((L.cp.Cout + 31) / 32, B); };
  use_gemv(P.back());
  // all AdaLayerNorm scale/shift projections of the condition in one GEMM (vocos.py:105-108)
  {
    std::string wn, bn;
    for (const Entry& e : h->lay.e) {
      if (e.name.rfind("cat:", 0) == 0 && e.kind == PACK_CONV) wn = e.name;
      if (e.name.rfind("cat:", 0) == 0 && e.kind == PACK_RAW) bn = e.name;
    }
    P.push_back(make_conv(h, "adaln_params", wn, bn.c_str(), ada_stride, c.pre_cond_dim, 1, 1, 1, 0, dvec, 1, c.spk_out_dim,
                          ada, nullptr, nullptr, nullptr, 1, ada_stride, len1, B, 1, ACT_NONE));
    use_gemv(P.back());
  }
  // --- semantic tokens -> codebook rows -> out_project (factorized_vector_quantize.py:154-167)
  float* zc = bufs[0];
  {
    Launch L; L.kind = 2; L.name = "codebook"; L.flops = 0;
    L.sem = sem_dev; L.semstride = T; L.cb = ent(h, "quantizer.codebook.weight"); L.D = c.codebook_dim; L.cbsize = c.codebook_size;
    L.lens = len0; L.Z = zc; L.zstride = T; L.zb = bs; L.grid = dim3((T + 127) / 128, B);
    P.push_back(L);
  }
  float* zq = other({zc});
  P.push_back(make_conv(h, "vq_out_project", "quantizer.out_project.weight", "quantizer.out_project.bias", c.vq_input_dim,
                        c.codebook_dim, 1, 1, 1, 0, zc, T, bs, zq, nullptr, nullptr, nullptr, T, bs, len0, B, T, ACT_NONE));
  if (h->debug) { Launch cp = P.back(); cp.cp.Y = h->dbg[0]; cp.name = "vq_out_project(dbg)"; P.push_back(cp); }
  // --- prenet (feat_decoder.py:78-94)
  float* cur = other({zq});
  P.push_back(make_conv(h, "prenet.linear_pre", "prenet.linear_pre.weight", "prenet.linear_pre.bias", D, c.pre_input_channels, 1,
                        1, 1, 0, zq, T, bs, cur, nullptr, nullptr, nullptr, T, bs, len0, B, T, ACT_NONE));
  P.back().cp.out_scale = c.pre_num_down > 0 ? 3.0f : 1.0f;   // first SamplingBlock(ratio 1): 3x
  int ada_idx = 0;
  const WSrc ws = wsrc(h);
  auto lnorm = [&](const std::string& name, const std::string& pfx, bool ada_norm, const float* X, float* Y, int triple) {
    const float* ap = ada_norm ? ada + (size_t)(ada_idx++) * 2 * D : nullptr;
    add_lnorm(P, ws, name, pfx, ap, ada_stride, nullptr, nullptr, X, Y, triple, D, T, bs, len0, B);
  };
  // embed conv7 -> norm -> nl x ConvNeXt -> final LN (vocos.py:324-335); consumes cur, leaves result in cur
  auto vocos = [&](const std::string& p, int nl, bool ada_norm, int triple_out) {
    float* n = other({cur});          // conv / norm output
    P.push_back(make_conv(h, p + ".embed", p + ".embed.weight", (p + ".embed.bias").c_str(), D, D, 7, 1, 1, 3, cur, T, bs, n,
                          nullptr, nullptr, nullptr, T, bs, len0, B, T, ACT_NONE));
    float* x = other({n});            // residual stream (cur is dead once embed has run)
    lnorm(p + ".norm", p + ".norm", ada_norm, n, x, 0);
    float* m = other({x, n});         // MLP hidden
    for (int j = 0; j < nl; ++j) {
      const float* ap = ada_norm ? ada + (size_t)(ada_idx++) * 2 * D : nullptr;
      add_convnext(P, ws, p + ".convnext." + std::to_string(j), D, I, x, n, m, ap, ada_stride, T, bs, len0, B);
    }
    lnorm(p + ".final_layer_norm", p + ".final_layer_norm", false, x, n, triple_out);
    cur = n;
  };
  for (int i = 0; i < c.pre_num_down; ++i)
    vocos("prenet.downsample." + std::to_string(i) + ".1", 2, false, (i + 1 < c.pre_num_down) ? 1 : 0);
  vocos("prenet.vocos_backbone", c.pre_layers, c.pre_cond_dim > 0, 0);
  SMI_REQUIRE(!c.pre_tanh_final, "smi_voc_forward: use_tanh_at_final is not supported");
  // prenet.linear, then + d_vector per utterance (bicodec.py:185-186)
  float* x0 = other({cur});
  P.push_back(make_conv(h, "prenet.linear+d", "prenet.linear.weight", "prenet.linear.bias", c.pre_out_channels, D, 1, 1, 1, 0, cur, T,
                        bs, x0, nullptr, nullptr, nullptr, T, bs, len0, B, T, ACT_NONE));
  P.back().cp.bbias = dvec;
  if (h->debug) { Launch cp = P.back(); cp.cp.Y = h->dbg[1]; cp.name = "prenet.linear+d(dbg)"; P.push_back(cp); }
  // --- WaveGenerator (wave_generator.py:56-88)
  int ch = c.dec_channels;
  int Lcur = T;
  float* s_in = other({x0});   // snake'd input of the next ConvTranspose
  {
    const std::string a0 = "decoder.model.1.block.0.alpha";
    P.push_back(make_conv(h, "decoder.conv_in", "decoder.model.0.weight", "decoder.model.0.bias", ch, c.dec_in, 7, 1, 1, 3, x0, T, bs,
                          h->debug ? h->dbg[2] : nullptr, s_in, ent(h, a0), nullptr, T, bs, len0, B, T, ACT_NONE));
  }
  for (int i = 0; i < c.dec_nblocks; ++i) {
    const int cin = ch >> i, cout = ch >> (i + 1), k = c.dec_ksizes[i], s = c.dec_rates[i];
    const std::string b = "decoder.model." + std::to_string(i + 1) + ".block";
    const int Lout = Lcur * s;
    float* U = other({s_in});
    float* US = other({s_in, U});
    float* A = other({s_in, U, US});
    // Snake (already applied by the producer) -> ConvTranspose1d -> three ResidualUnits; the last unit's second output is the
    // NEXT consumer's Snake of the block's result (the next block's, or the output conv's)
    const std::string next_alpha = i + 1 < c.dec_nblocks ? "decoder.model." + std::to_string(i + 2) + ".block.0.alpha"
                                                         : "decoder.model." + std::to_string(c.dec_nblocks + 1) + ".alpha";
    s_in = add_dec_block(P, ws, b, cin, cout, k, s, s_in, Lcur, bs, U, US, A, h->debug ? h->dbg[3 + i] : nullptr, ent(h, next_alpha), bs,
                         lens_at(i), lens_at(i + 1), B);
    Lcur = Lout;
  }
  const int clast = ch >> c.dec_nblocks;
  const int hop = Lcur / T;
  {
    const std::string n = "decoder.model." + std::to_string(c.dec_nblocks + 2);
    // final conv7 -> tanh, written straight into the caller's waveform buffer [B][hop*T]
    P.push_back(make_conv(h, "decoder.conv_out+tanh", n + ".weight", (n + ".bias").c_str(), 1, clast, 7, 1, 1, 3, s_in, Lcur, bs,
                          wav_dev, nullptr, nullptr, nullptr, Lcur, Lcur, lens_at(c.dec_nblocks), B, Lcur, ACT_TANH));
    {   // C -> 1: a thread per output sample instead of a 32-row MFMA tile with one live row (SPARKMI_VOC_C1=0: the MFMA kernel)
      Launch& L = P.back();
      const char* e = smi_env("SPARKMI_VOC_C1");
      const ConvP& q = L.cp;
      if (!(e && e[0] == '0') && !L.bf && q.Cout == 1 && q.S == 1 && q.istr == 1 && q.ntaps[0] == 7 && !q.X2 && !q.bbias && !q.gamma &&
          !q.beta && !q.R && !q.Ys && q.Y && q.out_scale == 1.0f && q.Cin * 7 * 4 <= 48 * 1024) {
        L.c1 = true; L.c1_len = Lcur;
      }
    }
    Launch Z; Z.kind = 4; Z.name = "zero_tail"; Z.flops = 0; Z.wav = wav_dev; Z.wstride = Lcur; Z.lens = len0; Z.hop = hop;
    Z.grid = dim3((Lcur + 255) / 256, B);
    P.push_back(Z);
  }
  int rc = check_launches(P, "smi_voc_forward");
  if (rc) return rc;
  for (const Launch& L : P)
    if ((rc = run_launch(L, st))) return rc;
  h->lastB = B; h->lastT = T;
  return SMI_OK;
}

int smi_voc_debug_stage(smi_voc* h, int stage, float* out_dev, size_t max_floats, size_t* n_floats, void* stream) {
  SMI_REQUIRE(h && out_dev && n_floats, "smi_voc_debug_stage: null argument");
  SMI_REQUIRE(h->lastB > 0, "smi_voc_debug_stage: no forward has run");
  const smi_voc_cfg& c = h->cfg;
  const float* src = nullptr;
  size_t n = 0;
  if (stage == -1) {   // d-vector [B][out_dim]
    src = h->small + (size_t)c.max_batch * c.spk_latent_dim * c.spk_token_num;
    n = (size_t)h->lastB * c.spk_out_dim;
  } else {
    SMI_REQUIRE(h->debug, "smi_voc_debug_stage: create the handle with SPARKMI_VOC_DEBUG=1");
    SMI_REQUIRE(stage >= 0 && stage < 3 + c.dec_nblocks, "smi_voc_debug_stage: stage %d out of range", stage);
    src = h->dbg[stage];
    n = h->buf_floats * h->lastB;   // rows are [C][stride] inside each batch slot
  }
  SMI_REQUIRE(n <= max_floats, "smi_voc_debug_stage: output buffer too small (%zu > %zu)", n, max_floats);
  SMI_HIP(hipMemcpyAsync(out_dev, src, n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  *n_floats = n;
  return SMI_OK;
}

int smi_voc_num_launches(smi_voc* h) { return h ? (int)h->prog.size() : 0; }

int smi_voc_time_launch(smi_voc* h, int index, int iters, float* ms_avg, double* flops, char* name, int name_cap, void* stream) {
  SMI_REQUIRE(h && ms_avg && iters > 0, "smi_voc_time_launch: bad argument");
  SMI_REQUIRE(index >= 0 && index < (int)h->prog.size(), "smi_voc_time_launch: index %d out of range", index);
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

// ---- one block of the vocoder on caller tensors (op-level tests; include/sparkmi.h)
int smi_voc_block_arena_count(const smi_voc_block_cfg* cfg) { return block_cfg_ok(cfg) ? (int)block_layout(cfg).e.size() : 0; }
size_t smi_voc_block_arena_bytes(const smi_voc_block_cfg* cfg) { return block_cfg_ok(cfg) ? block_layout(cfg).total : 0; }
int smi_voc_block_arena_entry(const smi_voc_block_cfg* cfg, int index, char* name, int name_cap, size_t* offset, size_t* bytes, int32_t* info) {
  SMI_REQUIRE(block_cfg_ok(cfg), "smi_voc_block_arena_entry: config outside the kernel contract");
  return layout_entry(block_layout(cfg), index, name, name_cap, offset, bytes, info, "smi_voc_block_arena_entry");
}

int smi_voc_block_run(const smi_voc_block_cfg* cfg, const void* arena_dev, size_t arena_bytes, const float* x_dev, const float* xs_dev,
                      const float* cond_dev, const int32_t* lens_host, int B, int L, float* y_dev, void* stream) {
  SMI_REQUIRE(block_cfg_ok(cfg), "smi_voc_block_run: config outside the kernel contract");
  const smi_voc_block_cfg& c = *cfg;
  VocLayout lay = block_layout(cfg);
  SMI_REQUIRE(arena_dev && arena_bytes >= lay.total, "smi_voc_block_run: arena too small (%zu < %zu)", arena_bytes, lay.total);
  SMI_REQUIRE(((uintptr_t)arena_dev & 255) == 0, "smi_voc_block_run: arena must be 256-byte aligned");
  SMI_REQUIRE(y_dev && B >= 1 && B <= 64 && L >= 1 && L <= (1 << 20), "smi_voc_block_run: bad B / L / output");
  SMI_REQUIRE(c.kind == SMI_VOC_BLOCK_DECBLOCK ? xs_dev != nullptr : x_dev != nullptr, "smi_voc_block_run: input tensor is null");
  SMI_REQUIRE(c.kind != SMI_VOC_BLOCK_RESUNIT || xs_dev, "smi_voc_block_run: a ResidualUnit needs x and snake(x)");
  SMI_REQUIRE(c.kind != SMI_VOC_BLOCK_CONVNEXT || c.cond_dim == 0 || cond_dev, "smi_voc_block_run: AdaLayerNorm needs the condition vector");
  hipStream_t st = (hipStream_t)stream;
  const WSrc w{&lay.e, (const unsigned char*)arena_dev};
  const int Lout = c.kind == SMI_VOC_BLOCK_DECBLOCK ? L * c.S : L;
  const int Cin = c.C, Cres = c.kind == SMI_VOC_BLOCK_DECBLOCK ? c.Cout : c.C;
  const int Cmax = std::max(std::max(Cin, Cres), c.kind == SMI_VOC_BLOCK_CONVNEXT ? c.I : 1);
  const long long bs = (long long)smi_align_up((size_t)Cmax * Lout, 64);   // one batch stride for every working buffer, as in the vocoder
  // working buffers: [0] input / x, [1] snake(input) / n, [2] U or m, [3] US, [4] A ; ints: lens_in, lens_out, ones ; ada
  float* buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float* ada = nullptr;
  int* lens = nullptr;
  auto cleanup = [&]() {
    for (float* f : buf) if (f) (void)hipFree(f);
    if (ada) (void)hipFree(ada);
    if (lens) (void)hipFree(lens);
  };
  bool ok = true;
  for (int i = 0; i < 5 && ok; ++i) ok = hipMalloc((void**)&buf[i], (size_t)bs * B * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&ada, (size_t)B * 2 * c.C * 4 + 256) == hipSuccess;
  ok = ok && hipMalloc((void**)&lens, (size_t)3 * 64 * 4) == hipSuccess;
  if (!ok) { cleanup(); smi_set_error("smi_voc_block_run: device allocation failed"); return SMI_ENOMEM; }
  std::vector<int32_t> hl((size_t)3 * 64, 0);
  for (int b = 0; b < B; ++b) {
    const int n = lens_host ? lens_host[b] : L;
    if (n < 1 || n > L) { cleanup(); smi_set_error("smi_voc_block_run: lens[%d]=%d outside 1..%d", b, n, L); return SMI_EINVAL; }
    hl[b] = n; hl[64 + b] = n * (Lout / L); hl[128 + b] = 1;
  }
  int rc = SMI_OK;
  auto fail = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == SMI_OK) { smi_set_error("smi_voc_block_run: %s: %s", what, hipGetErrorString(e)); rc = SMI_EHIP; } };
  fail(hipMemcpyAsync(lens, hl.data(), hl.size() * 4, hipMemcpyHostToDevice, st), "lens upload");
  for (int i = 0; i < 5; ++i) fail(hipMemsetAsync(buf[i], 0, (size_t)bs * B * 4, st), "memset");
  auto copy_in = [&](float* dst, const float* src, int C, int len) {   // contiguous [B][C][len] -> working layout
    fail(hipMemcpy2DAsync(dst, (size_t)bs * 4, src, (size_t)C * len * 4, (size_t)C * len * 4, (size_t)B, hipMemcpyDeviceToDevice, st), "copy in");
  };
  std::vector<Launch> P;
  const int* lens_in = lens; const int* lens_out = lens + 64; const int* len1 = lens + 128;
  float* result = nullptr;
  if (c.kind == SMI_VOC_BLOCK_RESUNIT) {
    copy_in(buf[0], x_dev, c.C, L);
    copy_in(buf[1], xs_dev, c.C, L);
    add_res_unit(P, w, "L.block", c.C, c.dil, buf[0], buf[1], buf[4], buf[2], false, nullptr, L, bs, lens_in, B);
    result = buf[2];
  } else if (c.kind == SMI_VOC_BLOCK_DECBLOCK) {
    copy_in(buf[1], xs_dev, c.C, L);
    add_dec_block(P, w, "L.block", c.C, c.Cout, c.K, c.S, buf[1], L, bs, buf[2], buf[3], buf[4], buf[0], nullptr, bs, lens_in, lens_out, B);
    result = buf[0];
  } else {
    copy_in(buf[0], x_dev, c.C, L);
    if (c.cond_dim > 0) {   // all scale / shift projections of the condition in one GEMV (vocos.py:105-108), as smi_voc_forward does
      P.push_back(make_conv(w, "adaln_params", "cat:L.norm.scale.weight|L.norm.shift.weight", "cat:L.norm.scale.bias|L.norm.shift.bias",
                            2 * c.C, c.cond_dim, 1, 1, 1, 0, cond_dev, 1, c.cond_dim, ada, nullptr, nullptr, nullptr, 1, 2 * c.C, len1, B, 1, ACT_NONE));
      P.back().gemv = true; P.back().grid = dim3((2 * c.C + 31) / 32, B);
    }
    add_convnext(P, w, "L", c.C, c.I, buf[0], buf[1], buf[2], c.cond_dim > 0 ? ada : nullptr, 2 * c.C, L, bs, lens_in, B);
    result = buf[0];
  }
  if (rc == SMI_OK) rc = check_launches(P, "smi_voc_block_run");
  for (size_t i = 0; i < P.size() && rc == SMI_OK; ++i) rc = run_launch(P[i], st);
  if (rc == SMI_OK)
    fail(hipMemcpy2DAsync(y_dev, (size_t)Cres * Lout * 4, result, (size_t)bs * 4, (size_t)Cres * Lout * 4, (size_t)B, hipMemcpyDeviceToDevice, st), "copy out");
  fail(hipStreamSynchronize(st), "synchronize");
  cleanup();
  return rc;
}

}  // extern "C"
