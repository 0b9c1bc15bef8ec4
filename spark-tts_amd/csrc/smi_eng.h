// smi_eng.h -- the one-row decode step's 24 layers as ONE persistent launch (included by smi_llm.hip).
//
// Stands where launch_layers() stands for a single live sequence (BASELINE configs[1]: batch-1 greedy decode, the
// reference's `model.generate` loop at cli/SparkTTS.py:197-204; arithmetic of transformers' modeling_qwen2.py:150-252):
// instead of four dependent kernels per layer (QKV, attention + o_proj, gate_up, down: 96 launches whose fixed costs --
// boundary, cold first byte, reduction tail -- are 80 % of the step), one workgroup per CU stays resident for the whole
// stack and the five all-to-all edges of a layer are hand-offs INSIDE the launch.
//
// Weights.  A ninth wave per workgroup -- the loader -- streams a re-packed copy of the layer matrices (built once at
// create from the standard arena) into the eight consumer waves' LDS rings by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// instruction).  It never stops at a phase edge: while the consumers wait for a hand-off it keeps a thin trickle of fills
// going (a few KiB in flight, so their granule sweeps are not queued behind a burst), a ring's worth ahead, so the weights
// of the NEXT phases are in LDS before their operand exists and HBM latency is off the critical path.  (Issued by the
// consuming waves themselves, the fills sit in front of the sweeps' loads in each wave's in-order memory queue and the
// layer takes edges + stream instead of max(edges, stream): tools/edge_probe.py, 9.4 vs 14.1 us per layer.)
//
// Arithmetic = the launch path's, bit for bit.  k_gemm sums a row's dot product as NW chains (k tile kt -> chain kt mod NW,
// each chain three MFMA accumulators for the exact bf16 split hi / mid / lo of the operand, (lo + mid) + hi, chains added in
// order).  Here ONE v_mfma_f32_16x16x32_bf16 advances FOUR chains of a 4-row part by one k tile each: the A operand's 16
// rows are 4 k-tile segments x 4 weight rows, the B operand's columns 3s + c hold split term c of segment s's k tile, and
// the diagonal blocks D[4s + r][3s + c] are exactly chain (4 set + s)'s three accumulators.  Every product, every chain
// and every addition order is the one k_gemm / k_attn use, so tokens, logits and KV rows equal the launch path's bits
// (tests/test_engine_gpu.py), which keeps "a sequence's tokens do not depend on what else is live" true across paths.
//
// Hand-offs: 8-byte {tag, value} granules, smi_eng_comm.h.  Per layer: h -> QKV (all CUs), q|k|v -> the 14 head CUs,
// attention -> o_proj CUs, h_mid -> gate_up (all CUs), act -> down CUs.  Buffers are double-buffered by layer parity.
#pragma once
#include "smi_eng_comm.h"

namespace {

constexpr int kEngWaves = 8, kEngThreads = kEngWaves * 64;   // waves / threads of a workgroup (one per CU)
constexpr int kEngGather = kEngThreads - 64;                  // threads that sweep hand-offs: the last wave moves weights meanwhile
constexpr int kEngBlock = kEngThreads;
constexpr int kEngRingTotal = 112;   // 1-KiB slots of a CU's weight ring
constexpr int kEngMaxSlots = 20;    // 4-row parts per CU and phase
constexpr int kEngMaxJobs = 16;     // (part, chain set) jobs per wave and layer
constexpr int kEngMaxParts = 52;    // parts per CU and layer, all phases
enum { EPH_QKV = 0, EPH_O = 1, EPH_GU = 2, EPH_DOWN = 3 };

struct EngJob { uint16_t part; uint8_t set, nimg; uint16_t slot, goff; };                     // 8 bytes; goff: first image, index in the CU's layer stream
struct EngWavePlan { uint16_t jstart[5]; uint16_t pad; uint16_t pimg[4]; EngJob jobs[kEngMaxJobs]; };
struct EngCuPlan { uint16_t pstart[5]; int16_t head; uint16_t parts[kEngMaxParts]; uint16_t len_cu, pad; uint16_t pcount[4]; EngWavePlan w[kEngWaves]; };
static_assert(sizeof(EngJob) == 8 && sizeof(EngCuPlan) % 4 == 0, "plan layout");

struct EngP {
  int H, Q, KV, I, n_heads, n_kv, layers, max_pos;
  float eps;
  int KT[4], NW[4];
  const EngCuPlan* plan;
  const unsigned char* stream;   // [layer][cu][maxlen] 1-KiB images, a CU's layer share in issue order
  int maxlen, ncu;
  const unsigned char* arena; size_t layers_base, layer_stride, off_ln1, off_bqkv, off_ln2;
  const float* final_norm; const float2* rope;
  const RowDesc* rows;
  float* h;                 // [H] in: the step's input row (k_finalize / k_embed); out: the last layer's output
  const float* ss_in;       // [H / 4] partial sums of squares of the input row (embed_row's layout)
  unsigned char* xs_out;    // [H / 32][3][4][16 B] triples of final_norm * h for lm_head
  float* ss_out;            // [H / 4]
  uint16_t* kcache; uint16_t* vcache; size_t kv_layer_elems;
  smi_u64* gran;            // [2][gran_per_buf]
  unsigned* serial; unsigned* err; unsigned* arrive;
  unsigned timeout_ticks;
  unsigned long long* stamps;   // diagnostics: [2][layers][8] or null
  int ld_burst, ld_sleep;       // loader pacing: fills per look at the arrival counter, pause (x 64 cycles) between bursts
};

// ---- LDS carve (bytes), all multiples of 16
struct EngLds { int ring, xs_act, xs_x, hbuf, ssp, red, plan, total; };
__host__ __device__ inline EngLds eng_lds(int H, int KTact, int KTx) {
  EngLds l;
  int o = 0;
  l.ring = o; o += kEngRingTotal * 1024;
  int act = KTact * 192;
  const int attn = 16384 + 256 + 2048 + 64 + kEngWaves * 512;   // attention scratch aliases the act image (head CUs have no down parts)
  if (act < attn) act = attn;
  l.xs_act = o; o += (act + 15) / 16 * 16;
  l.xs_x = o; o += KTx * 192;
  l.hbuf = o; o += H * 4;
  l.ssp = o; o += (H / 4 * 4 + 15) / 16 * 16;
  l.red = o; o += kEngMaxSlots * 16 * 16;
  l.plan = o; o += ((int)sizeof(EngCuPlan) + 15) / 16 * 16;
  l.total = o + 64;
  return l;
}

// ---- repack: image (cu, wave, i) of layer l = A operand of one MFMA: lane (k8, 4 s + r) holds W[4 part + r][tile(s)][8 k8 .. +8]
struct EngPackP {
  const uint32_t* desc;     // [ncu][maxlen]  phase:2 | set:4 | round:10 | part:16
  const uint16_t* lens;     // [ncu]
  int maxlen, ncw;
  const unsigned char* arena; size_t layers_base, layer_stride;
  size_t woff[4]; int KT[4], NW[4], wperm[4];
  uint4* out;
};
__global__ __launch_bounds__(256) void k_eng_pack(EngPackP p) {
  const int lane = threadIdx.x & 63;
  const size_t img = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int layer = blockIdx.y;
  if (img >= (size_t)p.ncw * p.maxlen) return;
  const int cw = (int)(img / p.maxlen), i = (int)(img % p.maxlen);   // cw: the CU
  if (i >= p.lens[cw]) return;
  const uint32_t d = p.desc[img];
  const int ph = d >> 30, set = (d >> 26) & 15, round = (d >> 16) & 1023, part = d & 0xffff;
  const int k8 = lane >> 4, s = (lane >> 2) & 3, r = lane & 3;
  const int KT = p.KT[ph], NW = p.NW[ph];
  const int T = round * NW + 4 * set + s;
  const bool valid = 4 * set + s < NW && T < KT;
  const int n = 4 * part + r, nt = n >> 4;
  const int srcl = k8 * 16 + (n & 15);
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (valid) {
    const uint4* W = (const uint4*)(p.arena + p.layers_base + (size_t)layer * p.layer_stride + p.woff[ph]);
    v = W[((size_t)nt * KT + T) * 64 + smi_wlane(srcl, p.wperm[ph])];
  }
  p.out[((size_t)layer * p.ncw * p.maxlen + img) * 64 + lane] = v;
}

// ---- the engine
__device__ __forceinline__ float eng_rms_from_ssp(const float* ssp, int npart, int lane, int K, float eps) {
  // smi_ss_lane_sum's order over an LDS array, then smi_wave_sum
  float v = 0.f;
  for (int i0 = lane; i0 < npart; i0 += 256) {
    const int i1 = i0 + 64, i2 = i0 + 128, i3 = i0 + 192;
    v += ssp[i0];
    v += i1 < npart ? ssp[i1] : 0.f;
    v += i2 < npart ? ssp[i2] : 0.f;
    v += i3 < npart ? ssp[i3] : 0.f;
  }
  v = smi_wave_sum(v);
  return 1.0f / sqrtf(v / (float)K + eps);
}

// element k of a [tile][3][4][16 B] operand image: byte offset of its hi term (mid: +64, lo: +128)
__device__ __forceinline__ int eng_xs_elem(int tile, int kin) { return tile * 192 + ((kin >> 3) & 3) * 16 + (kin & 7) * 2; }
__device__ __forceinline__ void eng_put3(unsigned char* xs, int off, float x) {
  uint32_t hi, mi, lo;
  split3(x, hi, mi, lo);
  *(uint16_t*)(xs + off) = (uint16_t)hi;
  *(uint16_t*)(xs + off + 64) = (uint16_t)mi;
  *(uint16_t*)(xs + off + 128) = (uint16_t)lo;
}

__global__ __launch_bounds__(kEngBlock, 1) void k_engine(EngP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cu = blockIdx.x;
  const bool loader = wave == kEngWaves - 1;   // wave 7 streams weights while waves 0..6 wait for hand-offs; all eight compute
  const EngLds L = eng_lds(p.H, p.KT[EPH_DOWN], p.KT[EPH_QKV] > p.KT[EPH_O] ? p.KT[EPH_QKV] : p.KT[EPH_O]);
  unsigned char* xs_act = smem + L.xs_act;
  unsigned char* xs_x = smem + L.xs_x;
  float* hbuf = (float*)(smem + L.hbuf);
  float* ssp = (float*)(smem + L.ssp);
  float4* red = (float4*)(smem + L.red);          // [slot][chain]
  EngCuPlan* pl = (EngCuPlan*)(smem + L.plan);
  unsigned* arrive = (unsigned*)(smem + L.plan + ((int)sizeof(EngCuPlan) + 15) / 16 * 16);   // consumer waves arrived at barriers so far
  // LDS-DMA takes the ABSOLUTE LDS address in M0: the dynamic segment starts behind the kernel's static LDS (none today; kept exact)
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint32_t lbase = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lptr_t)smem);

  EngSync sy;
  sy.err = p.err;
  sy.t_end = __builtin_amdgcn_s_memrealtime() + p.timeout_ticks;
  {   // this CU's plan -> LDS
    const uint32_t* src = (const uint32_t*)(p.plan + cu);
    for (int i = tid; i < (int)(sizeof(EngCuPlan) / 4); i += kEngBlock) ((uint32_t*)pl)[i] = src[i];
    if (tid == 0) { arrive[0] = 0u; arrive[1] = 0u; }
  }
  const unsigned serial = *(volatile const unsigned*)p.serial;
  const unsigned tbase = serial * (unsigned)(p.layers * 8 + 16) + 1u;
  const RowDesc rd = p.rows[0];
  const int pos = rd.pos;
  __syncthreads();
  const EngWavePlan& wp = pl->w[wave];
  const int head = __builtin_amdgcn_readfirstlane((int)pl->head);   // >= 0: this CU runs that head's attention
  int np[4], p0[4];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    p0[ph] = __builtin_amdgcn_readfirstlane((int)pl->pstart[ph]);
    np[ph] = __builtin_amdgcn_readfirstlane((int)pl->pstart[ph + 1]) - p0[ph];
  }
  const int H = p.H, Q = p.Q, KVd = p.KV, I = p.I;
  const int gran_per_buf = 2 * H + (Q + 2 * KVd) + Q + I;
  const int gA = 0, gB = H, gC = H + Q + 2 * KVd, gD = gC + Q, gE = gD + H;

  // ---- every barrier of the workgroup goes through bar(): the consumers count their arrivals in LDS first, which is what
  //      the loader watches while it trickles (it joins when the last consumer wave has arrived)
  unsigned bars = 0;   // barriers passed so far (same sequence in all nine waves)
  // bar_chk (behind a hand-off wait, the only thing that can give up): one barrier, then everybody reads the workgroup's
  // fail word; bar (behind compute): one barrier.
  volatile unsigned* failw = arrive + 1;
  auto bar_chk = [&](bool okv) -> bool {
    if (!loader && lane == 0) atomicAdd(arrive, 1u);
    if (!okv) *failw = 1u;
    ++bars;
    __syncthreads();
    return *failw == 0u;
  };
  auto bar = [&]() {
    if (!loader && lane == 0) atomicAdd(arrive, 1u);
    ++bars;
    __syncthreads();
  };

  // ---- the weight ring: ONE ring of kEngRingTotal 1-KiB slots per CU, filled in the CU's stream order (layer, phase, wave,
  //      job, image); image number n (counted over the whole launch) lives in slot n mod kEngRingTotal.  The loader wave
  //      issues as far ahead as the ring allows; a phase's slots are free again behind the barrier that closes its compute.
  const int len_cu = __builtin_amdgcn_readfirstlane((int)pl->len_cu);
  int P_ph[4], pre_ph[4];          // images per phase of this CU, and the running total through each phase
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    P_ph[ph] = __builtin_amdgcn_readfirstlane((int)pl->pcount[ph]);
    pre_ph[ph] = P_ph[ph] + (ph ? pre_ph[ph - 1] : 0);
  }
  const unsigned char* cu_stream = p.stream + (size_t)cu * (size_t)p.maxlen * 1024 + lane * 16;
  const size_t layer_imgs_bytes = (size_t)p.ncu * (size_t)p.maxlen * 1024;
  int issued_total = 0, consumed_total = 0, is_layer = 0, is_idx = 0;   // loader state (wave-uniform)
  auto ld_can = [&]() -> bool { return is_layer < p.layers && issued_total - consumed_total < kEngRingTotal; };
  auto ld_issue = [&]() {       // precondition: ld_can()
    const unsigned char* src = cu_stream + (size_t)is_layer * layer_imgs_bytes + (size_t)is_idx * 1024;
    smi_glds16(src, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lbase + (uint32_t)L.ring + (uint32_t)(issued_total % kEngRingTotal) * 1024)));
    ++issued_total;
    if (++is_idx == len_cu) { is_idx = 0; ++is_layer; }
  };
  // Until the last sweeping wave has arrived at barrier number `bars + 1`: keep the ring full (p.ld_burst fills per look
  // at the arrival counter; a pause of p.ld_sleep x 64 cycles between bursts thins the stream beside the sweeps).
  auto ld_trickle = [&]() {
    const unsigned target = (bars + 1) * (kEngWaves - 1);
    for (unsigned it = 0;; ++it) {
      if (*(volatile unsigned*)arrive >= target) break;
      if (len_cu > 0) for (int b = 0; b < p.ld_burst && ld_can(); ++b) ld_issue();
      if (p.ld_sleep > 0 || !ld_can()) __builtin_amdgcn_s_sleep(1);
      for (int z = 1; z < p.ld_sleep; ++z) __builtin_amdgcn_s_sleep(1);
      if ((it & 4095u) == 4095u && __builtin_amdgcn_s_memrealtime() > sy.t_end + 100000000ull) break;   // (the consumers' spins are bounded; this bounds the loader's too)
    }
  };
  // Everything up to the end of (layer, phase) issued and landed (called before the barrier that opens that phase's compute).
  auto ld_ensure = [&](int layer, int ph) {
    const int target = layer * len_cu + pre_ph[ph];
    while (issued_total < target && ld_can()) ld_issue();   // fits: a CU's images of one phase never exceed the ring
    const int k = issued_total - target;   // fills issued behind the last one needed may stay in flight (they land in order)
    if (k >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (k >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (k >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (k >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (k >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (k >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (k >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto ld_consumed = [&](int ph) { consumed_total += P_ph[ph]; };   // behind the barrier that closes a phase's compute

  // ---- per-lane constants of the MFMA B operand: column col = 3 s + c of segment s, split term c
  const int col = lane & 15, k8 = lane >> 4;
  const int bs = col < 12 ? col / 3 : 3, bc = col < 12 ? col % 3 : 2;
  const int boff = (bc * 4 + k8) * 16;

  // one phase's jobs of this wave: operand image xs (tiles of 192 bytes), results to red[slot][chain]
  auto run_jobs = [&](int layer, int ph, const unsigned char* xs) {
    const int KT = p.KT[ph], NW = p.NW[ph];
    const int j0 = __builtin_amdgcn_readfirstlane((int)wp.jstart[ph]), j1 = __builtin_amdgcn_readfirstlane((int)wp.jstart[ph + 1]);
    for (int j = j0; j < j1; ++j) {
      const EngJob jb = wp.jobs[j];
      const int set = __builtin_amdgcn_readfirstlane((int)jb.set), nimg = __builtin_amdgcn_readfirstlane((int)jb.nimg);
      const int slot = __builtin_amdgcn_readfirstlane((int)jb.slot);
      const int img0 = layer * len_cu + __builtin_amdgcn_readfirstlane((int)jb.goff);   // launch-wide number of the job's first image
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      int T = 4 * set + bs;
      auto lda = [&](int i) -> bf16x8 { return *(const bf16x8*)(smem + (uint32_t)L.ring + (uint32_t)((img0 + i) % kEngRingTotal) * 1024 + lane * 16); };
      auto ldb = [&](int Tt) -> bf16x8 { return *(const bf16x8*)(xs + (Tt < KT ? Tt : KT - 1) * 192 + boff); };
      if (nimg > 0) {   // the next image's operands are requested before the current MFMA (one LDS round trip per job, not per image)
        bf16x8 a = lda(0), b = ldb(T);
        for (int i = 0; i < nimg; ++i) {
          T += NW;
          const int in = i + 1 < nimg ? i + 1 : i;
          const bf16x8 an = lda(in), bn = ldb(i + 1 < nimg ? T : T - NW);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
          a = an; b = bn;
        }
      }
      // chain 4 set + s: hi / mid / lo accumulators are columns 3 s, 3 s + 1, 3 s + 2 of rows 4 s .. 4 s + 3 (lanes 19 s + c)
      f32x4 t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float mid = __shfl_down(acc[r], 1, 64), lo = __shfl_down(acc[r], 2, 64);
        t[r] = (lo + mid) + acc[r];
      }
      const int s = lane / 19;
      if (lane == 19 * s && s < 4 && 4 * set + s < NW) red[slot * 16 + 4 * set + s] = make_float4(t[0], t[1], t[2], t[3]);
    }
  };
  auto finish = [&](int ph, int slot) -> float4 {   // chains summed in order (all 16 loads leave together: one LDS round trip)
    const int NW = p.NW[ph];
    float4 t[16];
#pragma unroll
    for (int w = 0; w < 16; ++w) t[w] = red[slot * 16 + (w < NW ? w : 0)];
    float4 s = t[0];
#pragma unroll
    for (int w = 1; w < 16; ++w)
      if (w < NW) { s.x += t[w].x; s.y += t[w].y; s.z += t[w].z; s.w += t[w].w; }
    return s;
  };
  // A hidden-size vector arrives (hand-off granules, or the step's input row at layer 0): thread j takes elements 4j .. 4j+3,
  // leaves them in hbuf, their exact triples (times the norm weight) in xs_x -- one 8-byte LDS store per split term -- and
  // their partial sum of squares in ssp[j] (the RESID epilogue's arithmetic).  `goff` < 0: read p.h / p.ss_in instead.
  auto stage_h = [&](const smi_u64* gbuf, int goff, unsigned tag, const float* gamma, unsigned where) -> bool {
    bool okv = true;
    for (int j0 = 0; j0 < H / 4; j0 += kEngGather) {
      const int j = j0 + tid;
      const bool act = j < H / 4;
      const int jc = act ? j : 0;
      const float4 gam = *(const float4*)(gamma + 4 * jc);
      float4 hv;
      float ssv = 0.f;
      if (goff < 0) {
        hv = *(const float4*)(p.h + 4 * jc);
        ssv = p.ss_in[jc];
      } else {
        unsigned v[4];
        const int idx[4] = {act ? goff + 4 * j : -1, act ? goff + 4 * j + 1 : -1, act ? goff + 4 * j + 2 : -1, act ? goff + 4 * j + 3 : -1};
        okv = eng_sweep_idx<4>(gbuf, idx, tag, v, sy, where) && okv;
        hv = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        ssv = (hv.x * hv.x + hv.y * hv.y) + (hv.z * hv.z + hv.w * hv.w);
      }
      if (act) {
        *(float4*)(hbuf + 4 * j) = hv;
        ssp[j] = ssv;
        const float t[4] = {gam.x * hv.x, gam.y * hv.y, gam.z * hv.z, gam.w * hv.w};
        uint32_t hi[4], mi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(t[e], hi[e], mi[e], lo[e]);
        unsigned char* o = xs_x + eng_xs_elem((4 * j) >> 5, (4 * j) & 31);
        *(uint2*)(o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
        *(uint2*)(o + 64) = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
        *(uint2*)(o + 128) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
      }
    }
    return okv;
  };
  float* rfac = (float*)(arrive + 2);   // the norm factor of the phase under way: computed beside the jobs, read by the epilogue
  // rope factors of this CU's QKV parts do not depend on the layer
  float2 rope0 = make_float2(1.f, 0.f), rope1 = rope0;
  int qn = 0;
  if (tid < np[EPH_QKV]) {
    qn = 4 * (int)pl->parts[p0[EPH_QKV] + tid];
    if (qn < Q + KVd) {
      const int i0 = (qn & 63) >> 1;
      rope0 = p.rope[(size_t)pos * 32 + i0];
      rope1 = p.rope[(size_t)pos * 32 + i0 + 1];
    }
  }
  const int kvh = head >= 0 ? head / (p.n_heads / p.n_kv) : 0;
  const bool need_h = np[EPH_QKV] > 0 || np[EPH_O] > 0;        // (the o_proj epilogue adds its rows of h)
  const bool need_hmid = np[EPH_GU] > 0 || np[EPH_DOWN] > 0;   // (the down_proj epilogue adds its rows of h_mid)
  bool ok = true;
  for (int layer = 0; layer < p.layers; ++layer) {
    const unsigned char* lay = p.arena + p.layers_base + (size_t)layer * p.layer_stride;
    smi_u64* gb = p.gran + (size_t)(layer & 1) * gran_per_buf;
    smi_u64* gnext = p.gran + (size_t)((layer + 1) & 1) * gran_per_buf;
    const unsigned tl = tbase + (unsigned)layer * 8u;
    const bool stamp = p.stamps && tid == 0 && (cu == 0 || head == 0);
    unsigned long long* sp = p.stamps ? p.stamps + ((size_t)(cu == 0 ? 0 : 1) * p.layers + layer) * 8 : nullptr;
    constexpr int UNR = 4, DPL = 8, NGRP = kEngWaves * 8;
    const int tl8 = lane >> 3, dl = lane & 7, grp = wave * 8 + tl8;
    uint4 kr[UNR], vr[UNR];
    const uint16_t* kc = p.kcache + (size_t)layer * p.kv_layer_elems;
    const uint16_t* vc = p.vcache + (size_t)layer * p.kv_layer_elems;
    const size_t rowbase = (size_t)kvh * p.max_pos;   // slot 0
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);

    // ================= edge A: h -> QKV =================
    if (!loader) {
      // (requests that do not depend on the hand-off leave first: norm weights, bias)
      if (tid < np[EPH_QKV]) bq = *(const float4*)((const float*)(lay + p.off_bqkv) + qn);
      if (need_h)
        ok = stage_h(gb, layer == 0 ? -1 : gA, tl + 0, (const float*)(lay + p.off_ln1), (unsigned)(layer * 8 + 1)) && ok;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      ld_trickle();
      ld_ensure(layer, EPH_QKV);
    }
    ok = bar_chk(ok);
    if (stamp) sp[0] = __builtin_amdgcn_s_memrealtime();
    if (!ok) break;
    // the head CUs' first K/V chunk: requested here, used after the q / k / v hand-off (the QKV phase covers the round trip)
    if (head >= 0) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int t = u * NGRP + grp;
        const int tc = t < p.max_pos ? t : p.max_pos - 1;
        const size_t off = (rowbase + tc) * kHeadDim + dl * DPL;
        kr[u] = *(const uint4*)(kc + off);
        vr[u] = *(const uint4*)(vc + off);
      }
    }
    if (np[EPH_QKV] > 0) {
      if (loader) {   // RMSNorm factor of h, beside the jobs (the wave that did not sweep)
        const float r = eng_rms_from_ssp(ssp, H / 4, lane, p.KT[EPH_QKV] * 32, p.eps);
        if (lane == 0) rfac[0] = r;
      }
      run_jobs(layer, EPH_QKV, xs_x);
    }
    bar();
    if (stamp) sp[5] = __builtin_amdgcn_s_memrealtime();
    if (loader) ld_consumed(EPH_QKV);
    if (wave == 0 && np[EPH_QKV] > 0) {
      const float r1 = rfac[0];
      if (tid < np[EPH_QKV]) {
        float4 s = finish(EPH_QKV, tid);
        s.x *= r1; s.y *= r1; s.z *= r1; s.w *= r1;
        s.x += bq.x; s.y += bq.y; s.z += bq.z; s.w += bq.w;
        if (qn < Q + KVd) {
          float4 r;
          r.x = __fadd_rn(__fmul_rn(s.x, rope0.x), __fmul_rn(-s.y, rope0.y));
          r.y = __fadd_rn(__fmul_rn(s.y, rope0.x), __fmul_rn(s.x, rope0.y));
          r.z = __fadd_rn(__fmul_rn(s.z, rope1.x), __fmul_rn(-s.w, rope1.y));
          r.w = __fadd_rn(__fmul_rn(s.w, rope1.x), __fmul_rn(s.z, rope1.y));
          s = r;
        }
        eng_gstore(gb + gB + qn + 0, tl + 1, __float_as_uint(s.x));
        eng_gstore(gb + gB + qn + 1, tl + 1, __float_as_uint(s.y));
        eng_gstore(gb + gB + qn + 2, tl + 1, __float_as_uint(s.z));
        eng_gstore(gb + gB + qn + 3, tl + 1, __float_as_uint(s.w));
        if (qn >= Q) {   // K / V row of this step, for the later steps (this step's attention takes it from the granules)
          const bool isk = qn < Q + KVd;
          const int c = qn - Q - (isk ? 0 : KVd);
          const size_t off = ((size_t)(c >> 6) * p.max_pos + pos) * kHeadDim + (c & 63);
          uint2 pk;
          pk.x = smi_f32_to_bf16(s.x) | (smi_f32_to_bf16(s.y) << 16);
          pk.y = smi_f32_to_bf16(s.z) | (smi_f32_to_bf16(s.w) << 16);
          uint16_t* base = (isk ? p.kcache : p.vcache) + (size_t)layer * p.kv_layer_elems;
          *(uint2*)(base + off) = pk;
        }
      }
    }

    // ================= edge B + attention (head CUs): k_attn<bf16 KV, one row>'s arithmetic =================
    if (head >= 0) {
      constexpr float NEG = -1e30f, LOG2E = 1.4426950408889634f;
      unsigned char* asc = xs_act;   // attention scratch over the act image
      float (*so)[8][kHeadDim] = (float (*)[8][kHeadDim])asc;                   // [8][8][64]
      float (*slm)[8] = (float (*)[8])(asc + 16384);                             // [8][8]
      float (*pw)[kHeadDim] = (float (*)[kHeadDim])(asc + 16384 + 256);          // [8][64]
      float* plw = (float*)(asc + 16384 + 256 + 2048);                           // [8]
      float* wmax = plw + 8;                                                     // [8]
      {
        unsigned char* xw = asc + 16384 + 256 + 2048 + 64 + wave * 512;          // this wave's q (256 B) | k (128 B) | v (128 B)
        {
          unsigned v[3];
          const int idx[3] = {gB + head * 64 + lane, gB + Q + kvh * 64 + lane, gB + Q + KVd + kvh * 64 + lane};
          ok = eng_sweep_idx<3>(gb, idx, tl + 1, v, sy, (unsigned)(layer * 8 + 2));
          ((float*)xw)[lane] = __uint_as_float(v[0]);
          ((uint16_t*)(xw + 256))[lane] = (uint16_t)smi_f32_to_bf16(__uint_as_float(v[1]));
          ((uint16_t*)(xw + 384))[lane] = (uint16_t)smi_f32_to_bf16(__uint_as_float(v[2]));
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (stamp) sp[1] = __builtin_amdgcn_s_memrealtime();
        float qv[DPL];
#pragma unroll
        for (int i = 0; i < DPL; ++i) qv[i] = ((const float*)xw)[dl * DPL + i] * 0.125f;
        const uint4 knew = *(const uint4*)(xw + 256 + dl * 16), vnew = *(const uint4*)(xw + 384 + dl * 16);
        const int ctx = pos + 1 < kAttnSeg ? pos + 1 : kAttnSeg;
        float m_run = NEG, lrun = 0.f, o[DPL];
#pragma unroll
        for (int i = 0; i < DPL; ++i) o[i] = 0.f;
        int c0 = 0;
        do {
          if (c0 > 0) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
              const int t = c0 + u * NGRP + grp;
              const int tc = t < p.max_pos ? t : p.max_pos - 1;
              const size_t off = (rowbase + tc) * kHeadDim + dl * DPL;
              kr[u] = *(const uint4*)(kc + off);
              vr[u] = *(const uint4*)(vc + off);
            }
          }
#pragma unroll
          for (int u = 0; u < UNR; ++u)
            if (c0 + u * NGRP + grp == pos) { kr[u] = knew; vr[u] = vnew; }   // this step's own key: not in the cache yet for this launch
          float sc[UNR];
          float lmax = NEG;
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const uint32_t ku[4] = {kr[u].x, kr[u].y, kr[u].z, kr[u].w};
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              d += qv[(2 * i) % DPL] * __uint_as_float(ku[i] << 16);
              d += qv[(2 * i + 1) % DPL] * __uint_as_float(ku[i] & 0xffff0000u);
            }
            d = smi_sum8(d);
            sc[u] = (c0 + u * NGRP + grp < ctx) ? d : NEG;
            lmax = fmaxf(lmax, sc[u]);
          }
          lmax = fmaxf(lmax, smi_dpp<0x128>(lmax));   // row_ror:8
          const float wm = fmaxf(fmaxf(smi_readlane(lmax, 0), smi_readlane(lmax, 16)), fmaxf(smi_readlane(lmax, 32), smi_readlane(lmax, 48)));
          const float mn = fmaxf(m_run, wm);
          const float a = exp2f((m_run - mn) * LOG2E);
          lrun *= a;
#pragma unroll
          for (int i = 0; i < DPL; ++i) o[i] *= a;
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const float e = sc[u] > 0.5f * NEG ? exp2f((sc[u] - mn) * LOG2E) : 0.f;
            const uint32_t vu[4] = {vr[u].x, vr[u].y, vr[u].z, vr[u].w};
            lrun += e;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              o[(2 * i) % DPL] += e * __uint_as_float(vu[i] << 16);
              o[(2 * i + 1) % DPL] += e * __uint_as_float(vu[i] & 0xffff0000u);
            }
          }
          m_run = mn;
          c0 += NGRP * UNR;
        } while (c0 < ctx);
#pragma unroll
        for (int i = 0; i < DPL; i += 4) *(float4*)&so[wave][tl8][dl * DPL + i] = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
        if (dl == 0) slm[wave][tl8] = lrun;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
          float O = 0.f, Ls = 0.f;
#pragma unroll
          for (int g = 0; g < 8; ++g) { O += so[wave][g][lane]; Ls += slm[wave][g]; }
          pw[wave][lane] = O;
          if (lane == 0) { plw[wave] = Ls; wmax[wave] = m_run; }
        }
      }
      ok = bar_chk(ok);
      if (!ok) break;
      if (tid < kHeadDim) {
        float bm = wmax[0];
#pragma unroll
        for (int w = 1; w < kEngWaves; ++w) bm = fmaxf(bm, wmax[w]);
        float O = 0.f, Ls = 0.f;
#pragma unroll
        for (int w = 0; w < kEngWaves; ++w) {
          const float scw = exp2f((wmax[w] - bm) * LOG2E);
          O += pw[w][tid] * scw; Ls += plw[w] * scw;
        }
        eng_gstore(gb + gC + head * 64 + tid, tl + 2, __float_as_uint(O / Ls));
        if (stamp) sp[2] = __builtin_amdgcn_s_memrealtime();   // (head CUs have no o_proj parts: slot 2 = attention published)
      }
      bar();   // the scratch is reused by the next layer's attention
    }

    // ================= edge C: attention output -> o_proj CUs =================
    if (np[EPH_O] > 0) {
      if (!loader) {
        for (int e0 = 0; e0 < Q; e0 += 2 * kEngGather) {
          const int e[2] = {e0 + tid, e0 + tid + kEngGather};
          unsigned v[2];
          const int idx[2] = {e[0] < Q ? gC + e[0] : -1, e[1] < Q ? gC + e[1] : -1};
          ok = eng_sweep_idx<2>(gb, idx, tl + 2, v, sy, (unsigned)(layer * 8 + 3)) && ok;
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if (e[k] < Q) {
              const int hd = e[k] >> 6, d = e[k] & 63;
              eng_put3(xs_x, eng_xs_elem(o_ktile(hd, d, p.n_heads), d & 31), __uint_as_float(v[k]));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        ld_trickle();
        ld_ensure(layer, EPH_O);
      }
      ok = bar_chk(ok);
      if (stamp) sp[2] = __builtin_amdgcn_s_memrealtime();
      if (!ok) break;
      run_jobs(layer, EPH_O, xs_x);
      bar();
      if (loader) ld_consumed(EPH_O);
      if (tid < np[EPH_O]) {
        const int n = 4 * (int)pl->parts[p0[EPH_O] + tid];
        const float4 s = finish(EPH_O, tid);
        float4 h4 = *(const float4*)(hbuf + n);
        h4.x += s.x; h4.y += s.y; h4.z += s.z; h4.w += s.w;
        eng_gstore(gb + gD + n + 0, tl + 3, __float_as_uint(h4.x));
        eng_gstore(gb + gD + n + 1, tl + 3, __float_as_uint(h4.y));
        eng_gstore(gb + gD + n + 2, tl + 3, __float_as_uint(h4.z));
        eng_gstore(gb + gD + n + 3, tl + 3, __float_as_uint(h4.w));
      }
    }

    // ================= edge D: h_mid -> gate_up =================
    if (!loader) {
      if (need_hmid)
        ok = stage_h(gb, gD, tl + 3, (const float*)(lay + p.off_ln2), (unsigned)(layer * 8 + 4)) && ok;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      ld_trickle();
      ld_ensure(layer, EPH_GU);
    }
    ok = bar_chk(ok);
    if (stamp) sp[3] = __builtin_amdgcn_s_memrealtime();
    if (!ok) break;
    if (np[EPH_GU] > 0) {
      if (loader) {   // RMSNorm factor of h_mid, beside the jobs
        const float r = eng_rms_from_ssp(ssp, H / 4, lane, p.KT[EPH_GU] * 32, p.eps);
        if (lane == 0) rfac[0] = r;
      }
      run_jobs(layer, EPH_GU, xs_x);
    }
    bar();
    if (stamp) sp[6] = __builtin_amdgcn_s_memrealtime();
    if (loader) ld_consumed(EPH_GU);
    if (wave == 0 && np[EPH_GU] > 0) {
      const float r2 = rfac[0];
      if (tid < np[EPH_GU]) {
        const int n = 4 * (int)pl->parts[p0[EPH_GU] + tid];
        float4 s = finish(EPH_GU, tid);
        s.x *= r2; s.y *= r2; s.z *= r2; s.w *= r2;
        const float a0 = (s.x / (1.0f + expf(-s.x))) * s.y;   // rows are (gate, up, gate, up): silu(g) * u
        const float a1 = (s.z / (1.0f + expf(-s.z))) * s.w;
        eng_gstore(gb + gE + (n >> 1) + 0, tl + 4, __float_as_uint(a0));
        eng_gstore(gb + gE + (n >> 1) + 1, tl + 4, __float_as_uint(a1));
      }
    }

    // ================= edge E: act -> down CUs =================
    if (np[EPH_DOWN] > 0) {
      const bool last = layer + 1 == p.layers;
      float4 gn = make_float4(0.f, 0.f, 0.f, 0.f);
      int dn = 0;
      if (!loader) {
        if (tid < np[EPH_DOWN]) {
          dn = 4 * (int)pl->parts[p0[EPH_DOWN] + tid];
          if (last) gn = *(const float4*)(p.final_norm + dn);
        }
        constexpr int EPT = 12;   // granules per thread in flight in one sweep (12 x 448 covers intermediate sizes up to 5376 in one pass)
        for (int e0 = 0; e0 < I; e0 += EPT * kEngGather) {
          int idx[EPT];
          unsigned v[EPT];
#pragma unroll
          for (int k = 0; k < EPT; ++k) { const int e = e0 + tid + k * kEngGather; idx[k] = e < I ? gE + e : -1; }
          ok = eng_sweep_idx<EPT>(gb, idx, tl + 4, v, sy, (unsigned)(layer * 8 + 5)) && ok;
#pragma unroll
          for (int k = 0; k < EPT; ++k) {
            const int e = e0 + tid + k * kEngGather;
            if (e < I) eng_put3(xs_act, eng_xs_elem(e >> 5, e & 31), __uint_as_float(v[k]));
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        ld_trickle();
        ld_ensure(layer, EPH_DOWN);
      }
      ok = bar_chk(ok);
      if (stamp) sp[4] = __builtin_amdgcn_s_memrealtime();
      if (!ok) break;
      run_jobs(layer, EPH_DOWN, xs_act);
      bar();
      if (stamp) sp[7] = __builtin_amdgcn_s_memrealtime();
      if (loader) ld_consumed(EPH_DOWN);
      if (tid < np[EPH_DOWN]) {
        const float4 s = finish(EPH_DOWN, tid);
        float4 h4 = *(const float4*)(hbuf + dn);   // h_mid
        h4.x += s.x; h4.y += s.y; h4.z += s.z; h4.w += s.w;
        if (!last) {
          eng_gstore(gnext + gA + dn + 0, tl + 8, __float_as_uint(h4.x));
          eng_gstore(gnext + gA + dn + 1, tl + 8, __float_as_uint(h4.y));
          eng_gstore(gnext + gA + dn + 2, tl + 8, __float_as_uint(h4.z));
          eng_gstore(gnext + gA + dn + 3, tl + 8, __float_as_uint(h4.w));
        } else {   // the RESID epilogue's outputs for lm_head: h, triples of final_norm * h, partial sums of squares
          *(float4*)(p.h + dn) = h4;
          p.ss_out[dn >> 2] = (h4.x * h4.x + h4.y * h4.y) + (h4.z * h4.z + h4.w * h4.w);
          const float t[4] = {gn.x * h4.x, gn.y * h4.y, gn.z * h4.z, gn.w * h4.w};
          uint32_t hi[4], mi[4], lo[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split3(t[e], hi[e], mi[e], lo[e]);
          unsigned char* o = p.xs_out + (size_t)(dn >> 5) * 192 + ((dn >> 3) & 3) * 16 + ((dn >> 2) & 1) * 8;
          *(uint2*)(o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
          *(uint2*)(o + 64) = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
          *(uint2*)(o + 128) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of the weight stream is in flight when the LDS is given back
  __syncthreads();
  if (tid == 0) {   // the last workgroup to leave opens the next launch's epoch (every workgroup has read `serial` long before)
    const unsigned done = atomicAdd(p.arrive, 1u);
    if (done == (unsigned)p.ncu - 1) { *p.arrive = 0u; __threadfence(); atomicAdd(p.serial, 1u); }
  }
}

}  // namespace
