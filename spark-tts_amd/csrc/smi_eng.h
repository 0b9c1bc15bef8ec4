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
struct EngWavePlan { uint16_t jstart[5]; uint16_t pad[3]; uint16_t pimg[4]; EngJob jobs[kEngMaxJobs]; };   // jobs at byte 24: 8-byte aligned
struct EngCuPlan { uint16_t pstart[5]; int16_t head; uint16_t parts[kEngMaxParts]; uint16_t len_cu, pad; uint16_t pcount[4]; EngWavePlan w[kEngWaves]; };
static_assert(sizeof(EngJob) == 8 && sizeof(EngWavePlan) % 8 == 0 && sizeof(EngCuPlan) % 8 == 0, "plan layout");

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
  unsigned long long* stamps;   // diagnostics: [3][layers][16] or null
  int poll_quiet;               // hand-off sweeps: 1 = watch one granule first, 0 = sweep from the start, 2 = timed first sweep
  int edge_delay[5];            // mode 2: pause (x 64 cycles) before the first sweep of hand-off A (h), B, C, D, E
  int ld_burst, ld_sleep;       // loader pacing: fills per look at the arrival counter, pause (x 64 cycles) between bursts
};

// ---- LDS carve (bytes), all multiples of 16
struct EngLds { int ring, xs_act, xs_x, hbuf, ssp, red, cnt, plan, total; };
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
  l.cnt = o; o += kEngMaxSlots * 4 + kEngMaxSlots * 16 * 3;   // completion counters; QKV bias, rope pair, final-norm weights per part
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
  unsigned* cnt = (unsigned*)(smem + L.cnt);      // [slot] chain sets of the part finished in the phase under way
  float4* biasL = (float4*)(smem + L.cnt + kEngMaxSlots * 4);   // [slot] QKV bias of the part's 4 rows (this layer)
  float4* ropeL = biasL + kEngMaxSlots;           // [slot] (cos, sin) of the part's two rotation pairs at this step's position
  float4* gnL = ropeL + kEngMaxSlots;             // [slot] final-norm weights of the part's rows (down_proj parts)
  EngCuPlan* pl = (EngCuPlan*)(smem + L.plan);
  unsigned* arrive = (unsigned*)(smem + L.plan + ((int)sizeof(EngCuPlan) + 15) / 16 * 16);   // consumer waves arrived at barriers so far
  // LDS-DMA takes the ABSOLUTE LDS address in M0: the dynamic segment starts behind the kernel's static LDS (none today; kept exact)
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint32_t lbase = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lptr_t)smem);

  EngSync sy;
  sy.err = p.err;
  sy.t_end = __builtin_amdgcn_s_memrealtime() + p.timeout_ticks;
  sy.quiet = p.poll_quiet;
  sy.predelay = 0;
  {   // this CU's plan -> LDS
    const uint32_t* src = (const uint32_t*)(p.plan + cu);
    for (int i = tid; i < (int)(sizeof(EngCuPlan) / 4); i += kEngBlock) ((uint32_t*)pl)[i] = src[i];
    if (tid == 0) { arrive[0] = 0u; arrive[1] = 0u; arrive[2] = 0u; arrive[3] = 0u; }
  }
  sy.go = arrive + 2;          // the epoch whose first granule this workgroup's watching wave has seen
  sy.watcher = wave == 0;
  // the RMSNorm factor of the phase under way: computed once, by the wave that did not sweep, right behind the barrier that
  // completes the partial sums; whoever runs an epilogue (after its jobs) reads it -- rms_get waits for the epoch word, which
  // in practice is long there
  unsigned* rtag = arrive + 3;
  unsigned* rfac = arrive + 4;
  auto rms_put = [&](unsigned epoch, int K) {
    const float r = eng_rms_from_ssp(ssp, p.H / 4, lane, K, p.eps);
    if (lane == 0) { eng_lds_store(rfac, __float_as_uint(r)); eng_lds_store(rtag, epoch); }   // (a wave's LDS operations are performed in order)
  };
  auto rms_get = [&](unsigned epoch) -> float {
    for (unsigned it = 0; eng_lds_load(rtag) != epoch && it < (1u << 22); ++it) __builtin_amdgcn_s_sleep(1);
    return __uint_as_float(eng_lds_load(rfac));
  };
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
  unsigned* failw = arrive + 1;
  auto bar_chk = [&](bool okv) -> bool {
    if (!loader && lane == 0) atomicAdd(arrive, 1u);
    if (!okv) eng_lds_store(failw, 1u);
    ++bars;
    __syncthreads();
    return __builtin_amdgcn_readfirstlane((int)eng_lds_load(failw)) == 0;   // (uniform by construction; said so, or every loop-carried scalar of the layer loop turns into a vector register)
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
  const unsigned char* cu_stream = p.stream + (size_t)cu * (size_t)p.maxlen * 1024 + lane * 16;   // per lane: its 16 bytes of an image
  const size_t layer_imgs_bytes = (size_t)p.ncu * (size_t)p.maxlen * 1024;
  // loader state, all wave-uniform (kept in scalar registers: every loop condition below is made uniform explicitly)
  int issued_total = 0, consumed_total = 0, is_layer = 0, is_idx = 0, is_slot = 0;
  size_t is_off = 0;   // byte offset of the next image from cu_stream
  auto ld_can = [&]() -> bool { return is_layer < p.layers && issued_total - consumed_total < kEngRingTotal; };
  auto ld_issue = [&]() {       // precondition: ld_can()
    smi_glds16(cu_stream + is_off, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lbase + (uint32_t)L.ring + (uint32_t)is_slot * 1024)));
    ++issued_total;
    is_off += 1024;
    if (++is_slot == kEngRingTotal) is_slot = 0;
    if (++is_idx == len_cu) { is_idx = 0; ++is_layer; is_off = (size_t)is_layer * layer_imgs_bytes; }
  };
  // Until the last sweeping wave has arrived at barrier number `bars + 1`: keep the ring full (p.ld_burst fills per look
  // at the arrival counter; a pause of p.ld_sleep x 64 cycles between bursts thins the stream beside the sweeps).
  auto ld_trickle = [&]() {
    const unsigned target = (bars + 1) * (kEngWaves - 1);
    const int burst = p.ld_burst, pause = p.ld_sleep;
    for (unsigned it = 0;; ++it) {
      const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)eng_lds_load(arrive));
      if (a >= target) break;
      bool any = false;
      if (len_cu > 0)
        for (int b = 0; b < burst && ld_can(); ++b) { ld_issue(); any = true; }
      if (!any) __builtin_amdgcn_s_sleep(2);
      for (int z = 0; z < pause; ++z) __builtin_amdgcn_s_sleep(1);
      if ((it & 4095u) == 4095u && __builtin_amdgcn_s_memrealtime() > sy.t_end + 100000000ull) break;   // (the consumers' spins are bounded; this bounds the loader's too)
    }
  };
  // Everything up to the end of (layer, phase) issued and landed (called before the barrier that opens that phase's compute).
  auto ld_ensure = [&](int layer, int ph) {
    const int target = layer * len_cu + pre_ph[ph];
    while (issued_total < target && ld_can()) ld_issue();   // fits: a CU's images of one phase never exceed the ring
    const int k = issued_total - target;   // fills issued behind the last one needed may stay in flight (they land in order)
    // wait until at most k fills are outstanding, exactly (the count is an immediate: one case per value)
#define SMI_VMW(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
    switch (k < 63 ? k : 63) {
      SMI_VMW(0) SMI_VMW(1) SMI_VMW(2) SMI_VMW(3) SMI_VMW(4) SMI_VMW(5) SMI_VMW(6) SMI_VMW(7) SMI_VMW(8) SMI_VMW(9)
      SMI_VMW(10) SMI_VMW(11) SMI_VMW(12) SMI_VMW(13) SMI_VMW(14) SMI_VMW(15) SMI_VMW(16) SMI_VMW(17) SMI_VMW(18) SMI_VMW(19)
      SMI_VMW(20) SMI_VMW(21) SMI_VMW(22) SMI_VMW(23) SMI_VMW(24) SMI_VMW(25) SMI_VMW(26) SMI_VMW(27) SMI_VMW(28) SMI_VMW(29)
      SMI_VMW(30) SMI_VMW(31) SMI_VMW(32) SMI_VMW(33) SMI_VMW(34) SMI_VMW(35) SMI_VMW(36) SMI_VMW(37) SMI_VMW(38) SMI_VMW(39)
      SMI_VMW(40) SMI_VMW(41) SMI_VMW(42) SMI_VMW(43) SMI_VMW(44) SMI_VMW(45) SMI_VMW(46) SMI_VMW(47) SMI_VMW(48) SMI_VMW(49)
      SMI_VMW(50) SMI_VMW(51) SMI_VMW(52) SMI_VMW(53) SMI_VMW(54) SMI_VMW(55) SMI_VMW(56) SMI_VMW(57) SMI_VMW(58) SMI_VMW(59)
      SMI_VMW(60) SMI_VMW(61) SMI_VMW(62) SMI_VMW(63)
    }
#undef SMI_VMW
  };
  auto ld_consumed = [&](int ph) { consumed_total += P_ph[ph]; };   // behind the barrier that closes a phase's compute

  // ---- per-lane constants of the MFMA B operand: column col = 3 s + c of segment s, split term c
  const int col = lane & 15, k8 = lane >> 4;
  const int bs = col < 12 ? col / 3 : 3, bc = col < 12 ? col % 3 : 2;
  const int boff = (bc * 4 + k8) * 16;

  // one phase's jobs of this wave: operand image xs (tiles of 192 bytes), results to red[slot][chain].  Returns, per lane
  // j < (jobs of this wave in the phase), the part slot whose LAST chain set this wave finished (-1: none): that lane then
  // runs the part's epilogue -- no barrier and no single epilogue wave between the last MFMA and the hand-off.
  auto run_jobs = [&](int layer, int ph, const unsigned char* xs) -> int {
    const int KT = p.KT[ph], NW = p.NW[ph];
    const int j0 = __builtin_amdgcn_readfirstlane((int)wp.jstart[ph]), j1 = __builtin_amdgcn_readfirstlane((int)wp.jstart[ph + 1]);
    // the phase's job descriptors: lane j reads job j once; the loop takes them from there (no LDS round trip per job)
    const uint2 jd = *(const uint2*)&wp.jobs[j0 + (lane < j1 - j0 ? lane : 0)];
    for (int j = j0; j < j1; ++j) {
      const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)jd.x, j - j0), d1 = (unsigned)__builtin_amdgcn_readlane((int)jd.y, j - j0);
      const int set = (int)((d0 >> 16) & 0xffu), nimg = (int)(d0 >> 24);
      const int slot = (int)(d1 & 0xffffu);
      const int img0 = layer * len_cu + (int)(d1 >> 16);   // launch-wide number of the job's first image
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      int T = 4 * set + bs;
      auto lda = [&](int i) -> bf16x8 { return *(const bf16x8*)(smem + (uint32_t)L.ring + (uint32_t)((img0 + i) % kEngRingTotal) * 1024 + lane * 16); };
      auto ldb = [&](int Tt) -> bf16x8 { return *(const bf16x8*)(xs + (Tt < KT ? Tt : KT - 1) * 192 + boff); };
      // four images' operands are requested together, then their four MFMAs run: one LDS round trip per four images (the
      // scheduler barrier keeps hipcc from folding this back into load -> wait -> MFMA per image)
      for (int i0 = 0; i0 < nimg; i0 += 4) {
        bf16x8 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int iu = i0 + u < nimg ? i0 + u : nimg - 1;
          a[u] = lda(iu);
          b[u] = ldb(T + (iu - i0) * NW);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i0 + u < nimg) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b[u], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        T += 4 * NW;
      }
      // chain 4 set + s: hi / mid / lo accumulators are columns 3 s, 3 s + 1, 3 s + 2 of rows 4 s .. 4 s + 3 (lanes 19 s + c)
      f32x4 t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // lane i takes lanes i + 1 / i + 2 of its row of 16 (DPP row_shl: no LDS crossbar trip)
        const float mid = smi_dpp<0x101>(acc[r]), lo = smi_dpp<0x102>(acc[r]);
        t[r] = (lo + mid) + acc[r];
      }
      const int s = lane / 19;
      if (lane == 19 * s && s < 4 && 4 * set + s < NW) red[slot * 16 + 4 * set + s] = make_float4(t[0], t[1], t[2], t[3]);
    }
    // lane j reports job j (LDS operations of a wave are performed in order: the sums above are in place before the count)
    int mine = -1;
    if (lane < j1 - j0) {
      const int slot = (int)(jd.y & 0xffffu);
      const unsigned G = (unsigned)(NW + 3) / 4;
      if (atomicAdd(cnt + slot, 1u) == G - 1) mine = slot;
    }
    return mine;
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
  // (`pre`: this thread's norm weights, requested a phase ahead -- a cold load in front of the sweep would hold the sweep's
  // loads back by its own HBM round trip: a wave's memory operations return in order)
  const bool pre_ok = H / 4 <= kEngGather;   // one pass: thread j owns elements 4j .. 4j+3 in every hidden-size hand-off
  auto stage_h = [&](const smi_u64* gbuf, int goff, unsigned tag, const float* gamma, const float4& pre, unsigned where) -> bool {
    bool okv = true;
    for (int j0 = 0; j0 < H / 4; j0 += kEngGather) {
      const int j = j0 + tid;
      const bool act = j < H / 4;
      const int jc = act ? j : 0;
      const float4 gam = pre_ok ? pre : *(const float4*)(gamma + 4 * jc);
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
  // per-part constants that do not depend on the layer: the rotation factors of this CU's q / k parts at this step's
  // position, the final norm's weights of its down_proj rows -- in LDS, for whichever lane runs the part's epilogue
  if (tid < np[EPH_QKV]) {
    const int qn = 4 * (int)pl->parts[p0[EPH_QKV] + tid];
    float4 rp = make_float4(1.f, 0.f, 1.f, 0.f);
    if (qn < Q + KVd) {
      const int i0 = (qn & 63) >> 1;
      const float2 r0 = p.rope[(size_t)pos * 32 + i0], r1 = p.rope[(size_t)pos * 32 + i0 + 1];
      rp = make_float4(r0.x, r0.y, r1.x, r1.y);
    }
    ropeL[tid] = rp;
  }
  if (tid < np[EPH_DOWN]) gnL[tid] = *(const float4*)(p.final_norm + 4 * (int)pl->parts[p0[EPH_DOWN] + tid]);
  const int kvh = head >= 0 ? head / (p.n_heads / p.n_kv) : 0;
  const bool need_h = np[EPH_QKV] > 0 || np[EPH_O] > 0;        // (the o_proj epilogue adds its rows of h)
  const bool need_hmid = np[EPH_GU] > 0 || np[EPH_DOWN] > 0;   // (the down_proj epilogue adds its rows of h_mid)
  bool ok = true;
  // norm weights and QKV bias of the first layer (later layers: requested one phase ahead, below)
  const int jpre = tid < H / 4 ? tid : 0;
  float4 g1_pre = *(const float4*)((const float*)(p.arena + p.layers_base + p.off_ln1) + 4 * jpre), g2_pre = g1_pre;
  float4 bq_pre = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < np[EPH_QKV]) bq_pre = *(const float4*)((const float*)(p.arena + p.layers_base + p.off_bqkv) + 4 * (int)pl->parts[p0[EPH_QKV] + tid]);
  for (int layer = 0; layer < p.layers; ++layer) {
    const unsigned char* lay = p.arena + p.layers_base + (size_t)layer * p.layer_stride;
    smi_u64* gb = p.gran + (size_t)(layer & 1) * gran_per_buf;
    smi_u64* gnext = p.gran + (size_t)((layer + 1) & 1) * gran_per_buf;
    const unsigned tl = tbase + (unsigned)layer * 8u;
    // diagnostics: rows 0 / 2 = waves 0 / 7 of CU 0, row 1 = wave 0 of the first head CU
    const bool stamp = p.stamps && ((tid == 0 && (cu == 0 || head == 0)) || (tid == kEngGather && cu == 0));
    unsigned long long* sp = p.stamps ? p.stamps + ((size_t)(cu == 0 ? (tid == 0 ? 0 : 2) : 1) * p.layers + layer) * 16 : nullptr;
    constexpr int UNR = 4, DPL = 8, NGRP = kEngWaves * 8;
    const int tl8 = lane >> 3, dl = lane & 7, grp = wave * 8 + tl8;
    uint4 kr[UNR], vr[UNR];
    const uint16_t* kc = p.kcache + (size_t)layer * p.kv_layer_elems;
    const uint16_t* vc = p.vcache + (size_t)layer * p.kv_layer_elems;
    const size_t rowbase = (size_t)kvh * p.max_pos;   // slot 0
    // ---- the epilogues of the four GEMV phases (k_gemm's, value for value): run by ONE lane per part, in whichever wave
    //      finished the part's last chain set
    auto epi_qkv = [&](int slot, float r1) {
      const int qn = 4 * (int)pl->parts[p0[EPH_QKV] + slot];
      float4 s = finish(EPH_QKV, slot);
      s.x *= r1; s.y *= r1; s.z *= r1; s.w *= r1;
      const float4 bq = biasL[slot];
      s.x += bq.x; s.y += bq.y; s.z += bq.z; s.w += bq.w;
      if (qn < Q + KVd) {
        const float4 rp = ropeL[slot];
        float4 r;
        r.x = __fadd_rn(__fmul_rn(s.x, rp.x), __fmul_rn(-s.y, rp.y));
        r.y = __fadd_rn(__fmul_rn(s.y, rp.x), __fmul_rn(s.x, rp.y));
        r.z = __fadd_rn(__fmul_rn(s.z, rp.z), __fmul_rn(-s.w, rp.w));
        r.w = __fadd_rn(__fmul_rn(s.w, rp.z), __fmul_rn(s.z, rp.w));
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
    };
    auto epi_o = [&](int slot) {
      const int n = 4 * (int)pl->parts[p0[EPH_O] + slot];
      const float4 s = finish(EPH_O, slot);
      float4 h4 = *(const float4*)(hbuf + n);
      h4.x += s.x; h4.y += s.y; h4.z += s.z; h4.w += s.w;
      eng_gstore(gb + gD + n + 0, tl + 3, __float_as_uint(h4.x));
      eng_gstore(gb + gD + n + 1, tl + 3, __float_as_uint(h4.y));
      eng_gstore(gb + gD + n + 2, tl + 3, __float_as_uint(h4.z));
      eng_gstore(gb + gD + n + 3, tl + 3, __float_as_uint(h4.w));
    };
    auto epi_gu = [&](int slot, float r2) {
      const int n = 4 * (int)pl->parts[p0[EPH_GU] + slot];
      float4 s = finish(EPH_GU, slot);
      s.x *= r2; s.y *= r2; s.z *= r2; s.w *= r2;
      const float a0 = (s.x / (1.0f + expf(-s.x))) * s.y;   // rows are (gate, up, gate, up): silu(g) * u
      const float a1 = (s.z / (1.0f + expf(-s.z))) * s.w;
      eng_gstore(gb + gE + (n >> 1) + 0, tl + 4, __float_as_uint(a0));
      eng_gstore(gb + gE + (n >> 1) + 1, tl + 4, __float_as_uint(a1));
    };
    auto epi_down = [&](int slot) {
      const bool last = layer + 1 == p.layers;
      const int dn = 4 * (int)pl->parts[p0[EPH_DOWN] + slot];
      const float4 s = finish(EPH_DOWN, slot);
      float4 h4 = *(const float4*)(hbuf + dn);   // h_mid
      h4.x += s.x; h4.y += s.y; h4.z += s.z; h4.w += s.w;
      if (!last) {
        eng_gstore(gnext + gA + dn + 0, tl + 8, __float_as_uint(h4.x));
        eng_gstore(gnext + gA + dn + 1, tl + 8, __float_as_uint(h4.y));
        eng_gstore(gnext + gA + dn + 2, tl + 8, __float_as_uint(h4.z));
        eng_gstore(gnext + gA + dn + 3, tl + 8, __float_as_uint(h4.w));
      } else {   // the RESID epilogue's outputs for lm_head: h, triples of final_norm * h, partial sums of squares
        const float4 gn = gnL[slot];
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
    };

    // ================= edge A: h -> QKV =================
    if (!loader) {
      if (tid < kEngMaxSlots) cnt[tid] = 0u;
      if (tid < np[EPH_QKV]) biasL[tid] = bq_pre;
      if (need_h) {
        sy.predelay = p.edge_delay[0];
        ok = stage_h(gb, layer == 0 ? -1 : gA, tl + 0, (const float*)(lay + p.off_ln1), g1_pre, (unsigned)(layer * 8 + 1)) && ok;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      ld_trickle();
      ld_ensure(layer, EPH_QKV);
    }
    ok = bar_chk(ok);
    if (stamp) sp[0] = __builtin_amdgcn_s_memrealtime();
    if (stamp && loader) sp[14] = __builtin_amdgcn_s_memtime();   // (row 2: shader cycles at the same point: the clock the launch runs at)
    if (!ok) break;
    // what the NEXT hand-offs' stagers need from global memory is requested now (cold lines: a phase of cover): this layer's
    // post-attention norm weights, the next layer's input norm weights and QKV bias
    if (!loader) {
      g2_pre = *(const float4*)((const float*)(lay + p.off_ln2) + 4 * jpre);
      if (layer + 1 < p.layers) {
        g1_pre = *(const float4*)((const float*)(lay + p.layer_stride + p.off_ln1) + 4 * jpre);
        if (tid < np[EPH_QKV]) bq_pre = *(const float4*)((const float*)(lay + p.layer_stride + p.off_bqkv) + 4 * (int)pl->parts[p0[EPH_QKV] + tid]);
      }
    }
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
      if (loader) rms_put(tl + 0, p.KT[EPH_QKV] * 32);
      if (stamp) sp[8] = __builtin_amdgcn_s_memrealtime();
      const int mine = run_jobs(layer, EPH_QKV, xs_x);
      if (stamp) sp[9] = __builtin_amdgcn_s_memrealtime();
      if (mine >= 0) epi_qkv(mine, rms_get(tl + 0));
      if (stamp) sp[10] = __builtin_amdgcn_s_memrealtime();
    }
    bar();   // the operand image, the sums and the ring slots of this phase are free again
    if (stamp) sp[5] = __builtin_amdgcn_s_memrealtime();
    if (loader) ld_consumed(EPH_QKV);

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
          sy.predelay = p.edge_delay[1];
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
        if (tid < kEngMaxSlots) cnt[tid] = 0u;
        for (int e0 = 0; e0 < Q; e0 += 2 * kEngGather) {
          const int e[2] = {e0 + tid, e0 + tid + kEngGather};
          unsigned v[2];
          const int idx[2] = {e[0] < Q ? gC + e[0] : -1, e[1] < Q ? gC + e[1] : -1};
          sy.predelay = p.edge_delay[2];
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
      {
        const int mine = run_jobs(layer, EPH_O, xs_x);
        if (mine >= 0) epi_o(mine);
      }
      bar();
      if (loader) ld_consumed(EPH_O);
    }

    // ================= edge D: h_mid -> gate_up =================
    if (!loader) {
      if (tid < kEngMaxSlots) cnt[tid] = 0u;
      if (need_hmid) {
        sy.predelay = p.edge_delay[3];
        ok = stage_h(gb, gD, tl + 3, (const float*)(lay + p.off_ln2), g2_pre, (unsigned)(layer * 8 + 4)) && ok;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      ld_trickle();
      ld_ensure(layer, EPH_GU);
    }
    ok = bar_chk(ok);
    if (stamp) sp[3] = __builtin_amdgcn_s_memrealtime();
    if (!ok) break;
    if (np[EPH_GU] > 0) {
      if (loader) rms_put(tl + 3, p.KT[EPH_GU] * 32);
      if (stamp) sp[11] = __builtin_amdgcn_s_memrealtime();
      const int mine = run_jobs(layer, EPH_GU, xs_x);
      if (stamp) sp[12] = __builtin_amdgcn_s_memrealtime();
      if (mine >= 0) epi_gu(mine, rms_get(tl + 3));
      if (stamp) sp[13] = __builtin_amdgcn_s_memrealtime();
    }
    bar();
    if (stamp) sp[6] = __builtin_amdgcn_s_memrealtime();
    if (loader) ld_consumed(EPH_GU);

    // ================= edge E: act -> down CUs =================
    if (np[EPH_DOWN] > 0) {
      if (!loader) {
        if (tid < kEngMaxSlots) cnt[tid] = 0u;
        // thread t takes the quads t, t + 448, t + 896 of four consecutive act values (12 granules in flight in one sweep:
        // intermediate sizes up to 5376 in one pass); a quad's exact triples are one 8-byte LDS store per split term
        constexpr int QPT = 3;
        for (int q0 = 0; q0 < I / 4; q0 += QPT * kEngGather) {
          int idx[4 * QPT];
          unsigned v[4 * QPT];
#pragma unroll
          for (int k = 0; k < QPT; ++k) {
            const int q = q0 + tid + k * kEngGather;
#pragma unroll
            for (int e = 0; e < 4; ++e) idx[4 * k + e] = q < I / 4 ? gE + 4 * q + e : -1;
          }
          sy.predelay = p.edge_delay[4];
          ok = eng_sweep_idx<4 * QPT>(gb, idx, tl + 4, v, sy, (unsigned)(layer * 8 + 5)) && ok;
          if (stamp) sp[14] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
          for (int k = 0; k < QPT; ++k) {
            const int q = q0 + tid + k * kEngGather;
            if (q < I / 4) {
              uint32_t hi[4], mi[4], lo[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) split3(__uint_as_float(v[4 * k + e]), hi[e], mi[e], lo[e]);
              unsigned char* o = xs_act + eng_xs_elem((4 * q) >> 5, (4 * q) & 31);
              *(uint2*)(o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
              *(uint2*)(o + 64) = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
              *(uint2*)(o + 128) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
            }
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (stamp) sp[15] = __builtin_amdgcn_s_memrealtime();
      } else {
        ld_trickle();
        ld_ensure(layer, EPH_DOWN);
      }
      ok = bar_chk(ok);
      if (stamp) sp[4] = __builtin_amdgcn_s_memrealtime();
      if (!ok) break;
      {
        const int mine = run_jobs(layer, EPH_DOWN, xs_act);
        if (mine >= 0) epi_down(mine);
      }
      bar();
      if (stamp) sp[7] = __builtin_amdgcn_s_memrealtime();
      if (loader) ld_consumed(EPH_DOWN);
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
