// smi_llm.hip -- Qwen2 speech-token generator for gfx950 (MI355X).
//
// Stands behind AutoModelForCausalLM.generate(..., do_sample=False) as called at
// cli/SparkTTS.py:197-204 of the reference; the arithmetic restated is transformers'
// modeling_qwen2.py (RMSNorm :247-252, attention :150-235, RoPE :91-135, MLP :46-48).
//
// One "step" pushes M <= 32 rows through the model.  A row is (sequence slot, position, token):
// in decode the rows are the B live sequences, in prefill they are 32 prompt tokens.  Every row is
// independent in every kernel, so a batched step is bit-identical to B single-sequence steps.
//
// Kernels per layer (HBM-bound weight streaming; weights read exactly once per step):
//   k_gemm<QKV>   RMSNorm prologue -> [q|k|v] projection -> +bias -> RoPE -> q buffer / KV-cache append
//   k_attn        GQA attention over the cached keys (online softmax, wave64 shuffles + LDS combine)
//   k_gemm<RESID> o_proj, residual add in the epilogue
//   k_gemm<SWIGLU> RMSNorm prologue -> [gate|up] (row-interleaved) -> silu(g)*u
//   k_gemm<RESID> down_proj, residual add
// then k_gemm<LM> (final RMSNorm -> tied lm_head -> per-block argmax) and k_finalize (argmax over
// blocks, token/position bookkeeping, next step's embedding gather).
//
// The GEMM core: weights are bf16 in 16x32 tiles stored in MFMA A-operand order (one 1 KiB tile =
// one coalesced 16-B-per-lane wave load = one v_mfma_f32_16x16x32_bf16 operand).  Activations stay
// fp32-exact: every GEMM operand is kept in HBM/L2 as three bf16 terms (hi+mid+lo, 8+8+8 mantissa
// bits) in B-operand order, written by the epilogue of the kernel that produced the value (with the
// next RMSNorm's weight folded in and the row's partial sums of squares beside it), so consumers
// only stream.  Every product is exact and only the fp32 summation order differs from the CPU
// oracle.  MFMA columns are the rows m of the step.
#include "smi_common.h"
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr int kMaxRows = SMI_MAX_ROWS;
constexpr int kHeadDim = 64;

struct RowDesc {  // 16 bytes, lives in device memory
  int32_t slot, pos, token, flags;
};

// Generation controls in device memory (written at prefill / session_begin / admit; read by k_finalize and the sampler),
// so that nothing of them is baked into the captured decode graph: the graph survives from one utterance to the next.
struct Ctl {
  long long eos[SMI_MAX_EOS];     // a sequence stops counting after emitting any of the first n_eos ids
  int32_t n_eos, pad;
  unsigned long long seed;        // sampler stream key
  int32_t seqid[SMI_MAX_ROWS];    // per KV slot: admission number of the sequence living there (sampler stream key)
};

struct KvMap {
  const int32_t* ptab;   // [slots][ppslot] page ids, or null
  int pshift, ppslot;
};

enum { PRO_PLAIN = 0, PRO_NORM = 1, PRO_FUSEDO = 2 };   // FUSEDO: RMSNorm operand built in the kernel from the fused o_proj's per-head partials
enum { EPI_RESID = 0, EPI_SWIGLU = 1, EPI_QKV = 2, EPI_LM = 3 };

// Same-XCD L2 prefetch for a LATER kernel: block b of every kernel of a graph lands on XCD (s + b) mod 8
// (measured: strict round-robin with the same start for every kernel, tools/xccmap.py), so a helper
// block with blockIdx == c (mod 8) warms exactly the weight slices that consumer blocks == c (mod 8)
// will read: those then hit the XCD's own 4 MiB L2 (a 17 MB stream: 5 us cold, 2 us L2-warm).  Speed
// only -- nothing depends on the placement.
struct PfDesc {
  const unsigned char* base;   // consumer weight matrix
  int slice_bytes;             // bytes consumer block b reads: [b * slice_bytes, (b + 1) * slice_bytes)
  int nslices;                 // consumer work blocks
  int q0, q1;                  // this producer covers slices s with q0 <= s / 8 < q1
};

__device__ __forceinline__ void pf_run(const PfDesc& d, int helper, int nhelpers, int tid, int nthreads) {
  const int c = (int)blockIdx.x & 7, r = helper >> 3, R = (nhelpers + 7) >> 3;
  for (int q = d.q0 + r; q < d.q1; q += R) {
    const int sl = q * 8 + c;
    if (sl < d.nslices) smi_prefetch_range(d.base + (size_t)sl * d.slice_bytes, (size_t)d.slice_bytes, tid, nthreads);
  }
}

struct GemmP {
  const uint4* W;        // [NT][KT][64] 16-byte lane pieces (bf16 tiles in MFMA A-operand order)
  int NT, KT, M;         // n tiles, k tiles, rows
  const RowDesc* rows;
  const unsigned char* XS;   // operand: exact bf16 triples [KT][3][4][M][16 B] in MFMA B-operand order
  const float* sspart;   // PRO_NORM: [rows][npart] partial sums of squares of the operand's fp32 source row
  int npart;
  float eps;
  float* Y;              // RESID: h [M][N]; QKV: q [M][q_dim]; LM: logits [M][V] or null
  const float* bias;     // QKV bias [N] (arena order)
  const float2* rope;    // [max_pos][32] (cos, sin)
  void* kcache;          // this layer's K cache: [slot][kvh][max_pos][64]
  void* vcache;
  int q_dim, kv_dim, n_kv, max_pos;
  KvMap km;              // QKV: paged KV cache (ptab null: contiguous slots)
  int V;                 // LM: true vocab size
  float* pval;           // LM: [rows][work_blocks] per-block best logit
  int* pidx;
  // producer side: the next consumer's operand, written by the epilogue
  unsigned char* XSout;  // RESID: triples of gamma_next * h_new, [N/32][3][4][M][16]; SWIGLU: triples of act
  const float* gamma_next;  // RESID: [N] RMSNorm weight of the NEXT norm
  float* ssout;          // RESID: [rows][NT * 4] partial sums of squares of h_new, one per 4 columns (k_pgemm: [rows][NT])
  unsigned long long* stamps;
  int work_blocks;       // blocks >= work_blocks are prefetch helpers (pf)
  PfDesc pf;
  int ldsb;              // few rows: bytes of wave-private LDS for the activation triples (0 = read them from global per tile)
  int lt_shift;          // log2(lanes that fetch one k tile's 12*M pieces)
  // fused o_proj (one row): PRO_FUSEDO builds its operand from these instead of reading XS / sspart; RESID reads its residual from Yin
  const float* part_o;   // [n_oheads][K] per-head partials of the o_proj (k_attn<.., FUSE>)
  int n_oheads;
  const float* hres;     // [K] residual stream before the attention block's contribution
  const float* gam;      // [K] this RMSNorm's weight
  float* h2out;          // [K] block 0 leaves h + o_proj here (the residual down_proj adds to)
  const float* Yin;      // RESID: residual row source when it is not Y itself (null: Y)
  const unsigned char* pf2_base; int pf2_slice, pf2_nslices;   // PRO_FUSEDO: a later kernel's weights the work blocks touch (null: none)
  int wperm;             // weight tiles stored row-part-major [q:4][k8:4][r:4][8] (W_down): lane (k8, n = 4q + r) owns piece q*16 + k8*4 + r
  // k_pgemm<RESID>: the K range is summed in SEGMENTS of kseg k tiles (segment sums added in order) -- the association both the
  // in-block form (many rows: a second accumulator set) and the split-K form (few rows: one block per segment -> `slab`
  // [segment][n tile][m tile][lane] f32x4, combined in order by k_resid_comb) produce, so a prompt row has the same bits in
  // either.  kseg = 0: one running sum over all of K (the other GEMMs).
  int kseg;
  float4* slab;
  int slab_mt;           // m-tiles of the whole launch (slab indexing)
};
// piece index of `lane` inside a 1 KiB weight tile (see GemmP::wperm)
__device__ __forceinline__ int smi_wlane(int lane, int wperm) {
  return wperm ? ((lane & 12) << 2) + ((lane >> 4) << 2) + (lane & 3) : lane;
}
constexpr int kMaxOHeads = 16;

// One weight tile piece (16 B per lane) of a once-read decode weight stream: non-temporal policy (MI355X_MICROARCH.md, row
// nt-weights).  Measured as separate builds on one box: decode step 627.9 -> 625.3 us at one row (two alternating pairs);
// -DSMI_W_PLAIN builds the plain-load variant for A/B (make variant NAME=plain VARFLAGS=-DSMI_W_PLAIN).
__device__ __forceinline__ uint4 smi_ldw(const uint4* q) {
#ifndef SMI_W_PLAIN
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)q);
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *q;
#endif
}

// exact 3-way bf16 split: x == hi + mid + lo (8 + 8 + 8 mantissa bits)
__device__ __forceinline__ void split3(float x, uint32_t& hi, uint32_t& mi, uint32_t& lo) {
  hi = smi_f32_to_bf16(x);
  const float r1 = x - smi_bf16_to_f32(hi);
  mi = smi_f32_to_bf16(r1);
  const float r2 = r1 - smi_bf16_to_f32(mi);
  lo = smi_f32_to_bf16(r2);
}
// byte offset of the 16-byte piece (k tile kt, split s, octet k8, row m) in a triples buffer
__device__ __forceinline__ size_t xs_off(int kt, int s, int k8, int m, int M) {
  return ((size_t)((kt * 3 + s) * 4 + k8) * M + m) * 16;
}
// Where token `pos` of (slot, kv head) lives in a layer's K (or V) cache, in rows of 64 elements.  Contiguous slots:
// [slot][kvh][max_pos]; paged (ptab != null): the cache is a pool of pages [page][kvh][1 << pshift] and a slot's
// positions map to pages through its row of the page table (smi_llm_cfg.kv_page_tokens / kv_pages).
__device__ __forceinline__ size_t kv_row(const KvMap& km, int slot, int kvh, int n_kv, int max_pos, int pos) {
  if (km.ptab) {
    const int pg = km.ptab[(size_t)slot * km.ppslot + (pos >> km.pshift)];
    return (((size_t)pg * n_kv + kvh) << km.pshift) + (size_t)(pos & ((1 << km.pshift) - 1));
  }
  return ((size_t)slot * n_kv + kvh) * max_pos + pos;
}

// o_proj's reduction index is stored head-interleaved: k tile (half * n_heads + head) holds dims 32 * half .. + 31 of
// `head` (W_o's columns in the arena and the attention output operand alike).  With the k tile -> wave map
// kt mod n_heads of the o_proj kernel, wave w then sums exactly head w's two tiles: the per-head partial that the
// one-row attention kernel can also produce itself (fused o_proj, below) -- same bits either way.
__device__ __forceinline__ int o_ktile(int head, int d, int n_heads) { return (d >> 5) * n_heads + head; }

// ------------------------------------------------------------------------------------------
// GEMM: Y[m][n] = sum_k W[n][k] * X[m][k], X given as exact bf16 triples in global memory.
// MT m-tiles of 16 rows, NTB 16-row weight tiles per block, NW waves split K (k tile kt belongs to
// wave kt mod NW), U tiles per wave in flight per batch.  One accumulator chain per split term,
// summed (lo + mid) + hi; cross-wave reduction in fixed order.  Nothing depends on M, so a row's
// result is bit-identical whatever else is in the batch.  PRO_NORM: the operand is gamma * x and the
// RMSNorm factor rsqrt(mean(x^2) + eps) -- a per-row scalar -- is applied to the accumulator.
// ------------------------------------------------------------------------------------------
// H > 1 (RESID only): the 16 rows of a weight tile are split over H blocks (8 or 4 rows each: only those
// lanes load, the others feed zeros to the MFMA), so a matrix with few n tiles (N = 896: 56) still
// spreads its HBM stream over 112 / 224 CUs.  Every output element keeps its own summation order.
// LEAN: the few-row decode instantiation (M <= 5, operands staged through wave-private LDS, no phase stamps): the paths it
// never takes are compiled out -- these kernels are bound by instruction issue as much as by memory.
// Sum of a row's RMSNorm partials for one lane (index order lane, lane + 64, ...; the caller adds the lanes with
// smi_wave_sum).  Four loads are issued together from clamped addresses and masked by value: written as a plain loop
// the compiler emits load -> wait -> add per trip, i.e. one L2 round trip per 64 partials behind the weight stream.
__device__ __forceinline__ float smi_ss_lane_sum(const float* sp, int npart, int lane) {
  float v = 0.f;
  for (int i0 = lane; i0 < npart; i0 += 256) {   // block-uniform trip count is not required: lanes past npart add zeros
    const int i1 = i0 + 64, i2 = i0 + 128, i3 = i0 + 192, last = npart - 1;
    float a0 = sp[i0 < npart ? i0 : last], a1 = sp[i1 < npart ? i1 : last], a2 = sp[i2 < npart ? i2 : last], a3 = sp[i3 < npart ? i3 : last];
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    v += a0;                         // i0 < npart by the loop condition
    v += i1 < npart ? a1 : 0.f;
    v += i2 < npart ? a2 : 0.f;
    v += i3 < npart ? a3 : 0.f;
  }
  return v;
}

template <int MT, int NTB, int NW, int U, int WB, int PRO, int EPI, int KVF32, int H = 1, int OCC = 1, int LEAN = 0, int NOH = kMaxOHeads>
__global__ __launch_bounds__(NW * 64, OCC) void k_gemm(GemmP p) {
  static_assert(H == 1 || ((H == 2 || H == 4) && NTB == 1 && EPI == EPI_RESID), "row-split tiles: RESID, one tile per block");
  static_assert(PRO != PRO_FUSEDO || (LEAN == 2 && MT == 1 && EPI == EPI_SWIGLU), "PRO_FUSEDO: the one-row gate_up kernel");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= p.work_blocks) {
    pf_run(p.pf, (int)blockIdx.x - p.work_blocks, (int)gridDim.x - p.work_blocks, tid, NW * 64);
    return;
  }
  const int bx = (int)blockIdx.x;
  const int KT = p.KT, M = LEAN == 2 ? 1 : p.M, NT = p.NT;   // LEAN == 2: exactly one row (batch-1 decode), folded at compile time
  const int mbase = LEAN == 2 ? 0 : (int)blockIdx.y * (MT * 16);   // row group (grid.y > 1: a prompt's rows, 32 per block row)
  const int nblk = p.work_blocks / H;   // blocks per row part; part r of tile t is block r * nblk + t (same XCD for all r)
  const int part = H > 1 ? (int)blockIdx.x / nblk : 0;
  const int nt0 = (H > 1 ? (int)blockIdx.x % nblk : (int)blockIdx.x) * NTB;
  const bool wact = H == 1 || ((lane & 15) >> (H == 2 ? 3 : 2)) == part;   // this lane's weight row is in the part
  const bool ract = H == 1 || ((lane >> 4) >> (H == 2 ? 1 : 0)) == part;   // this lane's 4 output columns are
#define SMI_STAMP(i) do { if (!LEAN && p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  SMI_STAMP(0);
  float4* red = (float4*)smem;                                   // [NW][NTB][MT][64]
  float* rarr = (float*)(smem + (size_t)NW * NTB * MT * 1024);  // [32] per-row RMSNorm factor
  float* bestv = rarr + 32;                                      // LM: [NTB][32]
  int* besti = (int*)(bestv + NTB * 32);

  // ---- operands in flight: the (cold, HBM) weight tiles of the first WB batches are requested up
  //      front; the activation triples (small, L2) are fetched per batch
  uint4 w[WB][U][NTB];
  bf16x8 bf[U][3][MT];
  const int k8 = lane >> 4;
  int mrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { const int m = mbase + mt * 16 + (lane & 15); mrow[mt] = m < M ? m : M - 1; }
  const int wl = (H > 1 || EPI == EPI_RESID) ? smi_wlane(lane, p.wperm) : lane;
  auto load_w = [&](uint4 (&dst)[U][NTB], int j0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int j = j0 + u * NW;
      j = j < KT ? j : KT - 1;
#pragma unroll
      for (int nb = 0; nb < NTB; ++nb) {
        int nt = nt0 + nb;
        nt = nt < NT ? nt : NT - 1;
        dst[u][nb] = wact ? smi_ldw(&p.W[((size_t)nt * KT + j) * 64 + wl]) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto load_bf = [&](int j0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int j = j0 + u * NW;
      j = j < KT ? j : KT - 1;
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) bf[u][s][mt] = *(const bf16x8*)(p.XS + xs_off(j, s, k8, mrow[mt], M));
    }
  };
  // QKV: the epilogue's row descriptors are requested FIRST, so the counted wait for them does not include the weight
  // stream and the RoPE factors (addressed by the row's position) can be requested while the weights are in flight
  const int em = lane & 15;
  RowDesc erd[MT];
  if (EPI == EPI_QKV && wave < NTB && nt0 + wave < NT) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mbase + mt * 16 + em;
      if (LEAN != 2) erd[mt] = p.rows[m < M ? m : M - 1];
    }
  }
  // Non-LEAN PRO_NORM kernels (QKV / gate_up at 6+ rows): the partial sums of squares of the rows whose RMSNorm factor this wave
  // computes (rows wave, wave + NW, ...) are requested HERE, before the weight stream, and summed only after the MFMA loop (the
  // factor is an epilogue operand).  As a row-by-row loop in front of the MFMAs (where the compiler kept it: round-4 ISA of the
  // 32-row gate_up) every row was a dependent L2 round trip behind a vmcnt(0) that also covered all weight and operand loads:
  // four round trips per wave between the last operand byte and the first MFMA.
  constexpr int RPW = (MT * 16 + NW - 1) / NW;
  constexpr bool SS_EARLY = !LEAN && PRO == PRO_NORM && RPW <= 4;
  float ssv[SS_EARLY ? RPW : 1][4];
  const bool ss_early = SS_EARLY && p.npart <= 256;
  if constexpr (SS_EARLY) {
    if (ss_early) {
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        int m = mbase + wave + i * NW;
        m = m < M ? m : M - 1;
        const float* sp = p.sspart + (size_t)m * p.npart;
        const int last = p.npart - 1;
        ssv[i][0] = sp[lane < last ? lane : last]; ssv[i][1] = sp[lane + 64 < last ? lane + 64 : last];
        ssv[i][2] = sp[lane + 128 < last ? lane + 128 : last]; ssv[i][3] = sp[lane + 192 < last ? lane + 192 : last];
      }
    }
  }
  // PRO_FUSEDO: the per-head o_proj partials, the residual row and gamma of this wave's k tiles (two tiles per round, lane =
  // one k) are requested BEFORE the weight tiles: loads return in issue order, so behind the (cold) weights they would only
  // arrive after them and the whole operand build would sit between the weights' arrival and the first MFMA
  constexpr int FRND = PRO == PRO_FUSEDO ? 2 : 1;
  float fpv[FRND][NOH], fhv[FRND], fgv[FRND];   // NOH: the head count the instantiation is built for (PRO_FUSEDO: 14 or 4 -- no dead loads)
  int fkk[FRND];
  if constexpr (PRO == PRO_FUSEDO) {
    const int tw = (KT - wave + NW - 1) / NW, K = KT * 32;
#pragma unroll
    for (int r = 0; r < FRND; ++r) {
      const int tl = 2 * r + (lane >> 5);
      fkk[r] = (wave + (tl < tw ? tl : tw - 1) * NW) * 32 + (lane & 31);
#pragma unroll
      for (int hd = 0; hd < NOH; ++hd) fpv[r][hd] = p.part_o[(size_t)(hd < p.n_oheads ? hd : p.n_oheads - 1) * K + fkk[r]];
      fhv[r] = p.hres[fkk[r]]; fgv[r] = p.gam[fkk[r]];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int b = 0; b < WB; ++b)
    if (wave + b * NW * U < KT) load_w(w[b], wave + b * NW * U);
  // PRO_FUSEDO (gate_up at one row, 608 blocks): every block also touches its share -- one wave-instruction, a dword per 64-byte
  // line -- of the NEXT layer's QKV weights, right behind its own weight loads: the 2 MB are then in the L2 of the XCD whose QKV
  // blocks read them (consumer slice s is read by block s; 72 slices, 8 XCDs: block j takes slice j mod 72 -- same XCD) when
  // QKV starts two launches later, instead of a cold HBM round trip in front of a 3-us kernel.  The value is only kept alive
  // until the MFMA loop is over (the loads return in order behind the weights; nothing waits for them).
  uint32_t pf2v = 0;
  if constexpr (PRO == PRO_FUSEDO) {
    if (p.pf2_base && wave == NW - 1) {
      const int sl = bx % p.pf2_nslices, part = bx / p.pf2_nslices, nparts = (p.work_blocks + p.pf2_nslices - 1) / p.pf2_nslices;
      const int lines = p.pf2_slice >> 6, per = (lines + nparts - 1) / nparts;
      int ln = part * per + (lane < per ? lane : per - 1);
      ln = ln < lines ? ln : lines - 1;
      pf2v = *(const uint32_t*)(p.pf2_base + (size_t)sl * p.pf2_slice + (size_t)ln * 64);
    }
  }
  if constexpr (LEAN == 2 && EPI == EPI_QKV) {
    // one row: its descriptor sits at a uniform address, so it comes through the SCALAR cache -- not counted by vmcnt,
    // hence the RoPE load that needs the position no longer waits for the weight tiles issued above (the compiler's
    // counted wait for a vector load of the descriptor was vmcnt(1): descriptor AND weights).  The descriptor is rewritten
    // by k_finalize every step; like every compiler-scalarised load of data an earlier kernel wrote, this relies on the
    // scalar-cache invalidate of the dispatch's acquire fence (with `glc` the load is 1.3 us slower: 630 -> 662 us per step).
    if (wave < NTB && nt0 + wave < NT) {
      unsigned long long sp;
      asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sp) : "s"(p.rows) : "memory");
      erd[0].slot = (int32_t)(uint32_t)sp;
      erd[0].pos = (int32_t)(uint32_t)(sp >> 32);
    }
  }
  // ---- epilogue operands that do not depend on the GEMM: requested BEFORE the operand staging below, so that the
  // staging wait covers them too (behind it, the first MFMA waited one more L2 round trip for the bias / residual row)
  float4 epre[MT], egam = make_float4(0.f, 0.f, 0.f, 0.f);
  float2 erope[MT][2];
  if (wave < NTB && nt0 + wave < NT) {
    const int n = (nt0 + wave) * 16 + 4 * (lane >> 4);
    if (EPI == EPI_RESID) egam = *(const float4*)(p.gamma_next + n);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mbase + mt * 16 + em;
      if (m < M && ract) {
        if (EPI == EPI_QKV) {
          epre[mt] = *(const float4*)(p.bias + n);
          if (n < p.q_dim + p.kv_dim) {
            const int i0 = (n & 63) >> 1;
            erope[mt][0] = p.rope[(size_t)erd[mt].pos * 32 + i0];
            erope[mt][1] = p.rope[(size_t)erd[mt].pos * 32 + i0 + 1];
          }
        }
        if (EPI == EPI_RESID) epre[mt] = *(const float4*)((p.Yin ? p.Yin : p.Y) + (size_t)m * (NT * 16) + n);
      }
    }
  }
  // few rows (MT == 1, ldsb > 0): a k tile's 12*M operand pieces are contiguous in XS, so the wave pulls
  // them with full-width loads into its own LDS slice (no block barrier) instead of 3 narrow loads per tile
  const bool vlds = LEAN || (MT == 1 && p.ldsb > 0);
  // LEAN PRO_NORM kernels: the partial sums of squares of the row this wave owns (row `wave` of the group) are requested
  // together with its first staging round
  float ss0 = 0.f, ss1 = 0.f, ss2 = 0.f, ss3 = 0.f;
  const bool ss_pre = LEAN >= 1 && PRO == PRO_NORM && p.npart <= 256 && wave < MT * 16 && mbase + wave < M;
  unsigned char* bw = smem + (size_t)NW * NTB * MT * 1024 + 32 * 4 + NTB * 32 * 8 + (size_t)wave * p.ldsb;
  float* ssp = (float*)(smem + (size_t)NW * NTB * MT * 1024 + 32 * 4 + NTB * 32 * 8 + (size_t)NW * p.ldsb);   // PRO_FUSEDO: [K / 4] partial sums of squares
  if constexpr (PRO == PRO_FUSEDO) {
    // The operand of this GEMV is gamma * (h + o_proj(attention)), and the o_proj arrives as one partial per head.  Each wave
    // builds the 32-value tiles it owns: lane = one k (two tiles per round), heads added in order (the order k_gemm<RESID> with
    // NW = heads sums its waves in), then exactly the RESID epilogue's arithmetic: residual add, sum of squares per 4 columns,
    // gamma, exact 3-way split -- written as B-operand pieces into the wave's private staging slice.  All loads of a round
    // are issued together; they are L2 hits and land long before the weight tiles requested above.
    const int tw = (KT - wave + NW - 1) / NW;
#pragma unroll
    for (int r = 0; r < FRND; ++r) {
      const int tl = 2 * r + (lane >> 5), k = fkk[r];
      const bool ok = tl < tw;
      float (&pv)[FRND][NOH] = fpv;
      float (&hv)[FRND] = fhv;
      float (&gv)[FRND] = fgv;
      float y = pv[r][0];
#pragma unroll
      for (int hd = 1; hd < NOH; ++hd)
        if (hd < p.n_oheads) y += pv[r][hd];   // wave-uniform
      const float hm = hv[r] + y;
      uint32_t hi, mi, lo;
      split3(gv[r] * hm, hi, mi, lo);
      float sq = hm * hm;
      sq += __shfl_xor(sq, 1, 64);
      sq += __shfl_xor(sq, 2, 64);          // lanes k % 4 == 0: (x^2 + y^2) + (z^2 + w^2), the RESID epilogue's partial
      if (ok) {
        unsigned char* q = bw + (size_t)(tl * 12 + ((lane >> 3) & 3)) * 16 + (lane & 7) * 2;
        *(uint16_t*)q = (uint16_t)hi;
        *(uint16_t*)(q + 64) = (uint16_t)mi;
        *(uint16_t*)(q + 128) = (uint16_t)lo;
        if ((lane & 3) == 0) ssp[k >> 2] = sq;
        if (blockIdx.x == 0) p.h2out[k] = hm;
      }
    }
  } else if (vlds) {
    const int tw = (KT - wave + NW - 1) / NW;          // this wave's k tiles
    const int lts = LEAN == 2 ? 4 : p.lt_shift;
    const int per = 64 >> lts, pm = 12 * M;
    const int r = lane & ((1 << lts) - 1);
    // one row, many k tiles per wave (down_proj: 152 tiles over 16 waves = 3 rounds of 4 tiles): all rounds' loads leave
    // before the first LDS write -- one at a time each round was a full L2 round trip behind the weight stream
    constexpr int RIF = (LEAN == 2 && WB >= 4) ? 3 : 1;   // rounds in flight
    if constexpr (RIF > 1) {
      // (named registers and unconditional loads from clamped addresses: a conditionally written `uint4 v[3]` stays in
      // scratch memory, and `ok ? *p : zero` becomes a flat load through a select between p and a stack slot)
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const int rc = r < pm ? r : pm - 1;
      for (int t0 = 0; t0 < tw; t0 += per * 3) {
        const int ta = t0 + (lane >> lts), tb = ta + per, tc = tb + per;
        u32x4 va = *(const u32x4*)(p.XS + xs_off(wave + (ta < tw ? ta : tw - 1) * NW, 0, 0, 0, M) + rc * 16);
        u32x4 vb = *(const u32x4*)(p.XS + xs_off(wave + (tb < tw ? tb : tw - 1) * NW, 0, 0, 0, M) + rc * 16);
        u32x4 vc = *(const u32x4*)(p.XS + xs_off(wave + (tc < tw ? tc : tw - 1) * NW, 0, 0, 0, M) + rc * 16);
        asm volatile("" : "+v"(va), "+v"(vb), "+v"(vc));   // all three loads are issued here, not sunk into the branches below
        if (ta < tw && r < pm) *(u32x4*)(bw + ((size_t)ta * pm + r) * 16) = va;
        if (tb < tw && r < pm) *(u32x4*)(bw + ((size_t)tb * pm + r) * 16) = vb;
        if (tc < tw && r < pm) *(u32x4*)(bw + ((size_t)tc * pm + r) * 16) = vc;
      }
    } else {
      int t0 = 0;
      if constexpr (LEAN >= 1 && PRO == PRO_NORM) {
        // first staging round and this wave's RMSNorm partials (if it owns a row) behind ONE wait: two round trips -> one
        if (tw > 0) {
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          const int tl = lane >> lts;
          u32x4 v0 = *(const u32x4*)(p.XS + xs_off(wave + (tl < tw ? tl : tw - 1) * NW, 0, 0, 0, M) + (r < pm ? r : pm - 1) * 16);
          if (ss_pre) {
            const float* sp = p.sspart + (size_t)(mbase + wave) * p.npart;
            const int last = p.npart - 1;
            ss0 = sp[lane < last ? lane : last]; ss1 = sp[lane + 64 < last ? lane + 64 : last];
            ss2 = sp[lane + 128 < last ? lane + 128 : last]; ss3 = sp[lane + 192 < last ? lane + 192 : last];
          }
          asm volatile("" : "+v"(v0), "+v"(ss0), "+v"(ss1), "+v"(ss2), "+v"(ss3));
          if (tl < tw && r < pm) *(u32x4*)(bw + ((size_t)tl * pm + r) * 16) = v0;
          t0 = per;
        }
      }
      for (; t0 < tw; t0 += per) {
        const int tl = t0 + (lane >> lts);
        if (tl < tw && r < pm) {
          const uint4 v = *(const uint4*)(p.XS + xs_off(wave + tl * NW, 0, 0, 0, M) + r * 16);
          *(uint4*)(bw + ((size_t)tl * pm + r) * 16) = v;
        }
      }
    }
  } else if (wave < KT) {
    load_bf(wave);
  }

  // ---- RMSNorm factor per row from the producer's partial sums (fixed order: DPP tree over partials)
  if (PRO == PRO_NORM && !ss_early) {
    for (int ml = wave; ml < MT * 16 && mbase + ml < M; ml += NW) {
      float v;
      if (ss_pre && ml == wave) {   // same order as smi_ss_lane_sum
        v = lane < p.npart ? ss0 : 0.f;
        v += lane + 64 < p.npart ? ss1 : 0.f;
        v += lane + 128 < p.npart ? ss2 : 0.f;
        v += lane + 192 < p.npart ? ss3 : 0.f;
      } else {
        v = smi_ss_lane_sum(p.sspart + (size_t)(mbase + ml) * p.npart, p.npart, lane);
      }
      v = smi_wave_sum(v);
      if (lane == 0) rarr[ml] = 1.0f / sqrtf(v / (float)(KT * 32) + p.eps);
    }
  }
  SMI_STAMP(1);

  f32x4 acc[3][NTB][MT];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int a = 0; a < NTB; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) acc[c][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (vlds) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own staging writes have landed (same wave reads them)
  auto compute = [&](const uint4 (&wt)[U][NTB], int j0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (j0 + u * NW < KT) {
        if constexpr (LEAN == 2) {
          // ONE row: the three split terms of the operand are three COLUMNS of one MFMA (lane (k8, col): col 0 hi, 1 mid, 2 lo;
          // the other columns repeat hi and are never read), not three MFMAs with the same operand in every column.  A
          // column's sum does not depend on its neighbours, so column c is bit for bit the old accumulator of split c -- one
          // third of the matrix-pipe work and of the operand reads behind the last weight byte (the kernel's tail).
          const bf16x8 b1 = *(const bf16x8*)(bw + ((size_t)((j0 - wave) / NW + u) * 12 + ((lane & 15) < 3 ? (lane & 15) : 0) * 4 + k8) * 16);
#pragma unroll
          for (int nb = 0; nb < NTB; ++nb)
            acc[0][nb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wt[u][nb]), b1, acc[0][nb][0], 0, 0, 0);
          continue;
        }
        if (vlds) {
          const unsigned char* bp = bw + ((size_t)((j0 - wave) / NW + u) * 12 * M + k8 * M + mrow[0]) * 16;
#pragma unroll
          for (int s = 0; s < 3; ++s) bf[u][s][0] = *(const bf16x8*)(bp + (size_t)s * 4 * M * 16);
        }
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
          const bf16x8 a = __builtin_bit_cast(bf16x8, wt[u][nb]);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            acc[2][nb][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[u][2][mt], acc[2][nb][mt], 0, 0, 0);
            acc[1][nb][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[u][1][mt], acc[1][nb][mt], 0, 0, 0);
            acc[0][nb][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[u][0][mt], acc[0][nb][mt], 0, 0, 0);
          }
        }
      }
    }
  };
#pragma unroll
  for (int b = 0; b < WB; ++b) {
    const int j0 = wave + b * NW * U;
    if (j0 < KT) {
      if (b > 0 && !vlds) load_bf(j0);
      compute(w[b], j0);
    }
  }
  for (int j0 = wave + WB * NW * U; j0 < KT; j0 += NW * U) {   // only for K beyond NW*U*WB tiles
    load_w(w[0], j0);
    if (!vlds) load_bf(j0);
    compute(w[0], j0);
  }
  if constexpr (PRO == PRO_FUSEDO) asm volatile("" ::"v"(pf2v));
  if constexpr (SS_EARLY) {
    if (ss_early) {   // the partials requested at entry: same order as smi_ss_lane_sum + smi_wave_sum
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int ml = wave + i * NW;
        asm volatile("" : "+v"(ssv[i][0]), "+v"(ssv[i][1]), "+v"(ssv[i][2]), "+v"(ssv[i][3]));   // first looked at here, behind the MFMAs
        float v = lane < p.npart ? ssv[i][0] : 0.f;
        v += lane + 64 < p.npart ? ssv[i][1] : 0.f;
        v += lane + 128 < p.npart ? ssv[i][2] : 0.f;
        v += lane + 192 < p.npart ? ssv[i][3] : 0.f;
        v = smi_wave_sum(v);
        if (lane == 0 && ml < MT * 16 && mbase + ml < M) rarr[ml] = 1.0f / sqrtf(v / (float)(KT * 32) + p.eps);
      }
    }
  }
  SMI_STAMP(4);
  // ---- split-K reduction across the block's waves (fixed order => deterministic)
#pragma unroll
  for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 t;
      if constexpr (LEAN == 2) {   // columns 1 / 2 (lanes + 1 / + 2 of the row of 16) hold mid / lo; only column 0 -- the row -- is read later
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = (smi_dpp<0x102>(acc[0][nb][mt][r]) + smi_dpp<0x101>(acc[0][nb][mt][r])) + acc[0][nb][mt][r];
      } else {
        t = (acc[2][nb][mt] + acc[1][nb][mt]) + acc[0][nb][mt];   // (lo + mid) + hi
      }
      red[((wave * NTB + nb) * MT + mt) * 64 + lane] = make_float4(t[0], t[1], t[2], t[3]);
    }
  if (EPI == EPI_LM) {
    for (int i = tid; i < NTB * 32; i += NW * 64) { bestv[i] = -INFINITY; besti[i] = 0x7fffffff; }
  }
  __syncthreads();
  SMI_STAMP(5);

  const int N = NT * 16;
  // SPREAD (gate_up with two m-tiles): the NTB * MT (tile, m-tile) pairs go to NTB * MT different waves instead of NTB waves
  // finishing MT pairs each (five of the eight waves used to idle through the epilogue: two expf, two divides, two exact splits
  // per pair).  Only kernels whose epilogue needs no per-wave prefetched operands (SWIGLU) do this.
  constexpr bool SPREAD = EPI == EPI_SWIGLU && MT == 2 && NTB * MT <= NW;
  for (int nb = SPREAD ? wave / MT : wave; nb < NTB; nb += NW) {
    const int nt = nt0 + nb;
    if (nt >= NT) continue;  // wave-uniform
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (SPREAD && mt != wave % MT) continue;   // wave-uniform
      float4 s = red[((0 * NTB + nb) * MT + mt) * 64 + lane];
#pragma unroll
      for (int wv = 1; wv < NW; ++wv) {
        const float4 t = red[((wv * NTB + nb) * MT + mt) * 64 + lane];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      const int ml = mt * 16 + em, m = mbase + ml;
      const int n = nt * 16 + 4 * (lane >> 4);
      const bool valid = m < M && ract;
      if (PRO == PRO_NORM) {
        const float r = rarr[m < M ? ml : 0];
        s.x *= r; s.y *= r; s.z *= r; s.w *= r;
      }
      if constexpr (PRO == PRO_FUSEDO) {   // the waves left their partial sums of squares in LDS; summed as smi_ss_lane_sum would
        const int npart = KT * 8;
        float v = lane < npart ? ssp[lane] : 0.f;
        v += lane + 64 < npart ? ssp[lane + 64] : 0.f;
        v += lane + 128 < npart ? ssp[lane + 128] : 0.f;
        v += lane + 192 < npart ? ssp[lane + 192] : 0.f;
        v = smi_wave_sum(v);
        const float r = 1.0f / sqrtf(v / (float)(KT * 32) + p.eps);
        s.x *= r; s.y *= r; s.z *= r; s.w *= r;
      }
      if (EPI == EPI_RESID) {
        float ssq = 0.f;
        if (valid) {
          float4 h = epre[mt];   // fetched at kernel entry (only this lane ever writes it)
          h.x += s.x; h.y += s.y; h.z += s.z; h.w += s.w;
          *(float4*)(p.Y + (size_t)m * N + n) = h;
          ssq = (h.x * h.x + h.y * h.y) + (h.z * h.z + h.w * h.w);
          // next consumer's operand: exact triples of gamma_next * h_new (4 of the 8 values of an octet)
          const float t[4] = {egam.x * h.x, egam.y * h.y, egam.z * h.z, egam.w * h.w};
          uint32_t hi[4], mi[4], lo[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split3(t[e], hi[e], mi[e], lo[e]);
          const size_t o = xs_off(n >> 5, 0, (n >> 3) & 3, m, M) + ((n >> 2) & 1) * 8;
          const size_t pl = (size_t)4 * M * 16;
          *(uint2*)(p.XSout + o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
          *(uint2*)(p.XSout + o + pl) = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
          *(uint2*)(p.XSout + o + 2 * pl) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
        }
        // one partial per 4 columns (this tile's 16 columns of row m sit in lanes em, em+16, em+32, em+48)
        if (valid) p.ssout[((size_t)m * NT + nt) * 4 + (lane >> 4)] = ssq;
      } else if (EPI == EPI_SWIGLU) {
        if (valid) {
          // rows are (gate, up, gate, up): silu(g) * u, MQ:46-48
          const float a0 = (s.x / (1.0f + expf(-s.x))) * s.y;
          const float a1 = (s.z / (1.0f + expf(-s.z))) * s.w;
          uint32_t h0, m0, l0, h1, m1, l1;
          split3(a0, h0, m0, l0);
          split3(a1, h1, m1, l1);
          const int k = n >> 1;   // activation index of a0
          const size_t o = xs_off(k >> 5, 0, (k >> 3) & 3, m, M) + ((k >> 1) & 3) * 4;
          const size_t pl = (size_t)4 * M * 16;
          *(uint32_t*)(p.XSout + o) = h0 | (h1 << 16);
          *(uint32_t*)(p.XSout + o + pl) = m0 | (m1 << 16);
          *(uint32_t*)(p.XSout + o + 2 * pl) = l0 | (l1 << 16);
        }
      } else if (EPI == EPI_QKV) {
        if (valid) {
          const float4 b = epre[mt];
          s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
          const RowDesc rd = erd[mt];
          if (n < p.q_dim + p.kv_dim) {
            // rows inside a head are ordered (0,32,1,33,..): (s.x,s.y) and (s.z,s.w) are RoPE pairs
            const float2 c0 = erope[mt][0], c1 = erope[mt][1];   // requested in the prologue
            float4 r;
            r.x = __fadd_rn(__fmul_rn(s.x, c0.x), __fmul_rn(-s.y, c0.y));  // x1*cos + (-x2)*sin
            r.y = __fadd_rn(__fmul_rn(s.y, c0.x), __fmul_rn(s.x, c0.y));   // x2*cos + x1*sin
            r.z = __fadd_rn(__fmul_rn(s.z, c1.x), __fmul_rn(-s.w, c1.y));
            r.w = __fadd_rn(__fmul_rn(s.w, c1.x), __fmul_rn(s.z, c1.y));
            s = r;
          }
          if (n < p.q_dim) {
            *(float4*)(p.Y + (size_t)m * p.q_dim + n) = s;
          } else {
            const bool isk = n < p.q_dim + p.kv_dim;
            const int c = n - p.q_dim - (isk ? 0 : p.kv_dim);
            const size_t off = kv_row(p.km, rd.slot, c >> 6, p.n_kv, p.max_pos, rd.pos) * 64 + (c & 63);
            void* base = isk ? p.kcache : p.vcache;
            if (KVF32) {
              *(float4*)((float*)base + off) = s;
            } else {
              uint2 pk;
              pk.x = smi_f32_to_bf16(s.x) | (smi_f32_to_bf16(s.y) << 16);
              pk.y = smi_f32_to_bf16(s.z) | (smi_f32_to_bf16(s.w) << 16);
              *(uint2*)((uint16_t*)base + off) = pk;
            }
          }
        }
      } else {  // EPI_LM
        if (valid && p.Y) {
          float* y = p.Y + (size_t)m * p.V + n;
          if (n + 0 < p.V) y[0] = s.x;
          if (n + 1 < p.V) y[1] = s.y;
          if (n + 2 < p.V) y[2] = s.z;
          if (n + 3 < p.V) y[3] = s.w;
        }
        // lane-local best over its 4 rows (ascending n, strict > keeps the lowest index on ties)
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.V && sv[r] > bv) { bv = sv[r]; bi = n + r; }
        // combine the 4 lanes that share this m (lane, lane^16, lane^32, lane^48)
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
          float ov = __shfl_xor(bv, o, 64);
          int oi = __shfl_xor(bi, o, 64);
          if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane < 16 && valid) { bestv[nb * 32 + ml] = bv; besti[nb * 32 + ml] = bi; }
      }
    }
  }
  SMI_STAMP(6);
  if (EPI == EPI_LM) {
    __syncthreads();
    if (tid < M) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int nb = 0; nb < NTB; ++nb) {
        float ov = bestv[nb * 32 + tid];
        int oi = besti[nb * 32 + tid];
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      p.pval[(size_t)tid * p.work_blocks + blockIdx.x] = bv;   // [row][block]: finalize reads a row contiguously
      p.pidx[(size_t)tid * p.work_blocks + blockIdx.x] = bi;
    }
  }
}

// ------------------------------------------------------------------------------------------
// One-row RESID GEMV with four row parts per weight tile (down_proj at batch 1), FULL load instructions.
// k_gemm<.., H = 4> gives each of 16 waves one chain (k tiles kt mod 16) and, a part being 4 of a tile's 16 rows, loads with
// 16 of its 64 lanes: 160 quarter-full load instructions per block.  Here a wave owns FOUR chains and advances them with one
// MFMA: A row 4s + r = weight row r of the part in the k tile of chain 4 * wave + s, B column 3s + c = split c (hi, mid, lo)
// of that tile's operand; the four 4 x 3 diagonal blocks of D are the four chains' accumulators (the other blocks mix tiles
// and are never read).  One load instruction then carries 1 KiB (W_down is stored row-part-major: the part's 16 pieces of a
// k tile are 256 contiguous bytes), 40 per block.  A chain's sum, (lo + mid) + hi, and the in-order sum over the 16 chains are
// k_gemm's, so the result is the same bit for bit; measured on the stand-alone probe (tools/gemv_ab.py, csrc/diag/
// diag_gemv.hip): 5.92 -> 3.98 us per launch.  TPC = k tiles per chain the registers hold (KT <= 16 * TPC).
// ------------------------------------------------------------------------------------------
template <int TPC>
__global__ __launch_bounds__(256) void k_down1(GemmP p) {
  __shared__ __attribute__((aligned(16))) float red[16 * 4];   // read and written as float4
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= p.work_blocks) {
    pf_run(p.pf, (int)blockIdx.x - p.work_blocks, (int)gridDim.x - p.work_blocks, tid, 256);
    return;
  }
  const int KT = p.KT, NT = p.NT;
  const int nt = (int)blockIdx.x % NT, part = (int)blockIdx.x / NT;   // part r of tile t is block r * NT + t (same XCD for all r)
  const uint4* wt = p.W + (size_t)nt * KT * 64;
  const int row = lane & 15, k8 = lane >> 4;
  const int sa = row >> 2, r = row & 3;                                   // A: chain slot and weight row of this lane's row
  const int sb = row < 12 ? row / 3 : 0, c = row < 12 ? row % 3 : 0;      // B: chain slot and split term of this lane's column
  const int piece = p.wperm ? part * 16 + k8 * 4 + r : k8 * 16 + part * 4 + r;
  // the epilogue's operands (one lane per block owns the part's 4 output columns) are requested first: they are L2 hits
  // and must not queue behind the cold weight tiles
  const int n = nt * 16 + part * 4;
  float4 epre = make_float4(0.f, 0.f, 0.f, 0.f), egam = epre;
  if (tid == 0) {
    epre = *(const float4*)((p.Yin ? p.Yin : p.Y) + n);
    egam = *(const float4*)(p.gamma_next + n);
  }
  uint4 w[TPC];
  bf16x8 b[TPC];
#pragma unroll
  for (int u = 0; u < TPC; ++u) {
    int t = 4 * wave + sa + 16 * u;
    const bool ok = t < KT;
    t = ok ? t : KT - 1;
    w[u] = smi_ldw(wt + (size_t)t * 64 + piece);
    if (!ok) w[u] = make_uint4(0u, 0u, 0u, 0u);   // a chain's missing last tile adds zeros
  }
#pragma unroll
  for (int u = 0; u < TPC; ++u) {
    int t = 4 * wave + sb + 16 * u;
    t = t < KT ? t : KT - 1;
    b[u] = *(const bf16x8*)(p.XS + xs_off(t, c, k8, 0, 1));
  }
  __builtin_amdgcn_sched_barrier(0);   // every load of the wave is requested before the first MFMA's wait
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < TPC; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[u]), b[u], acc, 0, 0, 0);
  // D: lane (col = lane & 15, g = lane >> 4) holds rows 4g .. 4g + 3 of column col; chain slot s is row group s, columns 3s .. 3s + 2
  float t4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) t4[e] = (smi_dpp<0x102>(acc[e]) + smi_dpp<0x101>(acc[e])) + acc[e];   // (lo + mid) + hi
  if (row == 3 * k8) *(float4*)(red + (4 * wave + k8) * 4) = make_float4(t4[0], t4[1], t4[2], t4[3]);
  __syncthreads();
  if (tid == 0) {
    float4 sres = *(const float4*)red;
#pragma unroll
    for (int ch = 1; ch < 16; ++ch) {   // the 16 chains in order, as k_gemm sums its 16 waves
      const float4 q = *(const float4*)(red + ch * 4);
      sres.x += q.x; sres.y += q.y; sres.z += q.z; sres.w += q.w;
    }
    // RESID epilogue of k_gemm for row 0, columns n .. n + 3
    float4 h = epre;
    h.x += sres.x; h.y += sres.y; h.z += sres.z; h.w += sres.w;
    *(float4*)(p.Y + n) = h;
    const float ssq = (h.x * h.x + h.y * h.y) + (h.z * h.z + h.w * h.w);
    const float tv[4] = {egam.x * h.x, egam.y * h.y, egam.z * h.z, egam.w * h.w};
    uint32_t hi[4], mi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3(tv[e], hi[e], mi[e], lo[e]);
    const size_t o = xs_off(n >> 5, 0, (n >> 3) & 3, 0, 1) + ((n >> 2) & 1) * 8;
    const size_t pl = (size_t)4 * 16;
    *(uint2*)(p.XSout + o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
    *(uint2*)(p.XSout + o + pl) = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
    *(uint2*)(p.XSout + o + 2 * pl) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
    p.ssout[(size_t)nt * 4 + part] = ssq;
  }
}

// The same for 2 .. 8 rows (k_downS<TPC, MG>, rows in MG groups of four): a wave's load instruction still carries four chains'
// parts; B column 4s + j = row 4g + j of the tile of chain slot s, one MFMA per split term and row group (3 * MG per step
// instead of 12 with quarter-full loads), accumulators and summation order as k_gemm's -- (lo + mid) + hi per chain, chains in
// order -- so a row's result stays bit-identical whatever the batch.  All operands of a wave are in flight at once (one wave
// per SIMD: up to 512 registers).
template <int TPC, int MG>
__global__ __launch_bounds__(256) void k_downS(GemmP p) {
  __shared__ __attribute__((aligned(16))) float red[16 * 4 * MG * 4];   // [chain][row][4 columns]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KT = p.KT, NT = p.NT, M = p.M;
  const int nt = (int)blockIdx.x % NT, part = (int)blockIdx.x / NT;
  const uint4* wt = p.W + (size_t)nt * KT * 64;
  const int row = lane & 15, k8 = lane >> 4;
  const int sa = row >> 2, r = row & 3;      // A: chain slot and weight row of this lane's row; B: chain slot and row inside its group of four
  const int piece = p.wperm ? part * 16 + k8 * 4 + r : k8 * 16 + part * 4 + r;
  const int n = nt * 16 + part * 4;
  float4 epre = make_float4(0.f, 0.f, 0.f, 0.f), egam = epre;
  if (tid < M) {
    epre = *(const float4*)((p.Yin ? p.Yin : p.Y) + (size_t)tid * (NT * 16) + n);
    egam = *(const float4*)(p.gamma_next + n);
  }
  uint4 w[TPC];
  bf16x8 b[TPC][3][MG];
#pragma unroll
  for (int u = 0; u < TPC; ++u) {
    int t = 4 * wave + sa + 16 * u;
    const bool ok = t < KT;
    t = ok ? t : KT - 1;
    w[u] = smi_ldw(wt + (size_t)t * 64 + piece);
    if (!ok) w[u] = make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int u = 0; u < TPC; ++u) {
    int t = 4 * wave + sa + 16 * u;
    t = t < KT ? t : KT - 1;
#pragma unroll
    for (int g = 0; g < MG; ++g) {
      int m = 4 * g + r;
      m = m < M ? m : M - 1;
#pragma unroll
      for (int c = 0; c < 3; ++c) b[u][c][g] = *(const bf16x8*)(p.XS + xs_off(t, c, k8, m, M));
    }
  }
  __builtin_amdgcn_sched_barrier(0);   // every load of the wave is requested before the first MFMA's wait
  f32x4 acc[3][MG];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int g = 0; g < MG; ++g) acc[c][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < TPC; ++u) {
    const bf16x8 a = __builtin_bit_cast(bf16x8, w[u]);
#pragma unroll
    for (int g = 0; g < MG; ++g) {
      acc[2][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[u][2][g], acc[2][g], 0, 0, 0);
      acc[1][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[u][1][g], acc[1][g], 0, 0, 0);
      acc[0][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[u][0][g], acc[0][g], 0, 0, 0);
    }
  }
  // D: lane (col, rg = lane >> 4) holds weight rows 0..3 of chain slot rg for column col; the slot's own columns are 4 rg .. 4 rg + 3
  if (sa == k8) {
#pragma unroll
    for (int g = 0; g < MG; ++g) {
      const f32x4 t = (acc[2][g] + acc[1][g]) + acc[0][g];   // (lo + mid) + hi
      *(float4*)(red + ((4 * wave + k8) * 4 * MG + 4 * g + r) * 4) = make_float4(t[0], t[1], t[2], t[3]);
    }
  }
  __syncthreads();
  if (tid < M) {
    const int m = tid;
    float4 sres = *(const float4*)(red + m * 4);
#pragma unroll
    for (int ch = 1; ch < 16; ++ch) {   // the 16 chains in order
      const float4 q = *(const float4*)(red + (ch * 4 * MG + m) * 4);
      sres.x += q.x; sres.y += q.y; sres.z += q.z; sres.w += q.w;
    }
    float4 h = epre;   // RESID epilogue of k_gemm for row m, columns n .. n + 3
    h.x += sres.x; h.y += sres.y; h.z += sres.z; h.w += sres.w;
    *(float4*)(p.Y + (size_t)m * (NT * 16) + n) = h;
    const float ssq = (h.x * h.x + h.y * h.y) + (h.z * h.z + h.w * h.w);
    const float tv[4] = {egam.x * h.x, egam.y * h.y, egam.z * h.z, egam.w * h.w};
    uint32_t hi[4], mi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3(tv[e], hi[e], mi[e], lo[e]);
    const size_t o = xs_off(n >> 5, 0, (n >> 3) & 3, m, M) + ((n >> 2) & 1) * 8;
    const size_t pl = (size_t)4 * M * 16;
    *(uint2*)(p.XSout + o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
    *(uint2*)(p.XSout + o + pl) = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
    *(uint2*)(p.XSout + o + 2 * pl) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
    p.ssout[((size_t)m * NT + nt) * 4 + part] = ssq;
  }
}

// ------------------------------------------------------------------------------------------
// Exact-weights verification mode (smi_llm_cfg.weights_exact = 1; round 4).  The reference loads the checkpoint as saved
// (cli/SparkTTS.py:48-51) and the published LLM/model.safetensors is fp32: the bf16 arena rounds it (logits move ~1e-2, a
// tenth of the greedy decisions of a synthetic model change).  In this mode the arena holds the matrices as fp32 [N][K]
// (same row / column orders as the bf16 tiles) and every GEMM of the path runs here: one wave per (16-row weight tile, 16-row
// m-tile), v_mfma_f32_16x16x4_f32 -- an exact fp32 fused multiply-add chain over k, like the CPU's -- on operands rebuilt
// exactly from the triples (x = hi + mid + lo), with k_gemm's own prologue scalars and epilogues (RMSNorm factor, bias, RoPE,
// KV append, residual, SwiGLU, arg-max partials, the next operand's triples).  Throughput is not a goal (a weight byte is
// read once per 16 rows through 64-byte row segments: ~1.5 ms per step at one row); north_star's acceptance sentence --
// waveform within 1e-3 of the PyTorch CPU path for fixed greedy seeds -- on an fp32-saved checkpoint is.
// ------------------------------------------------------------------------------------------
template <int PRO, int EPI, int KVF32>
__global__ __launch_bounds__(256) void k_gemm_x(GemmP p, const float* W32) {
  __shared__ float bestv[4][16];
  __shared__ int besti[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KT = p.KT, K = KT * 32, M = p.M, NT = p.NT, N = NT * 16;
  const int nt = (int)blockIdx.x * 4 + wave, mbase = (int)blockIdx.y * 16;
  const int em = lane & 15, kq = lane >> 4;
  const int m = mbase + em, mc = m < M ? m : M - 1;
  const bool live = nt < NT;                       // wave-uniform
  const int ntc = live ? nt : NT - 1;
  // ---- y[n][m] = sum_k W[n][k] x[m][k]: 16 k per step, lane (kq, i) holds W[16 nt + i][k0 + 4 kq .. + 3] and x[m = i][same k]
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* wrow = W32 + (size_t)(ntc * 16 + em) * K + 4 * kq;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const float4 wv = *(const float4*)(wrow + k0);
    const int k = k0 + 4 * kq;                     // this lane's four k: tile k >> 5, octet (k >> 3) & 3, elements k & 7 .. + 3
    const unsigned char* xp = p.XS + xs_off(k >> 5, 0, (k >> 3) & 3, mc, M) + (k & 7) * 2;
    const size_t pl = (size_t)4 * M * 16;
    const uint2 h2 = *(const uint2*)xp, m2 = *(const uint2*)(xp + pl), l2 = *(const uint2*)(xp + 2 * pl);
    const uint32_t hb[4] = {h2.x << 16, h2.x & 0xffff0000u, h2.y << 16, h2.y & 0xffff0000u};
    const uint32_t mb[4] = {m2.x << 16, m2.x & 0xffff0000u, m2.y << 16, m2.y & 0xffff0000u};
    const uint32_t lb[4] = {l2.x << 16, l2.x & 0xffff0000u, l2.y << 16, l2.y & 0xffff0000u};
    const float wa[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x = (__uint_as_float(hb[c]) + __uint_as_float(mb[c])) + __uint_as_float(lb[c]);   // exact: the split terms do not overlap
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c], x, acc, 0, 0, 0);
    }
  }
  // ---- k_gemm's epilogue for (tile nt, rows mbase ..): lane (g = lane >> 4, m) holds columns n .. n + 3 of row m
  float4 s = make_float4(acc[0], acc[1], acc[2], acc[3]);
  const int n = ntc * 16 + 4 * (lane >> 4);
  const bool valid = live && m < M;
  if (PRO == PRO_NORM) {
    // every lane needs ITS row's factor: rows differ across the 16 lanes of a group, so each row is summed by the whole wave in turn
    float r = 0.f;
    for (int j = 0; j < 16; ++j) {
      const int mj = mbase + j < M ? mbase + j : M - 1;
      float vj = smi_ss_lane_sum(p.sspart + (size_t)mj * p.npart, p.npart, lane);
      vj = smi_wave_sum(vj);
      const float rj = 1.0f / sqrtf(vj / (float)K + p.eps);
      r = em == j ? rj : r;
    }
    s.x *= r; s.y *= r; s.z *= r; s.w *= r;
  }
  if (EPI == EPI_RESID) {
    if (valid) {
      const float4 egam = *(const float4*)(p.gamma_next + n);
      float4 h = *(const float4*)((p.Yin ? p.Yin : p.Y) + (size_t)m * N + n);
      h.x += s.x; h.y += s.y; h.z += s.z; h.w += s.w;
      *(float4*)(p.Y + (size_t)m * N + n) = h;
      const float ssq = (h.x * h.x + h.y * h.y) + (h.z * h.z + h.w * h.w);
      const float t[4] = {egam.x * h.x, egam.y * h.y, egam.z * h.z, egam.w * h.w};
      uint32_t hi[4], mi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split3(t[e], hi[e], mi[e], lo[e]);
      const size_t o = xs_off(n >> 5, 0, (n >> 3) & 3, m, M) + ((n >> 2) & 1) * 8;
      const size_t pl = (size_t)4 * M * 16;
      *(uint2*)(p.XSout + o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
      *(uint2*)(p.XSout + o + pl) = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
      *(uint2*)(p.XSout + o + 2 * pl) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
      p.ssout[((size_t)m * NT + nt) * 4 + (lane >> 4)] = ssq;
    }
  } else if (EPI == EPI_SWIGLU) {
    if (valid) {
      const float a0 = (s.x / (1.0f + expf(-s.x))) * s.y;
      const float a1 = (s.z / (1.0f + expf(-s.z))) * s.w;
      uint32_t h0, m0, l0, h1, m1, l1;
      split3(a0, h0, m0, l0);
      split3(a1, h1, m1, l1);
      const int k = n >> 1;
      const size_t o = xs_off(k >> 5, 0, (k >> 3) & 3, m, M) + ((k >> 1) & 3) * 4;
      const size_t pl = (size_t)4 * M * 16;
      *(uint32_t*)(p.XSout + o) = h0 | (h1 << 16);
      *(uint32_t*)(p.XSout + o + pl) = m0 | (m1 << 16);
      *(uint32_t*)(p.XSout + o + 2 * pl) = l0 | (l1 << 16);
    }
  } else if (EPI == EPI_QKV) {
    if (valid) {
      const float4 b = *(const float4*)(p.bias + n);
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      const RowDesc rd = p.rows[m];
      if (n < p.q_dim + p.kv_dim) {
        const int i0 = (n & 63) >> 1;
        const float2 c0 = p.rope[(size_t)rd.pos * 32 + i0], c1 = p.rope[(size_t)rd.pos * 32 + i0 + 1];
        float4 r;
        r.x = __fadd_rn(__fmul_rn(s.x, c0.x), __fmul_rn(-s.y, c0.y));
        r.y = __fadd_rn(__fmul_rn(s.y, c0.x), __fmul_rn(s.x, c0.y));
        r.z = __fadd_rn(__fmul_rn(s.z, c1.x), __fmul_rn(-s.w, c1.y));
        r.w = __fadd_rn(__fmul_rn(s.w, c1.x), __fmul_rn(s.z, c1.y));
        s = r;
      }
      if (n < p.q_dim) {
        *(float4*)(p.Y + (size_t)m * p.q_dim + n) = s;
      } else {
        const bool isk = n < p.q_dim + p.kv_dim;
        const int c = n - p.q_dim - (isk ? 0 : p.kv_dim);
        const size_t off = kv_row(p.km, rd.slot, c >> 6, p.n_kv, p.max_pos, rd.pos) * 64 + (c & 63);
        void* base = isk ? p.kcache : p.vcache;
        if (KVF32) {
          *(float4*)((float*)base + off) = s;
        } else {
          uint2 pk;
          pk.x = smi_f32_to_bf16(s.x) | (smi_f32_to_bf16(s.y) << 16);
          pk.y = smi_f32_to_bf16(s.z) | (smi_f32_to_bf16(s.w) << 16);
          *(uint2*)((uint16_t*)base + off) = pk;
        }
      }
    }
  } else {  // EPI_LM
    if (valid && p.Y) {
      float* y = p.Y + (size_t)m * p.V + n;
      if (n + 0 < p.V) y[0] = s.x;
      if (n + 1 < p.V) y[1] = s.y;
      if (n + 2 < p.V) y[2] = s.z;
      if (n + 3 < p.V) y[3] = s.w;
    }
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (live && n + r < p.V && sv[r] > bv) { bv = sv[r]; bi = n + r; }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane < 16) { bestv[wave][lane] = bv; besti[wave][lane] = bi; }
    __syncthreads();
    if (tid < 16 && mbase + tid < M) {
      bv = bestv[0][tid]; bi = besti[0][tid];
#pragma unroll
      for (int w2 = 1; w2 < 4; ++w2) {
        const float ov = bestv[w2][tid];
        const int oi = besti[w2][tid];
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      p.pval[(size_t)(mbase + tid) * gridDim.x + blockIdx.x] = bv;
      p.pidx[(size_t)(mbase + tid) * gridDim.x + blockIdx.x] = bi;
    }
  }
}

// the embedding row of `token` from the fp32 lm_head of the exact-weights arena ([vocab padded][K] row-major): h, the first
// norm's operand triples and the row's sum of squares, as embed_row leaves them
__device__ __forceinline__ void embed_row_x(const float* W32, int KT, int token, int m, int M, const float* gamma,
                                            float* h, unsigned char* xs, float* sspart, int npart, int lane) {
  const int K = KT * 32;
  float ss = 0.f;
  for (int pc = lane; pc < KT * 4; pc += 64) {
    const float4 a = *(const float4*)(W32 + (size_t)token * K + pc * 8), b = *(const float4*)(W32 + (size_t)token * K + pc * 8 + 4);
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float* dst = h + (size_t)m * K + pc * 8;
    *(float4*)dst = a;
    *(float4*)(dst + 4) = b;
    const float4 g0 = *(const float4*)(gamma + pc * 8), g1 = *(const float4*)(gamma + pc * 8 + 4);
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    uint32_t hi[8], mi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ss += x[e] * x[e]; split3(g[e] * x[e], hi[e], mi[e], lo[e]); }
    const size_t o = xs_off(pc >> 2, 0, pc & 3, m, M);
    const size_t pl = (size_t)4 * M * 16;
    *(uint4*)(xs + o) = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
    *(uint4*)(xs + o + pl) = make_uint4(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16), mi[4] | (mi[5] << 16), mi[6] | (mi[7] << 16));
    *(uint4*)(xs + o + 2 * pl) = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
  }
  ss = smi_wave_sum(ss);
  for (int i = lane; i < npart; i += 64) sspart[(size_t)m * npart + i] = i == 0 ? ss : 0.f;
}

// ------------------------------------------------------------------------------------------
// down_proj at 9 .. 64 rows: the 16 chains go to 16 DIFFERENT blocks (round 4).
// k_gemm<.., H> gives a block all 16 chains of its rows of a weight tile, so every block pulls the WHOLE operand of its 16 rows
// (K = 4864: 467 KB of triples per block at 32 rows, 105 MB of L2 -> CU traffic per launch: its loads land 5.6 us after the
// first wave starts, profiles/r03_stamps_batch.txt) and the two 16-row block rows both stream the weights from HBM (16.6 MB for
// 8.7 MB, r03_pmc_traffic_g_b32.json).  Here block (column group cg, chain c) takes chain c's k tiles (kt = c, c + 16, ...) of
// four weight tiles for ALL rows: 40 KB of weights (every weight byte leaves HBM once per launch) and 1/16 of the operand
// (61 KB at 32 rows), staged once per block in LDS.  A wave owns one weight tile -- no cross-wave reduction -- and leaves the
// chain's sum, (lo + mid) + hi exactly as k_gemm forms it, in `part` [chain][n tile][m tile][lane] (the accumulator's own
// layout: 1-KiB stores).  k_resid_comb then adds the 16 chains IN ORDER and runs the RESID epilogue: the same sums in the same
// order as k_gemm<RESID, NW = 16> / k_downS / k_down1, so a row keeps its bits whatever the batch.  Blocks of one chain sit on
// one XCD (block id = cg * 16 + c): they share the operand slice through one L2.
// ------------------------------------------------------------------------------------------
struct DownCP {
  const uint4* W; int NT, KT, M, wperm;
  const unsigned char* XS;          // act triples [KT][3][4][M][16 B]
  float4* part;                     // [16][NT][MTN][64]
  // combine
  float* Y; const float* Yin; const float* gamma_next; unsigned char* XSout; float* ssout;
};

template <int TPC, int MTN>
__global__ __launch_bounds__(256) void k_downC(DownCP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [tiles of the chain][12 * M] 16-byte operand pieces
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KT = p.KT, M = p.M, NT = p.NT;
  const int c = (int)blockIdx.x & 15, cg = (int)blockIdx.x >> 4;
  const int ntile = c < KT ? (KT - c + 15) >> 4 : 0;    // k tiles of this chain
  if (ntile == 0) {   // block-uniform: fewer than 16 k tiles -- this chain adds zeros, as an idle wave of k_gemm does
    const int nt0 = cg * 4 + (tid >> 6);
    if (nt0 < NT)
      for (int mt = 0; mt < MTN; ++mt) p.part[(((size_t)c * NT + nt0) * MTN + mt) * 64 + (tid & 63)] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const int pm = 12 * M, npieces = ntile * pm;
  const float pm_inv = 1.0f / (float)pm;
  // ---- operand slice: every piece requested before the weights (L2 hits; loads return in issue order)
  constexpr int PPT = (TPC * 12 * MTN * 16 + 255) / 256; // pieces per thread at most
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 pc[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    int q = tid + 256 * i;
    q = q < npieces ? q : npieces - 1;
    const int u = (int)(((float)q + 0.5f) * pm_inv), r = q - u * pm;   // q / pm, exact: q < 2^13, (q + 0.5) / pm is never within 6e-4 of an integer
    pc[i] = *(const u32x4*)(p.XS + ((size_t)(c + 16 * u) * pm + r) * 16);
  }
  // ---- this wave's weight tile, chain c's k tiles: full 1-KiB loads, all in flight
  const int nt = cg * 4 + wave;
  const int ntc = nt < NT ? nt : NT - 1;
  const int wl = smi_wlane(lane, p.wperm);
  uint4 w[TPC];
#pragma unroll
  for (int u = 0; u < TPC; ++u) {
    const int t = c + 16 * u;
    w[u] = smi_ldw(p.W + ((size_t)ntc * KT + (t < KT ? t : KT - 1)) * 64 + wl);
  }
#pragma unroll
  for (int i = 0; i < PPT; ++i) asm volatile("" : "+v"(pc[i]));   // the loads above stay above (not sunk into the guarded stores)
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int q = tid + 256 * i;
    if (q < npieces) *(u32x4*)(smem + (size_t)q * 16) = pc[i];
  }
  __syncthreads();
  f32x4 acc[3][MTN];
#pragma unroll
  for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt) acc[s2][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int k8 = lane >> 4, em = lane & 15;
#pragma unroll
  for (int u = 0; u < TPC; ++u) {
    if (u < ntile) {   // block-uniform
      const bf16x8 a = __builtin_bit_cast(bf16x8, w[u]);
#pragma unroll
      for (int mt = 0; mt < MTN; ++mt) {
        int m = mt * 16 + em;
        m = m < M ? m : M - 1;
        const unsigned char* bp = smem + ((size_t)u * pm + (size_t)k8 * M + m) * 16;
        const bf16x8 b0 = *(const bf16x8*)bp, b1 = *(const bf16x8*)(bp + (size_t)4 * M * 16), b2 = *(const bf16x8*)(bp + (size_t)8 * M * 16);
        acc[2][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b2, acc[2][mt], 0, 0, 0);
        acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc[1][mt], 0, 0, 0);
        acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[0][mt], 0, 0, 0);
      }
    }
  }
  if (nt < NT) {
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt) {
      const f32x4 t = (acc[2][mt] + acc[1][mt]) + acc[0][mt];   // (lo + mid) + hi
      p.part[(((size_t)c * NT + nt) * MTN + mt) * 64 + lane] = make_float4(t[0], t[1], t[2], t[3]);
    }
  }
}

// One wave per (n tile, m tile): the S partial sums in order, then the RESID epilogue (residual add, h, the next norm's operand
// triples and partial sums of squares).  Two producers: k_downC (S = 16 chains; the running sum starts from chain 0's term, as
// k_gemm's cross-wave reduction does; one partial sum of squares per 4 columns) and k_pgemm's split-K form (S segments; ZI: the
// running sum starts from zero, as its in-block fold does; SSTILE: one partial per n tile, combined across the lanes exactly
// as k_pgemm's epilogue combines them) -- in both cases the same sums in the same order as the single-launch form.
template <int ZI, int SSTILE>
__global__ __launch_bounds__(64) void k_resid_comb(DownCP p, int S, int MTN) {
  const int lane = threadIdx.x, unit = (int)blockIdx.x;
  const int nt = unit / MTN, mt = unit - nt * MTN;
  const int NT = p.NT, M = p.M, N = NT * 16;
  const int m = mt * 16 + (lane & 15), n = nt * 16 + 4 * (lane >> 4);
  const bool valid = m < M;
  const int mc = valid ? m : M - 1;
  const float4 egam = *(const float4*)(p.gamma_next + n);
  const float4 epre = *(const float4*)((p.Yin ? p.Yin : p.Y) + (size_t)mc * N + n);
  float4 q[16];
#pragma unroll
  for (int ch = 0; ch < 16; ++ch) q[ch] = p.part[(((size_t)(ch < S ? ch : S - 1) * NT + nt) * MTN + mt) * 64 + lane];
  float4 s = ZI ? make_float4(0.f, 0.f, 0.f, 0.f) : q[0];
#pragma unroll
  for (int ch = ZI ? 0 : 1; ch < 16; ++ch)
    if (ch < S) { s.x += q[ch].x; s.y += q[ch].y; s.z += q[ch].z; s.w += q[ch].w; }   // wave-uniform
  float ssq = 0.f;
  if (valid) {
    float4 h = epre;
    h.x += s.x; h.y += s.y; h.z += s.z; h.w += s.w;
    *(float4*)(p.Y + (size_t)m * N + n) = h;
    ssq = (h.x * h.x + h.y * h.y) + (h.z * h.z + h.w * h.w);
    const float t[4] = {egam.x * h.x, egam.y * h.y, egam.z * h.z, egam.w * h.w};
    uint32_t hi[4], mi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3(t[e], hi[e], mi[e], lo[e]);
    const size_t o = xs_off(n >> 5, 0, (n >> 3) & 3, m, M) + ((n >> 2) & 1) * 8;
    const size_t pl = (size_t)4 * M * 16;
    *(uint2*)(p.XSout + o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
    *(uint2*)(p.XSout + o + pl) = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
    *(uint2*)(p.XSout + o + 2 * pl) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
  }
  if (SSTILE) {   // k_pgemm's epilogue: the tile's four column groups of a row added across the lanes, one partial per (row, n tile)
    ssq += __shfl_xor(ssq, 16, 64);
    ssq += __shfl_xor(ssq, 32, 64);
    if (lane < 16 && valid) p.ssout[(size_t)m * NT + nt] = ssq;
  } else if (valid) {
    p.ssout[((size_t)m * NT + nt) * 4 + (lane >> 4)] = ssq;
  }
}

// ------------------------------------------------------------------------------------------
// Prefill GEMM: many rows (a whole batch of prompts) against one weight matrix, no split-K.
// Block = 4 waves = 8 weight tiles (128 output columns) x 128 rows; each wave owns 2 weight tiles
// (A operands straight from HBM, used for all 8 m-tiles) and the block shares the rows' operand
// triples through a double-buffered LDS image (24 KB per k tile), so a weight tile is read once per
// 128 rows and an operand piece once per 128 output columns.  One accumulator per (tile, m-tile),
// terms in the order lo, mid, hi per k tile: a row's result does not depend on M or on the batch.
// ------------------------------------------------------------------------------------------
// WR = 2: the waves form a 2 x 2 grid (row half x column half) -- a wave reads only its 64 rows' pieces from LDS (half the LDS
// bytes per MFMA of WR = 1, where every wave reads all 128 rows) and each weight tile is requested by the two waves of a column half.
// One LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to LDS bytes [lds_dst + 16 * lane) (lds_dst
// wave-uniform).  Inline asm, M0 written in the statement that reads it: hipcc waits vmcnt(0) in front of every ds_read while an
// LDS-DMA it knows of (the builtin) is pending; these it does not see, the counted waits are written by hand.
__device__ __forceinline__ void smi_glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void smi_keep4(const uint4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }   // (ablation builds: keeps a value live)

// RING >= 3: both operands of a k tile arrive by LDS-DMA (global_load_lds_dwordx4, no staging registers) in a ring of RING slots,
// RING - 1 tiles requested ahead of the one being multiplied; a counted vmcnt + one raw barrier per k tile (the loads of the
// tiles ahead stay in flight across it).  The register-staged form (RING = 0) waits a whole L2 / HBM round trip per k tile.
// XMAP (few column blocks per row block: QKV, o_proj, down): the launch is one-dimensional and a block's (column, row)
// block comes from its position in its XCD's share of the grid (blocks go to XCD id mod 8), so that the column blocks of a row
// block run on one XCD side by side: an operand piece then comes from the Infinity Cache once and from that XCD's L2 for the
// others (as launched, x fastest, the 6-7 column blocks of a row block sat on 6-7 different XCDs and every operand byte
// was served at the Infinity-Cache rate: down_proj 7.3 TB/s of L2 -> LDS traffic; a workgroup reads ~33 GB/s from there, ~70 from L2).
// TWO (RING = 2): two blocks per CU -- 80 KiB of LDS and at most 256 registers each -- instead of one with a deeper ring: one
// block's prologue, epilogue and barrier stalls are covered by the other's MFMAs (gate_up: 2432 blocks of 128 x 256 in 9.5 rounds,
// 10 us of un-overlapped prologue + epilogue per round at one block per CU).  One fragment set, the ring's other slot one k
// tile ahead, and the RMSNorm factors are computed after the k loop into the ring's memory (no LDS left for them beside it).
// MTB (ring form only): m-tiles per block, 8 (128 rows) or 4 (64 rows: the few-row shapes -- twice the blocks for a 127-row prompt).
template <int PRO, int EPI, int KVF32, int NTW = 2, int WR = 1, int RING = 0, int XMAP = 0, int TWO = 0, int MTB = 8>
__global__ __launch_bounds__(256, TWO ? 2 : 1) void k_pgemm(GemmP p) {
  static_assert(MTB == 8 || (MTB == 4 && RING >= 3 && !TWO), "64-row tiles: ring form");
  constexpr int ROWS = MTB * 16, PIECES = 3 * 4 * ROWS;   // 1536 (768) 16-byte pieces per k tile; NTW weight tiles per wave
  constexpr int MTW = MTB / WR, NWC = 4 / WR;                      // m-tiles per wave, waves across the block's columns
  constexpr int CT = NWC * NTW;                                    // weight tiles per block
  constexpr int SLOT = PIECES + CT * 64;                           // RING: 16-byte pieces per ring slot (operand image, then weight tiles)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* Bs = (uint4*)smem;                              // [2][PIECES], or [RING][SLOT]
  float* rarr = TWO ? (float*)smem : (float*)(smem + (size_t)(RING ? RING * SLOT : 2 * PIECES) * 16);  // [ROWS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KT = p.KT, M = p.M, NT = p.NT;
  int bx = blockIdx.x, by = blockIdx.y;
  if constexpr (XMAP) {   // bijective for any grid size: XCD c owns a contiguous range of (row block, column block) pairs, column fastest
    const int gx = (NT + CT - 1) / CT, nb = gridDim.x, q = nb >> 3, r = nb & 7, c = bx & 7;
    const int v = (c < r ? c * (q + 1) : r * (q + 1) + (c - r) * q) + (bx >> 3);
    by = v / gx;
    bx = v - by * gx;
  }
  const int m0 = by * ROWS;
  const int k8 = lane >> 4, em = lane & 15;
  const int wr = wave % WR, wc = wave / WR, mt0 = wr * MTW;
  int nts[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) { const int nt = (bx * NWC + wc) * NTW + i; nts[i] = nt < NT ? nt : NT - 1; }

  f32x4 acc[NTW][MTW];
#pragma unroll
  for (int a = 0; a < NTW; ++a)
#pragma unroll
    for (int b = 0; b < MTW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // K segments (RESID, ring form): in-block -- `tot` takes each finished segment's sum, in order; split-K -- this block sums
  // segment blockIdx.y only and leaves it in p.slab
  constexpr bool SEG = EPI == EPI_RESID && RING >= 3 && !TWO && XMAP;
  const int kseg = SEG ? p.kseg : 0;
  const bool part = SEG && p.slab != nullptr;
  const int k_lo = part ? (int)blockIdx.y * kseg : 0;
  const int k_hi = part ? (k_lo + kseg < KT ? k_lo + kseg : KT) : KT;
  f32x4 tot[SEG ? NTW : 1][SEG ? MTW : 1];
  if constexpr (SEG) {
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
      for (int b = 0; b < MTW; ++b) tot[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  auto fold = [&]() {   // a segment is complete: total += segment sum (0 + s0, then + s1, ...: the order k_resid_comb adds the slab in)
    if constexpr (SEG) {
#pragma unroll
      for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < MTW; ++b) { tot[a][b] = tot[a][b] + acc[a][b]; acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
  };
  if constexpr (RING != 0) {
    static_assert((TWO ? RING == 2 : RING >= 3) && CT % 4 == 0, "ring form: a wave-instruction moves one weight tile, four per pass of the block");
    constexpr int NI = PIECES / 256, PPI = 256 / ROWS;   // operand issues per thread and k tile; (split, octet) planes one issue covers
    constexpr int GA = CT / 4, G = NI + GA;              // LDS-DMA instructions per thread and k tile
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lbase = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lptr_t)smem);
    int brow = m0 + (tid & (ROWS - 1));
    brow = brow < M ? brow : M - 1;
    const unsigned char* bsrc = p.XS + ((size_t)(tid / ROWS) * M + brow) * 16;   // issue i of k tile kt: + ((kt * 12 + PPI i) * M) * 16
    const uint4* asrc[GA];
#pragma unroll
    for (int a = 0; a < GA; ++a) {
      const int nt = bx * CT + 4 * a + wv;
      asrc[a] = p.W + (size_t)(nt < NT ? nt : NT - 1) * KT * 64 + smi_wlane(lane, p.wperm);
    }
    auto issue = [&](int kt, int slot) {                 // tile kt (clamped: the extra requests of the last iterations keep the count uniform) -> ring slot
      const int kc = kt < k_hi ? kt : k_hi - 1;
      const uint32_t d = lbase + (uint32_t)(slot * SLOT + wv * 64) * 16;
#pragma unroll
      for (int i = 0; i < NI; ++i) smi_glds16(bsrc + ((size_t)(kc * 12 + PPI * i) * M) * 16, d + 256 * 16 * i);
#pragma unroll
      for (int a = 0; a < GA; ++a) smi_glds16(asrc[a] + (size_t)kc * 64, d + (uint32_t)(PIECES + 4 * a * 64) * 16);
    };
#pragma unroll
    for (int t = 0; t < RING - 1; ++t) issue(k_lo + t, t);
    // RMSNorm factors of the block's rows.  A wave takes 32 rows with all their loads in flight together (one row after the
    // other: 32 dependent L2 round trips, ~50 us in front of the k loop); per row the order of smi_ss_lane_sum + smi_wave_sum.
    constexpr int RPN = ROWS / 4;                       // rows whose factor a wave computes
    auto norm_factors = [&]() {
      const int np = p.npart, last = np - 1;
      if (np <= 64) {                                     // (the prefill GEMM's own RESID epilogue: one partial per n tile)
        float a[RPN];
        const int i0 = lane < np ? lane : last;
#pragma unroll
        for (int j = 0; j < RPN; ++j) {
          int m = m0 + wave * RPN + j;
          m = m < M ? m : M - 1;
          a[j] = p.sspart[(size_t)m * np + i0];
        }
#pragma unroll
        for (int j = 0; j < RPN; ++j) {
          float v = 0.f;
          v += lane < np ? a[j] : 0.f;
          v = smi_wave_sum(v);
          if (lane == 0) rarr[wave * RPN + j] = 1.0f / sqrtf(v / (float)(KT * 32) + p.eps);
        }
      } else if (np <= 256) {
        int idx[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) idx[q] = lane + 64 * q < np ? lane + 64 * q : last;
        for (int r0 = wave * RPN; r0 < wave * RPN + RPN; r0 += 8) {
          float a[8][4];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            int m = m0 + r0 + j;
            m = m < M ? m : M - 1;
            const float* sp = p.sspart + (size_t)m * np;
#pragma unroll
            for (int q = 0; q < 4; ++q) a[j][q] = sp[idx[q]];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) v += lane + 64 * q < np ? a[j][q] : 0.f;
            v = smi_wave_sum(v);
            if (lane == 0) rarr[r0 + j] = 1.0f / sqrtf(v / (float)(KT * 32) + p.eps);
          }
        }
      } else {
        for (int r = wave * RPN; r < wave * RPN + RPN; ++r) {
          int m = m0 + r;
          m = m < M ? m : M - 1;
          float v = smi_ss_lane_sum(p.sspart + (size_t)m * np, np, lane);
          v = smi_wave_sum(v);
          if (lane == 0) rarr[r] = 1.0f / sqrtf(v / (float)(KT * 32) + p.eps);
        }
      }
    };
    if (PRO == PRO_NORM && !TWO) norm_factors();
    // Fragment registers of two k tiles: tile kt + 1's are read from LDS while tile kt's feed the MFMAs (every wave of the block
    // reads at the same time -- right behind the barrier -- so without the second set the LDS phase and the MFMA phase of
    // an iteration add up instead of overlapping: measured 2060 cycles per k tile for 768 of MFMAs).
    struct Frag { uint4 a[NTW]; uint4 b[MTW][3]; };
    Frag f0, f1;
    auto fread = [&](Frag& f, int slot) {
      const uint4* Bb = Bs + (size_t)slot * SLOT;
#pragma unroll
      for (int i = 0; i < NTW; ++i) f.a[i] = Bb[PIECES + (wc * NTW + i) * 64 + lane];
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int s = 0; s < 3; ++s) f.b[mt][s] = Bb[(s * 4 + k8) * ROWS + (mt0 + mt) * 16 + em];
    };
    auto fmma = [&](const Frag& f) {
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, f.b[mt][0]), b1 = __builtin_bit_cast(bf16x8, f.b[mt][1]), b2 = __builtin_bit_cast(bf16x8, f.b[mt][2]);
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
          const bf16x8 a = __builtin_bit_cast(bf16x8, f.a[i]);
          acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b2, acc[i][mt], 0, 0, 0);
          acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc[i][mt], 0, 0, 0);
          acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[i][mt], 0, 0, 0);
        }
      }
    };
    // Iteration kt: read tile kt + 1's fragments (slot rn: landed before the last barrier), request tile kt + RING - 1 into
    // the slot of tile kt - 1 (its reads were waited for before the last barrier), multiply tile kt.  Then: this thread's
    // part of tile kt + 2 has landed (counted vmcnt: the RING - 3 tiles behind it stay in flight across the raw barrier; a
    // __syncthreads would drain them), this wave's LDS reads are done (lgkmcnt(0)), barrier.
#define SMI_PG_SYNC() asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((RING - 3) * G) : "memory")
#ifndef SMI_PG_ABL
#define SMI_PG_ABL 0   // timing-only builds (tools/pg_ablate.sh): 1 no MFMAs, 2 no LDS-DMA in the loop, 4 no epilogue, 8 no fragment reads
#endif
#define SMI_PG_STEP(cur_, nxt_, kt_) do { \
      if (!(SMI_PG_ABL & 8)) fread(nxt_, rn); \
      if (!(SMI_PG_ABL & 2)) issue((kt_) + RING - 1, ws); \
      if (!(SMI_PG_ABL & 1)) fmma(cur_); else { _Pragma("unroll") for (int i_ = 0; i_ < NTW; ++i_) smi_keep4((cur_).a[i_]); _Pragma("unroll") for (int m_ = 0; m_ < MTW; ++m_) { smi_keep4((cur_).b[m_][0]); smi_keep4((cur_).b[m_][1]); smi_keep4((cur_).b[m_][2]); } } \
      if (SEG && !part && kseg > 0 && ((kt_) + 1) % kseg == 0) fold();   /* (block-uniform) a segment ends with this tile */ \
      SMI_PG_SYNC(); \
      rn = rn + 1 == RING ? 0 : rn + 1; \
      ws = ws + 1 == RING ? 0 : ws + 1; \
    } while (0)
    if (SMI_PG_ABL & 8) { f0 = Frag{}; f1 = Frag{}; }
    if constexpr (TWO) {
      // tile kt from slot kt & 1 while tile kt + 1 arrives in the other slot (free: its reads were waited for before the last barrier)
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // tile 0
      for (int kt = 0; kt < KT; ++kt) {
        issue(kt + 1, (kt + 1) & 1);
        fread(f0, kt & 1);
        fmma(f0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if (PRO == PRO_NORM) {                               // the ring is idle now (every request landed, every read done): its memory takes the factors
        norm_factors();
        __syncthreads();
      }
    } else {
    SMI_PG_SYNC();                                        // tiles 0 and 1 (rarr: the fence below)
    __syncthreads();
    fread(f0, 0);
    int rn = 1, ws = RING - 1;
    for (int kt = k_lo; kt < k_hi; kt += 2) {
      SMI_PG_STEP(f0, f1, kt);
      if (kt + 1 < k_hi) SMI_PG_STEP(f1, f0, kt + 1);
      else f0 = f1;                                       // (odd count: the loop ends here; keeps the two paths' live sets alike)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped extra requests: nothing may still write LDS when the block ends
    if constexpr (SEG) {
      if (!part && kseg > 0) {
        if (KT % kseg) fold();                            // the last, shorter segment
#pragma unroll
        for (int a = 0; a < NTW; ++a)
#pragma unroll
          for (int b = 0; b < MTW; ++b) acc[a][b] = tot[a][b];
      }
    }
    }
#undef SMI_PG_STEP
#undef SMI_PG_SYNC
  } else {
  // The six staged pieces are named registers, not an array: as `uint4 sreg[6]` the compiler left them in scratch
  // memory (ScratchSize 112), which put a wait for the global loads right behind their issue -- in front of the k
  // tile's MFMAs instead of behind them.
  static_assert(PIECES / 256 == 6, "six staged pieces per thread");
  uint4 sr0, sr1, sr2, sr3, sr4, sr5;
  auto piece = [&](int kt, int i) -> uint4 {
    const int q = tid + 256 * i;                         // [s][k8][row]
    int row = m0 + (q & (ROWS - 1));
    row = row < M ? row : M - 1;
    return *(const uint4*)(p.XS + xs_off(kt, q / (4 * ROWS), (q / ROWS) & 3, row, M));
  };
#define stage_load(kt_) do { sr0 = piece((kt_), 0); sr1 = piece((kt_), 1); sr2 = piece((kt_), 2); sr3 = piece((kt_), 3); sr4 = piece((kt_), 4); sr5 = piece((kt_), 5); } while (0)
#define stage_store(buf_) do { uint4* d_ = Bs + (buf_) * PIECES + tid; d_[0] = sr0; d_[256] = sr1; d_[512] = sr2; d_[768] = sr3; d_[1024] = sr4; d_[1280] = sr5; } while (0)
  stage_load(0);
  if (PRO == PRO_NORM) {
    for (int r = wave; r < ROWS; r += 4) {
      int m = m0 + r;
      m = m < M ? m : M - 1;
      float v = smi_ss_lane_sum(p.sspart + (size_t)m * p.npart, p.npart, lane);
      v = smi_wave_sum(v);
      if (lane == 0) rarr[r] = 1.0f / sqrtf(v / (float)(KT * 32) + p.eps);
    }
  }
  stage_store(0);
  __syncthreads();
  // Weight tiles run two k tiles ahead of the MFMAs (HBM / L2 latency hidden behind 96 MFMAs per wave).  Three
  // register sets take the roles (in use, next, in flight) in turn -- the k loop is unrolled by three so that no
  // set is ever copied: with `wA = wB; wB = wC` at the loop end the compiler waited for the loads it had just issued.
  uint4 w0[NTW], w1[NTW], w2[NTW];
  const int wl_ = smi_wlane(lane, p.wperm);
#define wload(w_, kt_) do { const int kc_ = (kt_) < KT ? (kt_) : KT - 1; _Pragma("unroll") for (int i = 0; i < NTW; ++i) (w_)[i] = p.W[((size_t)nts[i] * KT + kc_) * 64 + wl_]; } while (0)
#define ktile(kt_, wuse_, wld_) do { \
    const int kq_ = (kt_); \
    wload(wld_, kq_ + 2); \
    stage_load(kq_ + 1 < KT ? kq_ + 1 : KT - 1);         /* in flight during this tile's MFMAs: unconditional (behind a branch the compiler waited for them in front of the MFMAs) and fenced (without the fence it sinks them below the MFMAs) */ \
    __builtin_amdgcn_sched_barrier(0); \
    const uint4* Bb = Bs + (kq_ & 1) * PIECES; \
    /* operand pieces of m-tile mt + 1 are read from LDS while the MFMAs of m-tile mt run */ \
    uint4 bq[2][3]; \
    _Pragma("unroll") for (int s = 0; s < 3; ++s) bq[0][s] = Bb[(s * 4 + k8) * ROWS + mt0 * 16 + em]; \
    _Pragma("unroll") for (int mt = 0; mt < MTW; ++mt) { \
      if (mt + 1 < MTW) { \
        _Pragma("unroll") for (int s = 0; s < 3; ++s) bq[(mt + 1) & 1][s] = Bb[(s * 4 + k8) * ROWS + (mt0 + mt + 1) * 16 + em]; \
      } \
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, bq[mt & 1][0]), b1 = __builtin_bit_cast(bf16x8, bq[mt & 1][1]), \
                   b2 = __builtin_bit_cast(bf16x8, bq[mt & 1][2]); \
      _Pragma("unroll") for (int i = 0; i < NTW; ++i) { \
        const bf16x8 a = __builtin_bit_cast(bf16x8, (wuse_)[i]); \
        acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b2, acc[i][mt], 0, 0, 0); \
        acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc[i][mt], 0, 0, 0); \
        acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[i][mt], 0, 0, 0); \
      } \
    } \
    if (kq_ + 1 < KT) { \
      stage_store((kq_ & 1) ^ 1);                        /* the other buffer was last read two barriers ago */ \
      __syncthreads(); \
    } \
  } while (0)
  wload(w0, 0);
  wload(w1, 1);
  for (int kt = 0; kt < KT; kt += 3) {
    ktile(kt, w0, w2);
    if (kt + 1 < KT) ktile(kt + 1, w1, w0);
    if (kt + 2 < KT) ktile(kt + 2, w2, w1);
  }
#undef ktile
#undef wload
#undef stage_load
#undef stage_store
  }   // RING == 0
#ifdef SMI_PG_ABL
  if (RING != 0 && (SMI_PG_ABL & 4)) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) asm volatile("" ::"v"(acc[i][mt]));
    return;
  }
#endif
  if constexpr (SEG) {
    if (part) {   // split-K form: this segment's sums as they lie in the accumulators (1-KiB stores); k_resid_comb finishes the rows
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        const int nt = (bx * NWC + wc) * NTW + i;
        if (nt >= NT) continue;   // wave-uniform
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const int mtile = (m0 >> 4) + mt0 + mt;
          if (mtile < p.slab_mt)
            p.slab[(((size_t)blockIdx.y * NT + nt) * p.slab_mt + mtile) * 64 + lane] = make_float4(acc[i][mt][0], acc[i][mt][1], acc[i][mt][2], acc[i][mt][3]);
        }
      }
      return;
    }
  }
  // ---- epilogue (same arithmetic as k_gemm's).  Its loads first, all together and from clamped addresses: written inside the
  // `if (valid)` bodies hipcc branched around every load and waited for it on the spot (QKV: 137 `s_waitcnt vmcnt(0)`, three
  // dependent round trips -- bias, row descriptor, RoPE pair -- for each of the 24 (tile, m-tile) pairs of a wave).
  const int N = NT * 16;
  int mrow[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) { const int m = m0 + (mt0 + mt) * 16 + em; mrow[mt] = m < M ? m : M - 1; }
  int ncol[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) { const int nt = (bx * NWC + wc) * NTW + i; ncol[i] = (nt < NT ? nt : NT - 1) * 16 + 4 * (lane >> 4); }
  float4 pre_h[EPI == EPI_RESID ? NTW : 1][EPI == EPI_RESID ? MTW : 1], pre_g[EPI != EPI_SWIGLU ? NTW : 1];
  RowDesc pre_rd[EPI == EPI_QKV ? MTW : 1];
  float2 pre_c[EPI == EPI_QKV ? NTW : 1][EPI == EPI_QKV ? MTW : 1][2];
  if constexpr (EPI == EPI_RESID) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      pre_g[i] = *(const float4*)(p.gamma_next + ncol[i]);
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) pre_h[i][mt] = *(const float4*)(p.Y + (size_t)mrow[mt] * N + ncol[i]);
    }
  } else if constexpr (EPI == EPI_QKV) {
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) pre_rd[mt] = p.rows[mrow[mt]];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      pre_g[i] = *(const float4*)(p.bias + ncol[i]);
      const int i0 = (ncol[i] & 63) >> 1;
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        pre_c[i][mt][0] = p.rope[(size_t)pre_rd[mt].pos * 32 + i0];
        pre_c[i][mt][1] = p.rope[(size_t)pre_rd[mt].pos * 32 + i0 + 1];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int nt = (bx * NWC + wc) * NTW + i;
    if (nt >= NT) continue;   // wave-uniform
    const int n = nt * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int m = m0 + (mt0 + mt) * 16 + em;
      const bool valid = m < M;
      float4 s = make_float4(acc[i][mt][0], acc[i][mt][1], acc[i][mt][2], acc[i][mt][3]);
      if (PRO == PRO_NORM) {
        const float r = rarr[(mt0 + mt) * 16 + em];
        s.x *= r; s.y *= r; s.z *= r; s.w *= r;
      }
      if (EPI == EPI_RESID) {
        float ssq = 0.f;
        if (valid) {
          float4 h = pre_h[i][mt];
          h.x += s.x; h.y += s.y; h.z += s.z; h.w += s.w;
          *(float4*)(p.Y + (size_t)m * N + n) = h;
          ssq = (h.x * h.x + h.y * h.y) + (h.z * h.z + h.w * h.w);
          const float4 g = pre_g[i];
          const float t[4] = {g.x * h.x, g.y * h.y, g.z * h.z, g.w * h.w};
          uint32_t hi[4], mi[4], lo[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split3(t[e], hi[e], mi[e], lo[e]);
          const size_t o = xs_off(n >> 5, 0, (n >> 3) & 3, m, M) + ((n >> 2) & 1) * 8;
          const size_t pl = (size_t)4 * M * 16;
          *(uint2*)(p.XSout + o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
          *(uint2*)(p.XSout + o + pl) = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
          *(uint2*)(p.XSout + o + 2 * pl) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
        }
        ssq += __shfl_xor(ssq, 16, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (lane < 16 && valid) p.ssout[(size_t)m * NT + nt] = ssq;
      } else if (EPI == EPI_SWIGLU) {
        if (valid) {
          const float a0 = (s.x / (1.0f + expf(-s.x))) * s.y;
          const float a1 = (s.z / (1.0f + expf(-s.z))) * s.w;
          uint32_t h0, q0, l0, h1, q1, l1;
          split3(a0, h0, q0, l0);
          split3(a1, h1, q1, l1);
          const int k = n >> 1;
          const size_t o = xs_off(k >> 5, 0, (k >> 3) & 3, m, M) + ((k >> 1) & 3) * 4;
          const size_t pl = (size_t)4 * M * 16;
          *(uint32_t*)(p.XSout + o) = h0 | (h1 << 16);
          *(uint32_t*)(p.XSout + o + pl) = q0 | (q1 << 16);
          *(uint32_t*)(p.XSout + o + 2 * pl) = l0 | (l1 << 16);
        }
      } else if (EPI == EPI_QKV) {
        if (valid) {
          const float4 b = pre_g[i];
          s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
          const RowDesc rd = pre_rd[mt];
          if (n < p.q_dim + p.kv_dim) {
            const float2 c0 = pre_c[i][mt][0], c1 = pre_c[i][mt][1];
            float4 r;
            r.x = __fadd_rn(__fmul_rn(s.x, c0.x), __fmul_rn(-s.y, c0.y));
            r.y = __fadd_rn(__fmul_rn(s.y, c0.x), __fmul_rn(s.x, c0.y));
            r.z = __fadd_rn(__fmul_rn(s.z, c1.x), __fmul_rn(-s.w, c1.y));
            r.w = __fadd_rn(__fmul_rn(s.w, c1.x), __fmul_rn(s.z, c1.y));
            s = r;
          }
          if (n < p.q_dim) {
            *(float4*)(p.Y + (size_t)m * p.q_dim + n) = s;
          } else {
            const bool isk = n < p.q_dim + p.kv_dim;
            const int c = n - p.q_dim - (isk ? 0 : p.kv_dim);
            const size_t off = kv_row(p.km, rd.slot, c >> 6, p.n_kv, p.max_pos, rd.pos) * 64 + (c & 63);
            void* base = isk ? p.kcache : p.vcache;
            if (KVF32) {
              *(float4*)((float*)base + off) = s;
            } else {
              uint2 pk;
              pk.x = smi_f32_to_bf16(s.x) | (smi_f32_to_bf16(s.y) << 16);
              pk.y = smi_f32_to_bf16(s.z) | (smi_f32_to_bf16(s.w) << 16);
              *(uint2*)((uint16_t*)base + off) = pk;
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// lm_head for up to 16 rows: persistent blocks.  The activation triples of the wave's k tiles are
// loaded once and stay in registers; the block then streams groups of two 16-row vocabulary tiles
// (next group's weights requested as soon as the MFMAs have consumed the current ones) and keeps
// a running arg-max per row.  Same k-tile -> wave map, accumulator chains and reduction order as
// k_gemm<EPI_LM>, so the logits are bit-identical to the two-m-tile path.
// ------------------------------------------------------------------------------------------
// HV = 2: two groups of four waves share a block and the weight stream (the second group's tile loads hit L2); group h
// owns rows m0 + 16 h .. and does exactly what the HV = 1 kernel does for 16 rows, so the logits are the same bits.
template <int HV>
__global__ __launch_bounds__(256 * HV) void k_lm(GemmP p, int ngroups, int m0_launch) {
  constexpr int NTB = 2, NW = 4, U = 8, MT = 1;
  constexpr size_t kHalfBytes = (size_t)NW * NTB * MT * 1024 + 32 * 4 + NTB * 32 * 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int half = HV == 2 ? (int)threadIdx.x >> 8 : 0;
  unsigned char* smem = smem_all + half * kHalfBytes;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int KT = p.KT, NT = p.NT;
  const int m0 = m0_launch + 16 * half;
  const int M = p.M - m0 < 16 ? (p.M - m0 > 0 ? p.M - m0 : 0) : 16;   // this wave group: rows m0 .. m0 + M - 1 (none: it only keeps the barriers)
  float4* red = (float4*)smem;                                   // [NW][NTB][64]
  float* rarr = (float*)(smem + (size_t)NW * NTB * MT * 1024);
  float* bestv = rarr + 32;                                      // [NTB][32] running best of this block
  int* besti = (int*)(bestv + NTB * 32);
  const int k8 = lane >> 4, em = lane & 15;
  const int mrow = em < M ? em : (M > 0 ? M - 1 : 0);
  const int xrow = m0 + mrow < p.M ? m0 + mrow : p.M - 1;
  bf16x8 bf[U][3];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int j = wave + u * NW;
    j = j < KT ? j : KT - 1;
#pragma unroll
    for (int s = 0; s < 3; ++s) bf[u][s] = *(const bf16x8*)(p.XS + xs_off(j, s, k8, xrow, p.M));
  }
  uint4 w[U][NTB];
  auto load_w = [&](int g) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int j = wave + u * NW;
      j = j < KT ? j : KT - 1;
#pragma unroll
      for (int nb = 0; nb < NTB; ++nb) {
        int nt = g * NTB + nb;
        nt = nt < NT ? nt : NT - 1;
        w[u][nb] = smi_ldw(&p.W[((size_t)nt * KT + j) * 64 + lane]);
      }
    }
  };
  int g = blockIdx.x;
  if (g < ngroups) load_w(g);
  for (int m = wave; m < M; m += NW) {
    float v = smi_ss_lane_sum(p.sspart + (size_t)(m0 + m) * p.npart, p.npart, lane);
    v = smi_wave_sum(v);
    if (lane == 0) rarr[m] = 1.0f / sqrtf(v / (float)(KT * 32) + p.eps);
  }
  for (int i = tid; i < NTB * 32; i += 256) { bestv[i] = -INFINITY; besti[i] = 0x7fffffff; }
  __syncthreads();
  const float rn = rarr[mrow];
  for (; g < ngroups; g += gridDim.x) {
    f32x4 acc[3][NTB];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int a = 0; a < NTB; ++a) acc[c][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (wave + u * NW < KT) {
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
          const bf16x8 a = __builtin_bit_cast(bf16x8, w[u][nb]);
          acc[2][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[u][2], acc[2][nb], 0, 0, 0);
          acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[u][1], acc[1][nb], 0, 0, 0);
          acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[u][0], acc[0][nb], 0, 0, 0);
        }
      }
    }
    const int gn = g + gridDim.x;
    if (gn < ngroups) load_w(gn);          // overlaps the reduction and the epilogue below
#pragma unroll
    for (int nb = 0; nb < NTB; ++nb) {
      const f32x4 t = (acc[2][nb] + acc[1][nb]) + acc[0][nb];
      red[(wave * NTB + nb) * 64 + lane] = make_float4(t[0], t[1], t[2], t[3]);
    }
    __syncthreads();
    if (wave < NTB) {
      const int nb = wave, nt = g * NTB + nb;
      if (nt < NT) {
        float4 s = red[(0 * NTB + nb) * 64 + lane];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) {
          const float4 t = red[(wv * NTB + nb) * 64 + lane];
          s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        s.x *= rn; s.y *= rn; s.z *= rn; s.w *= rn;
        const int n = nt * 16 + 4 * (lane >> 4);
        const bool valid = em < M;
        if (valid && p.Y) {
          float* y = p.Y + (size_t)(m0 + em) * p.V + n;
          if (n + 0 < p.V) y[0] = s.x;
          if (n + 1 < p.V) y[1] = s.y;
          if (n + 2 < p.V) y[2] = s.z;
          if (n + 3 < p.V) y[3] = s.w;
        }
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.V && sv[r] > bv) { bv = sv[r]; bi = n + r; }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
          const float ov = __shfl_xor(bv, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane < 16 && valid) {
          const float cv = bestv[nb * 32 + em];
          const int ci = besti[nb * 32 + em];
          if (bv > cv || (bv == cv && bi < ci)) { bestv[nb * 32 + em] = bv; besti[nb * 32 + em] = bi; }
        }
      }
    }
    __syncthreads();   // red is rewritten by the next group
  }
  if (tid < M) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int nb = 0; nb < NTB; ++nb) {
      const float ov = bestv[nb * 32 + tid];
      const int oi = besti[nb * 32 + tid];
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    p.pval[(size_t)(m0 + tid) * gridDim.x + blockIdx.x] = bv;
    p.pidx[(size_t)(m0 + tid) * gridDim.x + blockIdx.x] = bi;
  }
}


// lm_head for 17..32 rows per pass: ONE 4-wave group streams the weights once for both m-tiles.  Rows m0..m0+15 keep their
// operand triples in registers exactly as in k_lm<1>; rows m0+16..m0+31 keep theirs in LDS (86 KB at K = 896: [k tile][3][4][16][16 B]),
// read as MFMA B operands straight from there (3 KB of LDS reads per 1 KB weight tile -- far below the LDS rate).  k_lm<2>
// gave each 16-row group its own four waves, i.e. every weight tile was pulled into registers twice per CU (72 us at 32
// rows); here it is pulled once.  Same k-tile -> wave map, chains and reduction order per row as k_lm<1>: same logits bits.
// Round 4: TWO weight register sets (one wave per SIMD: the 512-register budget holds them).  With one set a group's tiles were
// requested only after the previous group's 84 MFMAs per wave had consumed the registers, so the stream stood still during the
// MFMAs, the reduction and the epilogue (3.2 us per group against 2.3 us for the 56 KB at the CU's share of the HBM rate); now
// the tiles of group g + 2 are requested as soon as group g's MFMAs are issued and the tiles of g + 1 are already in flight.
// UW = k tiles per wave (ceil(KT / 4)): no clamped duplicate loads (U = 8 at KT = 28 re-requested tile 27 once per wave and n tile).
template <int UW>
__global__ __launch_bounds__(256) void k_lm32(GemmP p, int ngroups, int m0) {
  constexpr int NTB = 2, NW = 4, U = UW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KT = p.KT, NT = p.NT;
  const int M = p.M - m0 < 32 ? p.M - m0 : 32;                    // rows of this pass (>= 1)
  float4* red = (float4*)smem;                                     // [NW][NTB][2][64]
  float* rarr = (float*)(smem + (size_t)NW * NTB * 2 * 1024);      // [32]
  float* bestv = rarr + 32;                                        // [NTB][32]
  int* besti = (int*)(bestv + NTB * 32);
  uint4* xl = (uint4*)(smem + (size_t)NW * NTB * 2 * 1024 + 32 * 4 + NTB * 32 * 8);   // [KT][3][4][16]
  const int k8 = lane >> 4, em = lane & 15;
  const int row0 = m0 + (em < M ? em : M - 1);                     // m-tile 0 row of this lane (clamped)
  // ---- prologue: EVERY load of it is requested before anything waits -- the RMSNorm partials of this wave's rows (8 rows x 4
  // dwords), the m-tile-0 operand registers, the 21 pieces per thread of m-tile 1's LDS image, then the first two groups' weight
  // tiles.  As first written (round 3) the image was copied load -> wait -> ds_write per piece and the partials row by row: 29
  // dependent round trips behind the cold weight tiles, 14 of the kernel's 65 us (round-4 stamps: block 0 streams its 20 groups
  // in 50 us).  Loads return in issue order, so the small L2-resident ones go first.
  constexpr int RPW = 8;                                           // rows per wave (32 / NW)
  float ssv[RPW][4];
  const bool ss_early = p.npart <= 256;
  if (ss_early) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      int mr = m0 + wave + i * NW;
      mr = mr < p.M ? mr : p.M - 1;
      const float* sp = p.sspart + (size_t)mr * p.npart;
      const int last = p.npart - 1;
      ssv[i][0] = sp[lane < last ? lane : last]; ssv[i][1] = sp[lane + 64 < last ? lane + 64 : last];
      ssv[i][2] = sp[lane + 128 < last ? lane + 128 : last]; ssv[i][3] = sp[lane + 192 < last ? lane + 192 : last];
    }
  }
  bf16x8 bf[U][3];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int j = wave + u * NW;
    j = j < KT ? j : KT - 1;
#pragma unroll
    for (int s = 0; s < 3; ++s) bf[u][s] = *(const bf16x8*)(p.XS + xs_off(j, s, k8, row0, p.M));
  }
  constexpr int XPT = (UW * NW * 12 * 16 + 255) / 256;             // LDS-image pieces per thread (KT <= UW * NW)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 xpc[XPT];
  const int nxl = KT * 12 * 16;
#pragma unroll
  for (int q = 0; q < XPT; ++q) {
    int i = tid + 256 * q;
    i = i < nxl ? i : nxl - 1;
    const int r = i & 15, pcx = i >> 4;                            // pcx = (kt * 3 + s) * 4 + k8
    int mr = m0 + 16 + r;
    mr = mr < p.M ? mr : p.M - 1;
    xpc[q] = *(const u32x4*)(p.XS + ((size_t)pcx * p.M + mr) * 16);
  }
  uint4 w0[U][NTB], w1[U][NTB];
  // EVERY request below is unconditional (group and tile indices are clamped, a block's surplus requests re-read the last group
  // out of L2): with `if (gn < ngroups) load_w(..)` the number of loads in flight depends on the path, the compiler's counted
  // waits must assume the smallest -- and the wait for one register set then drained the other set's requests too (round-3's
  // "two register sets change nothing", and this kernel's first round-4 build: s_waitcnt vmcnt(13) .. vmcnt(0) in front of a
  // group's MFMAs with 28 loads outstanding).
  auto load_w = [&](uint4 (&w)[U][NTB], int gq) {
    const int g = gq < ngroups ? gq : ngroups - 1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int j = wave + u * NW;
      j = j < KT ? j : KT - 1;
#pragma unroll
      for (int nb = 0; nb < NTB; ++nb) {
        int nt = g * NTB + nb;
        nt = nt < NT ? nt : NT - 1;
        w[u][nb] = smi_ldw(&p.W[((size_t)nt * KT + j) * 64 + lane]);
      }
    }
  };
  const int G = (int)gridDim.x;
  int g = blockIdx.x;
  load_w(w0, g);
  load_w(w1, g + G);
  // m-tile 1's operand pieces -> LDS (rows beyond M repeat the last row; their results are never stored)
#pragma unroll
  for (int q = 0; q < XPT; ++q) asm volatile("" : "+v"(xpc[q]));   // (the loads stay above the weight requests, not sunk into the guarded stores)
#pragma unroll
  for (int q = 0; q < XPT; ++q) {
    const int i = tid + 256 * q;
    if (i < nxl) *(u32x4*)&xl[i] = xpc[q];
  }
  if (ss_early) {   // same order as smi_ss_lane_sum + smi_wave_sum
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int m = wave + i * NW;
      float v = lane < p.npart ? ssv[i][0] : 0.f;
      v += lane + 64 < p.npart ? ssv[i][1] : 0.f;
      v += lane + 128 < p.npart ? ssv[i][2] : 0.f;
      v += lane + 192 < p.npart ? ssv[i][3] : 0.f;
      v = smi_wave_sum(v);
      if (lane == 0 && m < M) rarr[m] = 1.0f / sqrtf(v / (float)(KT * 32) + p.eps);
    }
  } else {
    for (int m = wave; m < M; m += NW) {
      float v = smi_ss_lane_sum(p.sspart + (size_t)(m0 + m) * p.npart, p.npart, lane);
      v = smi_wave_sum(v);
      if (lane == 0) rarr[m] = 1.0f / sqrtf(v / (float)(KT * 32) + p.eps);
    }
  }
  for (int i = tid; i < NTB * 32; i += 256) { bestv[i] = -INFINITY; besti[i] = 0x7fffffff; }
  __syncthreads();
  const int nb_e = wave & 1, mt_e = wave >> 1;                     // the (n tile, m tile) this wave finishes
  const int ml = mt_e * 16 + em;                                   // its row (local)
  const float rn = rarr[ml < M ? ml : 0];
  int gi = 0;   // (diagnostics) groups this block has finished
#ifdef SMI_DIAG
#define SMI_LMSTAMP(i) do { if (p.stamps && blockIdx.x == 0 && tid == 0 && gi < 32) p.stamps[gi * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SMI_LMSTAMP(i) do { } while (0)
#endif
  auto group = [&](uint4 (&w)[U][NTB], int gcur) {
    SMI_LMSTAMP(0);
    f32x4 acc[3][NTB][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int a = 0; a < NTB; ++a) { acc[c][a][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[c][a][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = wave + u * NW;
      if (j < KT) {
        bf16x8 b1[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) b1[s] = __builtin_bit_cast(bf16x8, xl[((j * 3 + s) * 4 + k8) * 16 + em]);
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
          const bf16x8 a = __builtin_bit_cast(bf16x8, w[u][nb]);
          acc[2][nb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[u][2], acc[2][nb][0], 0, 0, 0);
          acc[1][nb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[u][1], acc[1][nb][0], 0, 0, 0);
          acc[0][nb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[u][0], acc[0][nb][0], 0, 0, 0);
          acc[2][nb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1[2], acc[2][nb][1], 0, 0, 0);
          acc[1][nb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1[1], acc[1][nb][1], 0, 0, 0);
          acc[0][nb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1[0], acc[0][nb][1], 0, 0, 0);
        }
      }
    }
    load_w(w, gcur + 2 * G);               // this set's registers are free again: the group after next, while the other set's tiles are in flight
#pragma unroll
    for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const f32x4 t = (acc[2][nb][mt] + acc[1][nb][mt]) + acc[0][nb][mt];
        red[((wave * NTB + nb) * 2 + mt) * 64 + lane] = make_float4(t[0], t[1], t[2], t[3]);
      }
    SMI_LMSTAMP(1);
    __syncthreads();
    SMI_LMSTAMP(2);
    {
      const int nt = gcur * NTB + nb_e;
      if (nt < NT) {
        float4 s = red[((0 * NTB + nb_e) * 2 + mt_e) * 64 + lane];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) {
          const float4 t = red[((wv * NTB + nb_e) * 2 + mt_e) * 64 + lane];
          s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        s.x *= rn; s.y *= rn; s.z *= rn; s.w *= rn;
        const int n = nt * 16 + 4 * (lane >> 4);
        const bool valid = ml < M;
        if (valid && p.Y) {
          float* y = p.Y + (size_t)(m0 + ml) * p.V + n;
          if (n + 0 < p.V) y[0] = s.x;
          if (n + 1 < p.V) y[1] = s.y;
          if (n + 2 < p.V) y[2] = s.z;
          if (n + 3 < p.V) y[3] = s.w;
        }
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.V && sv[r] > bv) { bv = sv[r]; bi = n + r; }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
          const float ov = __shfl_xor(bv, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane < 16 && valid) {
          const float cv = bestv[nb_e * 32 + ml];
          const int ci = besti[nb_e * 32 + ml];
          if (bv > cv || (bv == cv && bi < ci)) { bestv[nb_e * 32 + ml] = bv; besti[nb_e * 32 + ml] = bi; }
        }
      }
    }
    __syncthreads();   // red is rewritten by the next group
    SMI_LMSTAMP(3);
    ++gi;
    (void)gi;
  };
#undef SMI_LMSTAMP
  for (; g < ngroups; g += 2 * G) {   // block-uniform trip counts (every wave keeps the barriers); a group beyond the last stores nothing
    group(w0, g);
    group(w1, g + G);
  }
  if (tid < M) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int nb = 0; nb < NTB; ++nb) {
      const float ov = bestv[nb * 32 + tid];
      const int oi = besti[nb * 32 + tid];
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    p.pval[(size_t)(m0 + tid) * gridDim.x + blockIdx.x] = bv;
    p.pidx[(size_t)(m0 + tid) * gridDim.x + blockIdx.x] = bi;
  }
}

struct AttnP {
  const float* q;      // [M][q_dim]
  const void* kcache;  // layer base
  const void* vcache;
  const RowDesc* rows;
  unsigned char* xs_out;  // o_proj operand: exact bf16 triples [q_dim/32][3][4][M][16 B]
  int M;
  int q_dim, n_kv, group, max_pos, n_heads;
  KvMap km;            // paged KV cache (general kernel and k_attn_pf only; the slot_is_row kernels need contiguous slots)
  int slot_is_row;     // every row m lives in KV slot m (decode): addresses need no descriptor
  int work_blocks;     // = n_heads * M * nseg; later blocks are prefetch helpers
  PfDesc pf;
  // long contexts: the keys are cut into fixed segments of kAttnSeg tokens, one block per (head, row, segment); each
  // block leaves (max, sum, out[64]) in `part` and k_attn_merge combines a row's segments in order.  nseg == 1: no split.
  int nseg;
  float* part;         // [M][n_heads][nseg][66]
  // fused o_proj (one-row kernel): this head's two k tiles of W_o against its own output -> per-head partial of the o_proj
  const uint4* Wo;     // [NTo][2 * n_heads][64] tiles (head-interleaved k order)
  int NTo;             // n tiles of W_o (hidden / 16)
  float* part_o;       // [n_heads][NTo * 16] f32 (FUSE = 2: [row][n_heads][NTo * 16])
  // FUSE = 2 (2 .. 8 rows): the last of a (row, quarter)'s n_heads blocks to arrive adds the heads' partials in order and runs
  // the RESID epilogue of the o_proj for that quarter of the row's columns
  unsigned int* cnt;   // [rows][kFuseQB] arrival counters (zero between launches: the last arriver resets its counter)
  float* h;            // [rows][hidden] residual stream: read and rewritten (h += o_proj)
  const float* gamma_next;   // the post-attention norm's weight
  unsigned char* xs_next;    // its operand triples [hidden / 32][3][4][M][16 B]
  float* ssout;        // [rows][NTo * 4] partial sums of squares of the new residual rows
};

constexpr int kAttnWaves = 8;
constexpr int kAttnSeg = 1024;   // tokens per segment (a multiple of the chunk of either KV type)
constexpr int kFuseQB = 4;       // fused o_proj: blocks per head -- each repeats the head's attention (latency-bound, the CUs are idle anyway)
                                 // and takes a quarter of W_o's n tiles: a CU pulls 28 KB of cold weights instead of 112 KB
constexpr int kFuse2Max = 8;     // rows up to which several live rows take the fused attention + o_proj (FUSE = 2: last-arriver head sum)
constexpr int kFuseOT = 2;       // W_o n tiles per wave (8 waves x 2 x 4 blocks x 16 = hidden sizes up to 1024)

// 8 waves; a group of LPT lanes owns one token per pass (16-byte K and V pieces per lane, dot product
// reduced on the DPP path), UNR passes of loads in flight together (256 tokens per chunk with bf16
// KV).  Softmax runs per chunk against a block-wide running max (one LDS word per wave, one barrier),
// so every lane accumulates at the same scale; the final merge is a plain sum: inside the wave through
// its private LDS slice, across waves through one more barrier.  Little redundant work per thread:
// with one block per head the kernel is bound by instruction issue of its own waves.
// ONE == 1: exactly one live row in KV slot 0 and one context segment (batch-1 decode): index arithmetic folded at compile
// time.  ONE == 2: several rows, row m in KV slot m, one context segment (plain batched decode): the K/V addresses do not
// depend on the row descriptor, so descriptor, q and the first K/V chunk travel together as in the one-row kernel.
// FUSE (with ONE == 1): the block goes on to multiply its head's output with that head's two k tiles of W_o (requested at
// kernel entry, in flight under the whole attention) and leaves a per-head partial of the o_proj; the consumer
// (k_gemm<PRO_FUSEDO>: gate_up) adds the heads in order, which is the order k_gemm<RESID> with NW = n_heads sums its waves in:
// one kernel boundary and one cold weight stream less per layer, same bits.
// The W_o tiles are held by eight EXTRA waves (8..15) that do nothing else: they request their tiles at entry, sit at the
// block's two barriers while waves 0..7 run the attention exactly as in the plain kernel, then multiply and store.  (Issued
// by the attention waves themselves, the cold weight loads either hold q / K / V back -- loads return in issue order -- or,
// issued behind them, are caught by the compiler's counted waits for K / V.)
// PG (with ONE != 0): the KV cache is paged -- a token's row comes through the slot's page-table row (one more dependent,
// cache-resident load in front of the K/V loads; every table entry is a valid page at all times, so the unconditional first
// chunk stays safe), everything else as the slot == row kernels.
// FUSE == 2 (with ONE == 2; round 4): the fused o_proj for 2 .. 8 live rows.  Four blocks per (row, head) as at one row; a block
// leaves its per-head partial of a quarter of the row's columns in part_o with write-through (sc1) stores, drains them, and -- behind
// a workgroup barrier -- one lane counts the block in on the (row, quarter)'s counter with an agent-scope atomic add.  The block whose
// add returns n_heads - 1 is the last: its first 56 lanes read the n_heads partials of their 4 columns with sc1 loads (the guide's
// hand-off form: every payload byte stored sc1 and drained before the count, every read of it an sc1 load behind the add), add
// them in head order -- the order k_gemm<RESID> with NW = n_heads adds its waves, and gate_up's one-row prologue its partials --
// and run the RESID epilogue (residual add, h, the next norm's operand triples, partial sums of squares).  No block ever waits for
// another: nothing can dead-lock and there is nothing to time out.  The o_proj launch (and its boundary) is gone for these row
// counts; gate_up reads an ordinary operand (with more rows the per-head partials would be re-read by every gate_up block).
template <int KVF32, int ONE = 0, int FUSE = 0, int PG = 0>
__global__ __launch_bounds__((FUSE ? 2 : 1) * kAttnWaves * 64) void k_attn(AttnP p) {
  static_assert(!FUSE || (FUSE == 1 && ONE == 1) || (FUSE == 2 && ONE == 2), "fused o_proj: one row (per-head partials for gate_up) or 2..8 rows (last-arriver head sum)");
  constexpr int LPT = KVF32 ? 16 : 8;   // lanes per token row (each lane 16 bytes)
  constexpr int DPL = kHeadDim / LPT;   // dims per lane
  constexpr int TPW = 64 / LPT;         // tokens per wave per pass
  constexpr int NGRP = kAttnWaves * TPW;  // tokens per block per pass
  constexpr int UNR = 4;                // passes whose K/V loads are issued together
  constexpr float NEG = -1e30f;
  constexpr float LOG2E = 1.4426950408889634f;
  __shared__ float wmax[kAttnWaves];
  __shared__ __attribute__((aligned(16))) float so[kAttnWaves][TPW][kHeadDim];   // wave-private merge slices
  __shared__ float sl[kAttnWaves][TPW];
  __shared__ float pw[kAttnWaves][kHeadDim], pl[kAttnWaves];
  __shared__ __attribute__((aligned(16))) unsigned char xsl[FUSE ? 2 * 3 * 4 * 16 : 16];   // FUSE: this head's output as B-operand pieces
  __shared__ unsigned int s_last;   // FUSE == 2: this block was the last of its (row, quarter) to arrive
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= p.work_blocks) {
    pf_run(p.pf, (int)blockIdx.x - p.work_blocks, (int)gridDim.x - p.work_blocks, tid, (FUSE ? 2 : 1) * kAttnWaves * 64);
    return;
  }
  // FUSE == 2: natural block order (row, quarter, head), head fastest -- the 7 query heads of a KV group and the four quarter
  // blocks of a row are neighbours; XCD x takes the x-th contiguous eighth of it (as the unfused batched kernel does)
  unsigned fbid = blockIdx.x;
  if constexpr (FUSE == 2) {
    const unsigned total = (unsigned)p.work_blocks, xcd = fbid & 7u, slot = fbid >> 3, q8 = total >> 3, r8 = total & 7u;
    fbid = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
  }
  const int f_head = FUSE ? (int)(fbid % (unsigned)p.n_heads) : 0, f_q = FUSE ? (int)(fbid / (unsigned)p.n_heads) % kFuseQB : 0,
            f_row = FUSE == 2 ? (int)(fbid / (unsigned)(p.n_heads * kFuseQB)) : 0;
  if constexpr (FUSE) {
    if (wave >= kAttnWaves) {   // the o_proj waves: tiles (nt, half * n_heads + head), nt = fq * fper + (wave - 8) + 8 i
      const int fq = f_q, fper = p.NTo / kFuseQB, fh = f_head;
      const int ow = wave - kAttnWaves, KTo = 2 * p.n_heads;
      uint4 wo[kFuseOT][2];
#pragma unroll
      for (int i = 0; i < kFuseOT; ++i) {
        int nl = ow + kAttnWaves * i;
        nl = nl < fper ? nl : fper - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) wo[i][j] = smi_ldw(&p.Wo[((size_t)(fq * fper + nl) * KTo + j * p.n_heads + fh) * 64 + lane]);
      }
      __syncthreads();   // the attention waves' merge barrier
      __syncthreads();   // the head's output is in xsl
      // MFMA columns 0 / 1 / 2 are the row's hi / mid / lo split terms (one MFMA per k tile instead of three; k_gemm LEAN == 2)
      bf16x8 bo[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bo[j] = *(const bf16x8*)(xsl + (size_t)((j * 3 + ((lane & 15) < 3 ? (lane & 15) : 0)) * 4 + (lane >> 4)) * 16);
#pragma unroll
      for (int i = 0; i < kFuseOT; ++i) {
        const int nl = ow + kAttnWaves * i, nt = fq * fper + nl;
        if (nl < fper) {   // wave-uniform
          f32x4 a0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 2; ++j)   // the order of k_gemm's chains: tile by tile, lo / mid / hi in their own accumulators (here: columns)
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wo[i][j]), bo[j], a0, 0, 0, 0);
          f32x4 t;
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = (smi_dpp<0x102>(a0[r]) + smi_dpp<0x101>(a0[r])) + a0[r];   // (lo + mid) + hi
          if ((lane & 15) == 0) {
            float* dst = p.part_o + ((size_t)f_row * p.n_heads + fh) * (p.NTo * 16) + nt * 16 + 4 * (lane >> 4);
            if constexpr (FUSE == 2) {   // write-through: the last arriver reads these from another CU inside this launch
#pragma unroll
              for (int r = 0; r < 4; ++r) __hip_atomic_store((uint32_t*)dst + r, __float_as_uint(t[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
              *(float4*)dst = make_float4(t[0], t[1], t[2], t[3]);
            }
          }
        }
      }
      if constexpr (FUSE != 2) return;
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partial stores have left before the block is counted in
    }
  }
  if (!FUSE || wave < kAttnWaves) {   // ---- the attention waves (FUSE == 2: the o_proj waves skip to the arrival protocol below)
  // Several rows: the query heads of one KV group (and a row's segments) are neighbours in the natural block order, but
  // consecutive block ids go to the 8 XCDs round-robin, so each of a group's 7 heads would pull the same K/V rows into a
  // different L2.  Re-number: XCD x works through the x-th contiguous eighth of the natural order.
  unsigned bid = blockIdx.x;
  if (ONE != 1) {
    const unsigned total = (unsigned)p.work_blocks, xcd = bid & 7u, slot = bid >> 3, q8 = total >> 3, r8 = total & 7u;
    bid = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
  }
  const int head = FUSE == 2 ? f_head : ONE == 1 ? (FUSE ? (int)bid % p.n_heads : (int)bid) : ONE == 2 ? (int)bid % p.n_heads : (int)(bid / p.nseg) % p.n_heads,
            m = FUSE == 2 ? f_row : ONE == 1 ? 0 : ONE == 2 ? (int)bid / p.n_heads : (int)bid / (p.nseg * p.n_heads),
            seg = ONE ? 0 : (int)bid % p.nseg;
  const int tl = lane / LPT, dl = lane % LPT;
  const int grp = wave * TPW + tl;
  const RowDesc rd = p.rows[m];
  const int kvh = head / p.group;
  // decode rows use slot == row, so the K/V addresses do not depend on the descriptor and the first
  // chunk's loads leave together with it (positions beyond ctx are valid cache memory, masked later)
  const bool sir = ONE || p.slot_is_row;
  const int slot = sir ? m : rd.slot;
  const size_t rowbase = ((size_t)slot * p.n_kv + kvh) * p.max_pos;
  const int ctx_all = rd.pos + 1;
  int ctx = ctx_all < (seg + 1) * kAttnSeg ? ctx_all : (seg + 1) * kAttnSeg;   // this block: keys [seg * kAttnSeg, ctx)
  if (!ONE && seg * kAttnSeg >= ctx_all) return;   // block-uniform: this row has no such segment (ONE: a single segment -- and no wait for the descriptor before the q / K / V loads are requested)

  // q is requested here but first USED after the chunk's K/V loads have been requested (the scaling sits in the loop behind
  // an opaque asm so that it is not hoisted back): descriptor, q and the first K/V chunk share one memory round trip
  // instead of queueing up as descriptor -> q -> K/V
  float qraw[DPL];
  {
    const float* qp = p.q + (size_t)m * p.q_dim + head * kHeadDim + dl * DPL;
#pragma unroll
    for (int i = 0; i < DPL; ++i) qraw[i] = qp[i];
  }
  float m_run = NEG, lrun = 0.f, o[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) o[i] = 0.f;

  int c0 = seg * kAttnSeg;
  do {   // block-uniform trip count; the first chunk never waits for ctx before its loads
    uint4 kr[UNR], vr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = c0 + u * NGRP + grp;
      const int tc = sir ? (t < p.max_pos ? t : p.max_pos - 1) : (t < ctx ? t : ctx - 1);
      const size_t off = ((ONE && !PG) ? rowbase + tc : kv_row(p.km, slot, kvh, p.n_kv, p.max_pos, tc)) * kHeadDim + dl * DPL;
      if (KVF32) {
        kr[u] = *(const uint4*)((const float*)p.kcache + off);
        vr[u] = *(const uint4*)((const float*)p.vcache + off);
      } else {
        kr[u] = *(const uint4*)((const uint16_t*)p.kcache + off);
        vr[u] = *(const uint4*)((const uint16_t*)p.vcache + off);
      }
    }
    if (ONE) {   // the descriptor is first looked at here, behind the loads above (same opaque-asm device as for q below)
      int pr = rd.pos;
      asm volatile("" : "+v"(pr));
      ctx = pr + 1 < kAttnSeg ? pr + 1 : kAttnSeg;
    }
    float qv[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      float t = qraw[i];
      asm volatile("" : "+v"(t));
      qv[i] = t * 0.125f;  // head_dim^-0.5, exact
    }
    float sc[UNR];
    float lmax = NEG;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const uint32_t ku[4] = {kr[u].x, kr[u].y, kr[u].z, kr[u].w};
      float d = 0.f;
      if (KVF32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) d += qv[i % DPL] * __uint_as_float(ku[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d += qv[(2 * i) % DPL] * __uint_as_float(ku[i] << 16);
          d += qv[(2 * i + 1) % DPL] * __uint_as_float(ku[i] & 0xffff0000u);
        }
      }
      d = KVF32 ? smi_sum16(d) : smi_sum8(d);
      sc[u] = (c0 + u * NGRP + grp < ctx) ? d : NEG;
      lmax = fmaxf(lmax, sc[u]);
    }
    // wave max: inside a 16-lane row on the DPP path, across the four rows by readlane
    lmax = fmaxf(lmax, smi_dpp<0x128>(lmax));   // row_ror:8
    const float wm = fmaxf(fmaxf(smi_readlane(lmax, 0), smi_readlane(lmax, 16)), fmaxf(smi_readlane(lmax, 32), smi_readlane(lmax, 48)));
    // running max per WAVE (wave-uniform, no LDS, no barrier in the chunk loop); the waves' streams are brought to a
    // common scale once, in the final merge
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
      if (KVF32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i % DPL] += e * __uint_as_float(vu[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[(2 * i) % DPL] += e * __uint_as_float(vu[i] << 16);
          o[(2 * i + 1) % DPL] += e * __uint_as_float(vu[i] & 0xffff0000u);
        }
      }
    }
    m_run = mn;
    c0 += NGRP * UNR;
  } while (c0 < ctx);
  // every stream of a wave is at that wave's scale exp(-m_run): plain sums in fixed order inside the wave (step 1);
  // across the waves (step 2) each wave's sums are rescaled to the block maximum.
#pragma unroll
  for (int i = 0; i < DPL; i += 4)
    *(float4*)&so[wave][tl][dl * DPL + i] = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
  if (dl == 0) sl[wave][tl] = lrun;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes have landed
  {
    float O = 0.f, Ls = 0.f;
#pragma unroll
    for (int g = 0; g < TPW; ++g) { O += so[wave][g][lane]; Ls += sl[wave][g]; }
    pw[wave][lane] = O;
    if (lane == 0) { pl[wave] = Ls; wmax[wave] = m_run; }
  }
  __syncthreads();
  if (tid < kHeadDim) {   // step 2, across the waves
    float bm = wmax[0];
#pragma unroll
    for (int w = 1; w < kAttnWaves; ++w) bm = fmaxf(bm, wmax[w]);
    float O = 0.f, Ls = 0.f;
#pragma unroll
    for (int w = 0; w < kAttnWaves; ++w) {
      const float sc = exp2f((wmax[w] - bm) * LOG2E);   // 0 for a wave that saw no valid token (m_run = NEG)
      O += pw[w][tid] * sc; Ls += pl[w] * sc;
    }
    if (!ONE && p.nseg > 1) {   // segment partial at scale exp(-bm); k_attn_merge finishes the row
      float* pp = p.part + (((size_t)m * p.n_heads + head) * p.nseg + seg) * 66;
      pp[2 + tid] = O;
      if (tid == 0) { pp[0] = bm; pp[1] = Ls; }
      return;
    }
    uint32_t hi, mi, lo;
    split3(O / Ls, hi, mi, lo);
    if constexpr (FUSE) {   // B-operand pieces of this head's 64 values: [half][split][k8][8 bf16]
      unsigned char* q = xsl + (size_t)((tid >> 5) * 12 + ((tid >> 3) & 3)) * 16 + (tid & 7) * 2;
      *(uint16_t*)q = (uint16_t)hi;
      *(uint16_t*)(q + 64) = (uint16_t)mi;
      *(uint16_t*)(q + 128) = (uint16_t)lo;
    } else {
      const int Mx = ONE == 1 ? 1 : p.M;
      const size_t ob = xs_off(o_ktile(head, tid, p.n_heads), 0, (tid >> 3) & 3, m, Mx) + (tid & 7) * 2;
      const size_t pl2 = (size_t)4 * Mx * 16;
      *(uint16_t*)(p.xs_out + ob) = (uint16_t)hi;
      *(uint16_t*)(p.xs_out + ob + pl2) = (uint16_t)mi;
      *(uint16_t*)(p.xs_out + ob + 2 * pl2) = (uint16_t)lo;
    }
  }
  if constexpr (FUSE) __syncthreads();   // the o_proj waves take over: xsl holds this head's output
  }   // attention waves
  if constexpr (FUSE == 2) {
    __syncthreads();                          // every o_proj wave of this block has drained its partial stores (vmcnt(0) above)
    unsigned int* cnt = p.cnt + f_row * kFuseQB + f_q;
    if (tid == 0) s_last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(p.n_heads - 1);
    __syncthreads();                          // (the adding lane joins only after its add has returned)
    if (!s_last) return;                      // block-uniform
    const int H = p.NTo * 16, ncol4 = H / kFuseQB / 4;     // float4 column groups of this quarter (56 at 0.5B)
    if (tid < ncol4) {
      const int n = f_q * (H / kFuseQB) + 4 * tid, m = f_row, M = p.M;
      // the heads in order (k_gemm<RESID, NW = n_heads>'s wave order, k_gemm<PRO_FUSEDO>'s partial order), every read an sc1 load
      // (ALL reads are requested before the first is used: as a loop over the heads they were 14 dependent memory round trips --
      // write-through stores leave no copy in any L2 -- and the kernel took 17 us at 8 rows)
      uint32_t pv[kMaxOHeads][4];
#pragma unroll
      for (int hd = 0; hd < kMaxOHeads; ++hd) {
        const uint32_t* src = (const uint32_t*)(p.part_o + ((size_t)m * p.n_heads + (hd < p.n_heads ? hd : p.n_heads - 1)) * H + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[hd][r] = __hip_atomic_load(src + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      float y[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = __uint_as_float(pv[0][r]);
#pragma unroll
      for (int hd = 1; hd < kMaxOHeads; ++hd)
        if (hd < p.n_heads) {   // uniform
#pragma unroll
          for (int r = 0; r < 4; ++r) y[r] += __uint_as_float(pv[hd][r]);
        }
      // k_gemm's RESID epilogue for row m, columns n .. n + 3
      const float4 egam = *(const float4*)(p.gamma_next + n);
      float4 h = *(const float4*)(p.h + (size_t)m * H + n);
      h.x += y[0]; h.y += y[1]; h.z += y[2]; h.w += y[3];
      *(float4*)(p.h + (size_t)m * H + n) = h;
      const float ssq = (h.x * h.x + h.y * h.y) + (h.z * h.z + h.w * h.w);
      const float t[4] = {egam.x * h.x, egam.y * h.y, egam.z * h.z, egam.w * h.w};
      uint32_t hi[4], mi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split3(t[e], hi[e], mi[e], lo[e]);
      const size_t o = xs_off(n >> 5, 0, (n >> 3) & 3, m, M) + ((n >> 2) & 1) * 8;
      const size_t pl = (size_t)4 * M * 16;
      *(uint2*)(p.xs_next + o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
      *(uint2*)(p.xs_next + o + pl) = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
      *(uint2*)(p.xs_next + o + 2 * pl) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
      p.ssout[((size_t)m * p.NTo + (n >> 4)) * 4 + ((n >> 2) & 3)] = ssq;
    }
    if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  }
}

// Prompt rows (prefill of more than one chunk): one WAVE per (row, head), eight consecutive rows of one head per
// block.  A prompt row's context is short on average (half the prompt) and thousands of rows arrive together, so the
// decode kernel's shape -- eight waves and two barriers around one 256-token chunk -- is mostly idle lanes and merge
// overhead there (199 us per layer for 32 x 127 rows).  Here a wave walks its row's keys 32 (bf16 KV) at a time
// with the same per-token arithmetic, keeps its own running max, and merges its token streams through a private LDS
// slice; nothing is shared between waves, so there is no barrier and no context segmentation.  Neighbouring rows of
// a sequence read the same K/V lines out of L1/L2.
template <int KVF32>
__global__ __launch_bounds__(kAttnWaves * 64) void k_attn_pf(AttnP p) {
  constexpr int LPT = KVF32 ? 16 : 8;
  constexpr int DPL = kHeadDim / LPT;
  constexpr int TPW = 64 / LPT;
  constexpr int UNR = 4;
  constexpr float NEG = -1e30f;
  constexpr float LOG2E = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) float so[kAttnWaves][TPW][kHeadDim];
  __shared__ float sl[kAttnWaves][TPW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // natural order: the 14 heads of a row tile are neighbours; XCD x takes the x-th contiguous eighth of it (see k_attn)
  const unsigned total = gridDim.x * gridDim.y, pb = blockIdx.x + gridDim.x * blockIdx.y;
  const unsigned lb = (pb & 7u) * (total >> 3) + ((pb & 7u) < (total & 7u) ? (pb & 7u) : (total & 7u)) + (pb >> 3);
  const int head = (int)(lb % gridDim.y), m = (int)(lb / gridDim.y) * kAttnWaves + wave;
  if (m >= p.M) return;   // wave-uniform; the kernel has no barrier
  const int tl = lane / LPT, dl = lane % LPT;
  const RowDesc rd = p.rows[m];
  const int kvh = head / p.group;
  const int ctx = rd.pos + 1;
  float qv[DPL];
  {
    const float* qp = p.q + (size_t)m * p.q_dim + head * kHeadDim + dl * DPL;
#pragma unroll
    for (int i = 0; i < DPL; ++i) qv[i] = qp[i] * 0.125f;  // head_dim^-0.5, exact
  }
  float m_run = NEG, lrun = 0.f, o[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) o[i] = 0.f;
  for (int c0 = 0; c0 < ctx; c0 += TPW * UNR) {
    uint4 kr[UNR], vr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = c0 + u * TPW + tl;
      const size_t off = kv_row(p.km, rd.slot, kvh, p.n_kv, p.max_pos, t < ctx ? t : ctx - 1) * kHeadDim + dl * DPL;
      if (KVF32) {
        kr[u] = *(const uint4*)((const float*)p.kcache + off);
        vr[u] = *(const uint4*)((const float*)p.vcache + off);
      } else {
        kr[u] = *(const uint4*)((const uint16_t*)p.kcache + off);
        vr[u] = *(const uint4*)((const uint16_t*)p.vcache + off);
      }
    }
    float sc[UNR];
    float lmax = NEG;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const uint32_t ku[4] = {kr[u].x, kr[u].y, kr[u].z, kr[u].w};
      float d = 0.f;
      if (KVF32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) d += qv[i % DPL] * __uint_as_float(ku[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d += qv[(2 * i) % DPL] * __uint_as_float(ku[i] << 16);
          d += qv[(2 * i + 1) % DPL] * __uint_as_float(ku[i] & 0xffff0000u);
        }
      }
      d = KVF32 ? smi_sum16(d) : smi_sum8(d);
      sc[u] = (c0 + u * TPW + tl < ctx) ? d : NEG;
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
      if (KVF32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i % DPL] += e * __uint_as_float(vu[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[(2 * i) % DPL] += e * __uint_as_float(vu[i] << 16);
          o[(2 * i + 1) % DPL] += e * __uint_as_float(vu[i] & 0xffff0000u);
        }
      }
    }
    m_run = mn;
  }
  // the wave's token streams are all at scale exp(-m_run): plain sums in fixed order
#pragma unroll
  for (int i = 0; i < DPL; i += 4)
    *(float4*)&so[wave][tl][dl * DPL + i] = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
  if (dl == 0) sl[wave][tl] = lrun;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes have landed
  float O = 0.f, Ls = 0.f;
#pragma unroll
  for (int g = 0; g < TPW; ++g) { O += so[wave][g][lane]; Ls += sl[wave][g]; }
  uint32_t hi, mi, lo;
  split3(O / Ls, hi, mi, lo);
  const size_t ob = xs_off(o_ktile(head, lane, p.n_heads), 0, (lane >> 3) & 3, m, p.M) + (lane & 7) * 2;
  const size_t pl2 = (size_t)4 * p.M * 16;
  *(uint16_t*)(p.xs_out + ob) = (uint16_t)hi;
  *(uint16_t*)(p.xs_out + ob + pl2) = (uint16_t)mi;
  *(uint16_t*)(p.xs_out + ob + 2 * pl2) = (uint16_t)lo;
}


// ------------------------------------------------------------------------------------------
// Prefill attention on the matrix pipes (bf16 KV cache): one wave per (tile of up to 16 consecutive rows of one prompt,
// head).  k_attn_pf gives every (row, head) its own wave, which re-reads the head's K/V for every row (8 x 460-token prompts:
// 3 GB of L2 traffic and 210 us per layer, the largest prefill kernel there); here the 16 queries of a tile share each K/V load.
//   scores, transposed: S^T[key][query] = K (A operand: a key's 8 consecutive dims per lane, straight from the cache row) x
//     q^T (B operand: the query's dims, fp32 -> exact bf16 triple, three MFMAs lo/mid/hi) -- v_mfma_f32_16x16x32_bf16;
//   its accumulator layout (lane (g, query): keys 4g .. 4g+3) IS the A operand layout of v_mfma_f32_16x16x4_f32 over those
//     keys, so P = exp2((S - max) log2 e) feeds P x V in fp32 with no shuffle; V rows are read as they lie (lane (g, n): dims
//     4n .. 4n+3 of key 4g + t), output column n of tile d standing for dim 4n + d.
// Two passes over the keys (row maxima, then P and P x V: the scores are recomputed bit for bit), so no running rescale;
// causal mask key <= query position; per query the sum of P and the division happen once at the end.
// ------------------------------------------------------------------------------------------
struct PfTile { int32_t m0, n, slot, pos0; };   // rows m0 .. m0+n-1 are tokens pos0 .. pos0+n-1 of KV slot `slot`

__global__ __launch_bounds__(256) void k_attn_pf2(AttnP p, const PfTile* tiles, int ntiles) {
  constexpr float NEG = -1e30f;
  constexpr float LOG2E = 1.4426950408889634f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int task = (int)blockIdx.x * 4 + wave;
  const int ti = task / p.n_heads, head = task - ti * p.n_heads;
  if (ti >= ntiles) return;   // wave-uniform; the kernel has no barrier
  const PfTile t = tiles[ti];
  const int g = lane >> 4, j = lane & 15;
  const int kvh = head / p.group;
  const uint16_t* Kc = (const uint16_t*)p.kcache;
  const uint16_t* Vc = (const uint16_t*)p.vcache;
  // q^T operand: query j's dims s*32 + g*8 .. +8, scaled by head_dim^-0.5 (exact), as bf16 triples
  bf16x8 qh[2], qm[2], ql[2];
  {
    const float* qp = p.q + (size_t)(t.m0 + (j < t.n ? j : t.n - 1)) * p.q_dim + head * kHeadDim + g * 8;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float4 a = *(const float4*)(qp + s2 * 32), b = *(const float4*)(qp + s2 * 32 + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      uint32_t hi[8], mi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split3(v[e] * 0.125f, hi[e], mi[e], lo[e]);
      const uint4 H = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
      const uint4 Mi = make_uint4(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16), mi[4] | (mi[5] << 16), mi[6] | (mi[7] << 16));
      const uint4 Lo = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
      qh[s2] = __builtin_bit_cast(bf16x8, H); qm[s2] = __builtin_bit_cast(bf16x8, Mi); ql[s2] = __builtin_bit_cast(bf16x8, Lo);
    }
  }
  const int nkeys = t.pos0 + t.n;                 // keys 0 .. nkeys-1; query j sees keys <= pos0 + j
  const int nblk = (nkeys + 15) >> 4;
  const int qlast = t.pos0 + j;
  struct KReg { uint4 a, b; };
  auto kload = [&](int kb) -> KReg {              // A operand of block kb: this lane holds key kb*16 + j's dims g*8 .. +8 of both halves
    int key = kb * 16 + j;
    key = key < nkeys ? key : nkeys - 1;
    const uint16_t* kp = Kc + kv_row(p.km, t.slot, kvh, p.n_kv, p.max_pos, key) * kHeadDim + g * 8;
    return KReg{*(const uint4*)kp, *(const uint4*)(kp + 32)};
  };
  auto scores = [&](int kb, const KReg& kr) -> f32x4 {   // S^T block: lane (g, query j), register r: key kb*16 + 4g + r
    const bf16x8 k0 = __builtin_bit_cast(bf16x8, kr.a), k1 = __builtin_bit_cast(bf16x8, kr.b);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, ql[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, ql[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qm[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qm[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qh[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qh[1], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (kb * 16 + 4 * g + r <= qlast) ? acc[r] : NEG;
    return acc;
  };
  // ---- pass 1: the queries' maxima (the next block's K rows are requested before this block's MFMAs)
  // (two register sets used in turn, two key blocks per trip: with one set and `kc = kn` at the end of a trip hipcc copies the registers
  // there and waits vmcnt(0) -- for the block it has just requested -- in front of the trip's last MFMA: round-4 ISA)
  float mx = NEG;
  {
    KReg ka = kload(0);
    for (int kb = 0; kb < nblk; kb += 2) {
      const KReg kbr = kload(kb + 1 < nblk ? kb + 1 : kb);
      const f32x4 sa = scores(kb, ka);
      mx = fmaxf(mx, fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sa[2], sa[3])));
      ka = kload(kb + 2 < nblk ? kb + 2 : kb);
      if (kb + 1 < nblk) {
        const f32x4 sb = scores(kb + 1, kbr);
        mx = fmaxf(mx, fmaxf(fmaxf(sb[0], sb[1]), fmaxf(sb[2], sb[3])));
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  // ---- pass 2: P and P x V
  f32x4 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float lsum = 0.f;
  struct VReg { uint2 v[4]; };
  auto vload = [&](int kb) -> VReg {
    VReg r;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      int key = kb * 16 + 4 * g + tt;
      key = key < nkeys ? key : nkeys - 1;
      r.v[tt] = *(const uint2*)(Vc + kv_row(p.km, t.slot, kvh, p.n_kv, p.max_pos, key) * kHeadDim + 4 * j);
    }
    return r;
  };
  auto pv = [&](int kb, const KReg& kr, const VReg& vr) {
    const f32x4 sc = scores(kb, kr);
    float pr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pr[r] = sc[r] > 0.5f * NEG ? exp2f((sc[r] - mx) * LOG2E) : 0.f;
      lsum += pr[r];
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const float v0 = __uint_as_float(vr.v[tt].x << 16), v1 = __uint_as_float(vr.v[tt].x & 0xffff0000u);
      const float v2 = __uint_as_float(vr.v[tt].y << 16), v3 = __uint_as_float(vr.v[tt].y & 0xffff0000u);
      o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[tt], v0, o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[tt], v1, o[1], 0, 0, 0);
      o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[tt], v2, o[2], 0, 0, 0);
      o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[tt], v3, o[3], 0, 0, 0);
    }
  };
  KReg ka = kload(0);
  VReg va = vload(0);
  for (int kb = 0; kb < nblk; kb += 2) {     // sets A and B in turn (see pass 1); the key blocks are visited in the same order
    const int k1 = kb + 1 < nblk ? kb + 1 : kb, k2 = kb + 2 < nblk ? kb + 2 : kb;
    const KReg kbr = kload(k1);
    const VReg vbr = vload(k1);
    pv(kb, ka, va);
    ka = kload(k2);
    va = vload(k2);
    if (kb + 1 < nblk) pv(kb + 1, kbr, vbr);
  }
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);               // lane (any g, j): the sum for query j
  // ---- output: lane (g, n) holds O[query 4g + r][dims 4n .. 4n+3] in o[0..3][r]
  const size_t pl2 = (size_t)4 * p.M * 16;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = 4 * g + r;
    const float ls = __shfl(lsum, qi, 64);
    if (qi >= t.n) continue;
    uint32_t hi[4], mi[4], lo[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) split3(o[d][r] / ls, hi[d], mi[d], lo[d]);
    const size_t ob = xs_off(o_ktile(head, 4 * j, p.n_heads), 0, (j >> 1) & 3, t.m0 + qi, p.M) + (j & 1) * 8;
    *(uint2*)(p.xs_out + ob) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
    *(uint2*)(p.xs_out + ob + pl2) = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
    *(uint2*)(p.xs_out + ob + 2 * pl2) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
  }
}

// Combines a row's context segments in order (a row with one segment reproduces the unsplit result bit for bit:
// its scale factor is exp2(0) = 1).  One block per row, one thread per (head, dim).
__global__ __launch_bounds__(1024) void k_attn_merge(AttnP p) {
  constexpr float LOG2E = 1.4426950408889634f;
  const int m = blockIdx.x;
  const int ns = (p.rows[m].pos + 1 + kAttnSeg - 1) / kAttnSeg;
  for (int i = threadIdx.x; i < p.n_heads * kHeadDim; i += blockDim.x) {
    const int head = i / kHeadDim, d = i - head * kHeadDim;
    const float* pp = p.part + ((size_t)m * p.n_heads + head) * p.nseg * 66;
    float bm = pp[0];
    for (int s2 = 1; s2 < ns; ++s2) bm = fmaxf(bm, pp[s2 * 66]);
    float O = 0.f, Ls = 0.f;
    for (int s2 = 0; s2 < ns; ++s2) {
      const float sc = exp2f((pp[s2 * 66] - bm) * LOG2E);
      O += pp[s2 * 66 + 2 + d] * sc;
      Ls += pp[s2 * 66 + 1] * sc;
    }
    uint32_t hi, mi, lo;
    split3(O / Ls, hi, mi, lo);
    const size_t ob = xs_off(o_ktile(head, d, p.n_heads), 0, (d >> 3) & 3, m, p.M) + (d & 7) * 2;
    const size_t pl2 = (size_t)4 * p.M * 16;
    *(uint16_t*)(p.xs_out + ob) = (uint16_t)hi;
    *(uint16_t*)(p.xs_out + ob + pl2) = (uint16_t)mi;
    *(uint16_t*)(p.xs_out + ob + 2 * pl2) = (uint16_t)lo;
  }
}

// ------------------------------------------------------------------------------------------
// Token -> residual row.  One wave per row: gathers the embedding row from the tiled lm_head (tied
// weights), writes h (fp32), the first norm's operand (exact triples of gamma * h) and the row's
// sum of squares (partial slot 0; the other slots are zeroed so consumers sum a fixed count).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void embed_row(const uint16_t* Wlm, int KT, int token, int m, int M, const float* gamma,
                                          float* h, unsigned char* xs, float* sspart, int npart, int lane) {
  const int K = KT * 32, pieces = KT * 4;
  float ss = 0.f;
  for (int pc = lane; pc < pieces; pc += 64) {      // pc = k / 8
    const size_t tile = (size_t)(token >> 4) * KT + (pc >> 2);
    const uint4 v = *(const uint4*)(Wlm + tile * 512 + ((pc & 3) * 16 + (token & 15)) * 8);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
    float x[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(u[e] << 16); x[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
    float* dst = h + (size_t)m * K + pc * 8;
    *(float4*)dst = make_float4(x[0], x[1], x[2], x[3]);
    *(float4*)(dst + 4) = make_float4(x[4], x[5], x[6], x[7]);
    const float4 g0 = *(const float4*)(gamma + pc * 8), g1 = *(const float4*)(gamma + pc * 8 + 4);
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    uint32_t hi[8], mi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ss += x[e] * x[e]; split3(g[e] * x[e], hi[e], mi[e], lo[e]); }
    const size_t o = xs_off(pc >> 2, 0, pc & 3, m, M);
    const size_t pl = (size_t)4 * M * 16;
    *(uint4*)(xs + o) = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
    *(uint4*)(xs + o + pl) = make_uint4(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16), mi[4] | (mi[5] << 16), mi[6] | (mi[7] << 16));
    *(uint4*)(xs + o + 2 * pl) = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
  }
  ss = smi_wave_sum(ss);
  for (int i = lane; i < npart; i += 64) sspart[(size_t)m * npart + i] = i == 0 ? ss : 0.f;
}

// Drops the rows of the retired KV slots from the live row list, in place and in order (one block; B <= 64 rows).
__global__ __launch_bounds__(64) void k_rows_drop(RowDesc* rows, int B, unsigned long long drop_slots) {
  const int i = threadIdx.x;
  RowDesc rd = RowDesc{0, 0, 0, 0};
  bool keep = false;
  if (i < B) { rd = rows[i]; keep = !((drop_slots >> rd.slot) & 1ull); }
  const unsigned long long m = __ballot(keep);
  const int dst = __popcll(m & ((1ull << i) - 1ull));
  __syncthreads();   // (one wave: every row was read before any is overwritten)
  if (keep) rows[dst] = rd;
  else if (i < 64) { /* nothing: the tail beyond the new count is never read */ }
}

__global__ __launch_bounds__(256) void k_embed(const uint16_t* Wlm, int KT, const RowDesc* rows, int M, const float* gamma,
                                               float* h, unsigned char* xs, float* sspart, int npart, int exact) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
    if (exact) embed_row_x((const float*)Wlm, KT, rows[m].token, m, M, gamma, h, xs, sspart, npart, lane);   // exact-weights arena: fp32 [vocab][K]
    else embed_row(Wlm, KT, rows[m].token, m, M, gamma, h, xs, sspart, npart, lane);
  }
}

#ifdef SMI_DIAG
// Tests (smi_llm_debug_layer): caller-given fp32 residual rows -> the state a layer starts from (h, the first norm's operand
// triples, the row's sum of squares in partial slot 0), exactly as embed_row leaves it for an embedding row.
__global__ __launch_bounds__(256) void k_load_hidden(const float* src, int KT, int M, const float* gamma, float* h, unsigned char* xs,
                                                     float* sspart, int npart) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = KT * 32;
  for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
    float ss = 0.f;
    for (int pc = lane; pc < KT * 4; pc += 64) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = src[(size_t)m * K + pc * 8 + e];
      float* dst = h + (size_t)m * K + pc * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e] = x[e];
      uint32_t hi[8], mi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { ss += x[e] * x[e]; split3(gamma[pc * 8 + e] * x[e], hi[e], mi[e], lo[e]); }
      const size_t o = xs_off(pc >> 2, 0, pc & 3, m, M);
      const size_t pl = (size_t)4 * M * 16;
      *(uint4*)(xs + o) = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
      *(uint4*)(xs + o + pl) = make_uint4(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16), mi[4] | (mi[5] << 16), mi[6] | (mi[7] << 16));
      *(uint4*)(xs + o + 2 * pl) = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
    }
    ss = smi_wave_sum(ss);
    for (int i = lane; i < npart; i += 64) sspart[(size_t)m * npart + i] = i == 0 ? ss : 0.f;
  }
}
#endif

// ------------------------------------------------------------------------------------------
// Sampling (the reference's default: do_sample=True, top_k=50, top_p=0.95, temperature=0.8 at
// cli/SparkTTS.py:166-168,197-204).  Restates the HF warper chain: logits / T -> keep top k ->
// nucleus (drop the low tail whose cumulative probability <= 1 - top_p, keep >= 1) -> softmax ->
// one multinomial draw.  The draw uses a counter-based Philox4x32-10 stream keyed by
// (seed; the sequence's own token index, its admission number), so a run is reproducible but not
// bit-identical to torch.multinomial.
// ------------------------------------------------------------------------------------------
struct SampleP {
  const float* logits;  // [M][V]
  int V, top_k;
  float inv_temp, top_p;
  const Ctl* ctl;        // seed and the per-slot admission numbers
  const RowDesc* rows;   // the live rows: (slot, ..., flags = tokens this sequence has emitted so far)
  int* tok;             // [kMaxRows] sampled token per row
  const float* pval;    // [M][nblk] the lm_head blocks' best logits (a bound for the top-k threshold) or null
  int nblk;
  // candidates at or above the bound, collected by k_sample_scan: [kMaxRows][kCandCap] values / indices, [kMaxRows] counts
  float* cand_v; int* cand_i; unsigned int* cand_n;
};

__device__ __forceinline__ uint32_t sortable(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

__device__ inline uint32_t philox_u32(unsigned long long seed, uint32_t c0, uint32_t c1) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t x0 = c0, x1 = c1, x2 = 0x5eed5eedu, x3 = 0;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * x0, p1 = (unsigned long long)0xCD9E8D57u * x2;
    const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1;
    const uint32_t y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
    x0 = y0; x1 = y1; x2 = y2; x3 = y3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return x0;
}

constexpr int kSampleCap = 256;  // top_k upper bound
constexpr int kCandCap = 1024;   // candidates at or above the threshold taken from the lm_head blocks' maxima

// Exact key of the k-th largest logit by 4 passes of 8-bit radix selection (fallback path: one block walks
// the whole row 4 times, with LDS-atomic histograms).
__device__ uint32_t radix_kth_key(const float* lg, int V, int top_k, unsigned int* hist, unsigned int* s_prefix, unsigned int* s_need) {
  const int tid = threadIdx.x;
  if (tid == 0) { *s_prefix = 0; *s_need = (unsigned)top_k; }
  __syncthreads();
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = *s_prefix;
    const uint32_t mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < V; i += 1024) {
      const uint32_t k = sortable(lg[i]);
      if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned need = *s_need;
      int b = 255;
      for (; b > 0; --b) {
        if (hist[b] >= need) break;
        need -= hist[b];
      }
      *s_need = need;                       // rank inside bin b
      *s_prefix = prefix | ((uint32_t)b << shift);
    }
    __syncthreads();
  }
  return *s_prefix;
}

constexpr int kScanBlocks = 16;  // blocks per row in k_sample_scan

// The top_k-th largest of the lm_head blocks' maxima is a lower bound of the top_k-th largest logit (those maxima are
// top_k distinct logits at or above it), so ONE pass over the row collects a short candidate list that contains the
// whole top-k.  kScanBlocks blocks per row share that pass (one CU pulls a 664 KB row at ~150 GB/s: 25 us for a
// single block); each finds the bound itself and appends its candidates to the row's global list.  The list order
// depends on the race, the result does not: k_sample rank-sorts the list by (value, index).
__global__ __launch_bounds__(256) void k_sample_scan(SampleP p) {
  __shared__ float s_thr;
  const int m = blockIdx.y, tid = threadIdx.x;
  if (!(p.pval && p.nblk <= kCandCap && p.nblk >= p.top_k)) return;   // no usable bound: k_sample selects by radix
  // the bound, by ONE wave: each lane holds up to 16 of the (<= 1024) maxima as order-preserving keys and the answer is
  // built bit by bit from ballots (K = max{x : #{keys >= x} >= top_k}) -- no LDS, no barrier
  if (tid < 64) {
    uint32_t key[kCandCap / 64];
#pragma unroll
    for (int i = 0; i < kCandCap / 64; ++i) {
      const int j = tid + 64 * i;
      key[i] = j < p.nblk ? sortable(p.pval[(size_t)m * p.nblk + j]) : 0u;   // 0 sorts below every float
    }
    uint32_t K = 0;
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t t = K | (1u << bit);
      int c = 0;
#pragma unroll
      for (int i = 0; i < kCandCap / 64; ++i)
        if (i * 64 < p.nblk) c += __popcll(__ballot(key[i] >= t));   // uniform: slots beyond nblk hold no key
      if (c >= p.top_k) K = t;   // wave-uniform
    }
    if (tid == 0) s_thr = __uint_as_float((K & 0x80000000u) ? (K ^ 0x80000000u) : ~K);
  }
  __syncthreads();
  const float thr = s_thr;
  const float* lg = p.logits + (size_t)m * p.V;
  auto take = [&](float v, int idx) {
    if (v >= thr) {
      const unsigned pos = atomicAdd(&p.cand_n[m], 1u);
      if (pos < (unsigned)kCandCap) { p.cand_v[(size_t)m * kCandCap + pos] = v; p.cand_i[(size_t)m * kCandCap + pos] = idx; }
    }
  };
  if ((p.V & 3) == 0) {
    const float4* l4 = (const float4*)lg;
    const int n4 = p.V / 4, per = (n4 + kScanBlocks - 1) / kScanBlocks;
    const int lo = (int)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    constexpr int UL = 4;   // loads in flight per thread
    for (int base = lo; base < hi; base += 256 * UL) {
      float4 v[UL];
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        const int i = base + u * 256 + tid;
        v[u] = i < hi ? l4[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      }
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        const int i = base + u * 256 + tid;
        take(v[u].x, 4 * i); take(v[u].y, 4 * i + 1); take(v[u].z, 4 * i + 2); take(v[u].w, 4 * i + 3);
      }
    }
  } else {
    const int per = (p.V + kScanBlocks - 1) / kScanBlocks;
    const int lo = (int)blockIdx.x * per, hi = lo + per < p.V ? lo + per : p.V;
    for (int i = lo + tid; i < hi; i += 256) take(lg[i], i);
  }
}

// One block per row: takes the candidates k_sample_scan collected (without a usable bound, or with more than kCandCap of
// them, it selects the top_k-th key exactly by radix and walks the row itself); the candidates are rank-sorted exactly
// (value descending, index ascending on ties), then temperature -> top-k -> top-p -> multinomial.
__global__ __launch_bounds__(1024) void k_sample(SampleP p) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned int s_prefix, s_need, s_cnt;
  __shared__ int s_k;
  __shared__ float s_thr;
  __shared__ float cv[kCandCap], sv[kSampleCap];
  __shared__ int ci[kCandCap], si[kSampleCap];
  const int m = blockIdx.x, tid = threadIdx.x;
  const float* lg = p.logits + (size_t)m * p.V;
  if (tid == 0) { s_cnt = 0; s_thr = -INFINITY; s_k = 0; }
  __syncthreads();
  const bool bound_ok = p.pval && p.nblk <= kCandCap && p.nblk >= p.top_k;   // the condition k_sample_scan ran under
  if (bound_ok) {
    if (tid == 0) { s_cnt = p.cand_n[m]; p.cand_n[m] = 0; s_thr = 0.f; }   // the counter is left at zero for the next step
    __syncthreads();
    const unsigned ng = s_cnt;
    if (ng <= (unsigned)kCandCap && tid < (int)ng) {
      cv[tid] = p.cand_v[(size_t)m * kCandCap + tid];
      ci[tid] = p.cand_i[(size_t)m * kCandCap + tid];
    }
    __syncthreads();
  }
  const float thr0 = s_thr;
  if (thr0 == -INFINITY || s_cnt > (unsigned)kCandCap) {
    // no usable bound (or a pathological row with > 1024 logits above it): exact radix selection of the k-th key
    __syncthreads();
    const uint32_t thr = radix_kth_key(lg, p.V, p.top_k, hist, &s_prefix, &s_need);
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int i = tid; i < p.V; i += 1024) {
      const float v = lg[i];
      if (sortable(v) >= thr) {
        const unsigned pos = atomicAdd(&s_cnt, 1u);
        if (pos < kCandCap) { cv[pos] = v; ci[pos] = i; }
      }
    }
    __syncthreads();
  }
  const int n = s_cnt < (unsigned)kCandCap ? (int)s_cnt : kCandCap;
  // rank sort: descending value, ascending index on ties; only the first top_k ranks are kept
  if (tid < n) {
    const float v = cv[tid];
    const int ix = ci[tid];
    int r = 0;
    for (int j = 0; j < n; ++j) r += (cv[j] > v) || (cv[j] == v && ci[j] < ix);
    if (r < kSampleCap) { sv[r] = v; si[r] = ix; }
  }
  __syncthreads();
  // TopKLogitsWarper removes what is strictly BELOW the k-th largest value (logits_process.py: `scores < topk[..., -1]`), so
  // every logit that ties with the k-th stays: the kept set is the sorted prefix of ranks < top_k plus the ranks behind it
  // that hold the same value (up to the kSampleCap ranks this kernel sorts)
  {
    const int nn = n < kSampleCap ? n : kSampleCap, kb = n < p.top_k ? n : p.top_k;
    if (tid < nn && (tid < kb || sv[tid] == sv[kb - 1])) atomicAdd(&s_k, 1);
  }
  __syncthreads();
  const int ktop = s_k;
  {   // softmax numerators of the top-k of logits / T (fp32, like the warpers), one per thread
    const int k = ktop;
    float e = 0.f;
    if (tid < k) e = expf(sv[tid] * p.inv_temp - sv[0] * p.inv_temp);
    __syncthreads();   // cv (the candidates) has been read by the rank sort
    if (tid < k) cv[tid] = e;
    __syncthreads();
  }
  // Nucleus cut and multinomial draw on ONE wave, four consecutive ranks per lane (top_k <= kSampleCap = 256): sums, suffix
  // sums and prefix sums come from shuffles instead of four serial loops with a division each on one thread (15 us).
  if (tid < 64) {
    static_assert(kSampleCap == 256, "four ranks per lane");
    const int k = ktop;
    const int i0 = tid * 4;
    float c[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = i0 + e < k ? cv[i0 + e] : 0.f;
    const float sum = smi_wave_sum((c[0] + c[1]) + (c[2] + c[3]));
    // token i is dropped when the probability mass from rank i on, sum_{j >= i} p_j, is <= 1 - top_p
    float sfx[4];
    sfx[3] = c[3] / sum;
    sfx[2] = c[2] / sum + sfx[3];
    sfx[1] = c[1] / sum + sfx[2];
    sfx[0] = c[0] / sum + sfx[1];
    float t = sfx[0];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float v = __shfl_down(t, d, 64);
      if (tid + d < 64) t += v;
    }
    float above = __shfl_down(t, 1, 64);   // mass of the lanes after this one
    if (tid == 63) above = 0.f;
    int best = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i0 + e >= 1 && i0 + e < k && sfx[e] + above > 1.0f - p.top_p) best = i0 + e;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_xor(best, d, 64); best = v > best ? v : best; }
    const int keep = best + 1;   // rank 0 always stays
    // inverse-CDF draw over the kept ranks
    float pre[4];
    pre[0] = i0 < keep ? c[0] : 0.f;
    pre[1] = pre[0] + (i0 + 1 < keep ? c[1] : 0.f);
    pre[2] = pre[1] + (i0 + 2 < keep ? c[2] : 0.f);
    pre[3] = pre[2] + (i0 + 3 < keep ? c[3] : 0.f);
    float q = pre[3];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float v = __shfl_up(q, d, 64);
      if (tid >= d) q += v;
    }
    const float ksum = __shfl(q, 63, 64);
    float before = __shfl_up(q, 1, 64);
    if (tid == 0) before = 0.f;
    // one stream per SEQUENCE: (its admission number, its own token index) -- a request's draws do not depend on which
    // row it occupies, on what else is live or on when its neighbours were admitted
    const RowDesc rd = p.rows[m];
    const uint32_t r = philox_u32(p.ctl->seed, (uint32_t)rd.flags, (uint32_t)p.ctl->seqid[rd.slot]);
    const float u = (float)(r >> 8) * (1.0f / 16777216.0f) * ksum;
    int pick = keep - 1;
#pragma unroll
    for (int e = 3; e >= 0; --e)
      if (i0 + e < keep && u < before + pre[e]) pick = i0 + e;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_xor(pick, d, 64); pick = v < pick ? v : pick; }
    if (tid == 0) p.tok[m] = si[pick];
  }
}

struct FinP {
  const int* tok;       // non-null: tokens already chosen by k_sample
  const float* pval;
  const int* pidx;
  int nblk, M, KT, V;
  RowDesc* rows;
  int64_t* hist;      // [max_steps][kMaxRows]
  int32_t* count;     // [kMaxRows] tokens counted per sequence
  int32_t* finished;  // [kMaxRows]
  int32_t* step;      // [1]
  const Ctl* ctl;     // eos ids
  const uint16_t* Wlm;
  float* h;
  int max_steps;
  const float* gamma0;   // first layer's input norm weight
  unsigned char* xs;     // first layer's operand
  float* sspart; int npart;
  int exact;             // exact-weights arena: Wlm is fp32 [vocab][K]
};

// One block per row: arg-max over the lm_head blocks' partials (or the sampled token), per-row
// bookkeeping, and the next step's residual row / first-norm operand.  The per-row step index lives
// in the row descriptor (flags), so blocks never race on a shared counter; block 0 publishes it.
__global__ __launch_bounds__(256) void k_finalize(FinP p) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int tok_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = blockIdx.x;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  if (!p.tok) {
    constexpr int NP = 16;   // all partial loads in flight together (one memory round trip)
    float pv[NP];
    int pi[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int i = tid + 256 * j;
      const bool in = i < p.nblk;
      pv[j] = in ? p.pval[(size_t)m * p.nblk + i] : -INFINITY;
      pi[j] = in ? p.pidx[(size_t)m * p.nblk + i] : 0x7fffffff;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j)
      if (pv[j] > bv || (pv[j] == bv && pi[j] < bi)) { bv = pv[j]; bi = pi[j]; }
    for (int i = tid + 256 * NP; i < p.nblk; i += 256) {
      const float v = p.pval[(size_t)m * p.nblk + i];
      const int ix = p.pidx[(size_t)m * p.nblk + i];
      if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
    if (p.tok) bi = p.tok[m];
    if ((unsigned)bi >= (unsigned)p.V) bi = 0;   // no finite logit (NaN weights / inputs): a defined token instead of an out-of-range embedding row
    tok_s = bi;
    RowDesc rd = p.rows[m];
    const int step = rd.flags;                 // tokens this row has emitted so far
    const int sl = rd.slot;                    // history / counters live per KV slot (== m outside sessions)
    if (step < p.max_steps) p.hist[(size_t)step * kMaxRows + sl] = bi;
    if (!p.finished[sl]) {
      p.count[sl] = step + 1;
      bool stop = false;   // HF generate(): any id of generation_config.eos_token_id ends the sequence
#pragma unroll
      for (int e = 0; e < SMI_MAX_EOS; ++e) stop |= e < p.ctl->n_eos && (long long)bi == p.ctl->eos[e];
      if (stop) p.finished[sl] = 1;
    }
    rd.token = bi;
    rd.pos += 1;
    rd.flags = step + 1;
    p.rows[m] = rd;
    if (m == 0) *p.step = step + 1;
  }
  __syncthreads();
  if (wave == 0) {
    if (p.exact) embed_row_x((const float*)p.Wlm, p.KT, tok_s, m, p.M, p.gamma0, p.h, p.xs, p.sspart, p.npart, lane);
    else embed_row(p.Wlm, p.KT, tok_s, m, p.M, p.gamma0, p.h, p.xs, p.sspart, p.npart, lane);
  }
}

}  // namespace
#ifdef SMI_DIAG   // the one-row decode engine (built, bit-exact, slower than the launch path: DESIGN 3.7) lives in the diagnostics build only
#include "smi_eng.h"        // all layers of a step in one persistent launch
#include "smi_eng_host.h"
#endif
namespace {

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct Layout {
  size_t off[SMI_LLM_NUM_SECTIONS];   // per-layer sections: offset inside a layer block
  size_t bytes[SMI_LLM_NUM_SECTIONS];
  size_t layer_stride, layers_base, total;
  int vpad;
};

bool cfg_ok(const smi_llm_cfg* c) {
  if (!c) return false;
  if (c->head_dim != kHeadDim || c->hidden_size <= 0 || c->hidden_size % 32) return false;
  if (c->intermediate_size <= 0 || c->intermediate_size % 32) return false;
  if (c->num_heads <= 0 || c->num_kv_heads <= 0 || c->num_heads % c->num_kv_heads) return false;
  if (c->num_layers <= 0 || c->vocab_size <= 0) return false;
  if (c->max_slots < 1 || c->max_slots > kMaxRows || c->max_positions < 2) return false;
  if ((c->num_heads * c->head_dim) % 32) return false;
  if (c->kv_dtype != 0 && c->kv_dtype != 1) return false;
  if (c->wd_plain != 0 && c->wd_plain != 1) return false;
  if (c->weights_exact != 0 && c->weights_exact != 1) return false;
  if (c->kv_page_tokens != 0) {   // paged KV: a power of two in 16..1024 that divides max_positions, and a pool of at least one page
    const int t = c->kv_page_tokens;
    if (t < 16 || t > 1024 || (t & (t - 1)) || c->max_positions % t || c->kv_pages < 1) return false;
  }
  return true;
}

Layout make_layout(const smi_llm_cfg* c) {
  Layout L;
  memset(&L, 0, sizeof(L));
  const size_t H = c->hidden_size, Q = (size_t)c->num_heads * c->head_dim, KV = (size_t)c->num_kv_heads * c->head_dim;
  const size_t I = c->intermediate_size;
  L.vpad = (c->vocab_size + 15) / 16 * 16;
  L.bytes[SMI_LLM_LN1] = H * 4;
  const size_t wb = c->weights_exact ? 4 : 2;   // exact-weights mode: fp32 [N][K] row-major instead of bf16 tiles
  L.bytes[SMI_LLM_WQKV] = (Q + 2 * KV) * H * wb;
  L.bytes[SMI_LLM_BQKV] = (Q + 2 * KV) * 4;
  L.bytes[SMI_LLM_WO] = H * Q * wb;
  L.bytes[SMI_LLM_LN2] = H * 4;
  L.bytes[SMI_LLM_WGU] = 2 * I * H * wb;
  L.bytes[SMI_LLM_WD] = H * I * wb;
  L.bytes[SMI_LLM_FINAL_NORM] = H * 4;
  L.bytes[SMI_LLM_LM_HEAD] = (size_t)L.vpad * H * wb;
  L.bytes[SMI_LLM_ROPE] = (size_t)c->max_positions * (kHeadDim / 2) * 8;
  L.bytes[SMI_LLM_TAG] = sizeof(smi_llm_arena_tag);
  size_t o = 0;
  for (int s = SMI_LLM_LN1; s <= SMI_LLM_WD; ++s) { L.off[s] = o; o += smi_align_up(L.bytes[s], 256); }
  L.layer_stride = o;
  size_t g = 0;
  for (int s = SMI_LLM_FINAL_NORM; s <= SMI_LLM_TAG; ++s) { L.off[s] = g; g += smi_align_up(L.bytes[s], 256); }
  L.layers_base = g;
  L.total = g + L.layer_stride * c->num_layers;
  return L;
}

}  // namespace

struct smi_llm {
  smi_llm_cfg cfg;
  Layout lay;
  const unsigned char* arena;
  int H, Q, KV, I, KTh, KTq, KTi, NTqkv, NTh, NTgu, NTlm;
  // device scratch
  float *h, *qbuf;
  unsigned char *xs_h, *xs_attn, *xs_act;   // GEMM operands as exact bf16 triples ([K/32][3][4][M][16 B])
  float* sspart;       // [kMaxRows][NTh * 4] partial sums of squares of h (RMSNorm), one per 4 columns
  // prefill workspace for up to big_rows rows at once (allocated at the first multi-chunk prefill)
  float *bh, *bq; unsigned char *bxs_h, *bxs_attn, *bxs_act; float* bss; int big_rows;
  float4* pslab; size_t pslab_bytes;   // k_pgemm split-K partial sums (prompt rows passes of up to kPgSplitRows rows)
  int pf_long;          // prompt rows (len - 1) from which a sequence's prompt runs through the prefill-GEMM family (default 65)
  int pg_split_rows;    // passes of up to this many rows take the few-row prefill GEMM shapes / split-K (kPgSplitRows; SPARKMI_PG_SPLIT_ROWS: A/B, same bits)
  int pg_forced;        // diagnostics: SPARKMI_PGEMM_MIN_* set -- kernels picked per launch by the pass's row count, as in round 3
  float *part_o, *h2;  // fused o_proj (one row): per-head partials [heads][H]; h + o_proj [H]
  float4* dpart;       // chain-split down_proj (k_downC): [16 chains][NTh][4 m-tiles][64] chain sums
  int dc_min;          // rows from which down_proj runs chain-split (default 7; SPARKMI_DC_MIN in the diagnostics build)
  int fuse_o;          // config allows the fused o_proj (SPARKMI_NO_FUSE_O=1 turns it off)
  int fuse1_last;      // diagnostics (SPARKMI_FUSE1_LAST=1): ONE live row also takes the last-arriver form instead of per-head partials summed by every gate_up block (A/B for DESIGN 3.9)
  int fuse2_rows;      // rows up to which 2+ live rows take the fused attention + o_proj with the last-arriver head sum (diagnostics: SPARKMI_FUSE2_ROWS; default 0 = off: measured slower)
  unsigned int* fuse_cnt;   // [kFuse2Max][kFuseQB] arrival counters of that kernel (zero between launches)
  RowDesc* rows;       // live decode rows [kMaxRows]
  RowDesc* plan;       // prefill plan
  PfTile* pf_tiles;     // prefill attention tiles of the row group in flight (k_attn_pf2)
  size_t pf_tiles_cap;
  int pf_ntiles;
  std::vector<PfTile> host_tiles;
  int attn_pf2;         // bf16 KV: prefill attention on the matrix pipes (SPARKMI_ATTN_PF2=0: one wave per (row, head))
  size_t plan_cap;     // rows
  float* pval; int* pidx; int lm_blocks, lm_cap;
  int64_t* hist; int32_t *count, *finished, *step;
  void *kcache, *vcache; size_t kv_layer_elems;
  int B; int started;
  // paged KV cache (cfg.kv_page_tokens > 0): page table [max_slots][ppslot] on the device, mirrored on the host
  int paged, pshift, ppslot;
  int32_t* ptab;
  std::vector<int32_t> hptab;
  std::vector<int32_t> free_pages;
  int slot_pages[kMaxRows];     // pages a slot holds
  int plen[kMaxRows];           // plain (non-session) generation: prompt length per slot
  Ctl hctl; Ctl* ctl;   // host copy / device block of the generation controls
  int admit_seq;        // sequences admitted so far in this generation / session (sampler stream ids)
  // continuous batching (smi_llm_session_*): live rows map to arbitrary KV slots
  int session, identity_slots;
  std::vector<int> live_order;   // session: the KV slot of every live row, in row order (what the device row list holds)
  int attn_seg;                        // context segments per (head, row) of the attention launches being issued (1 = unsplit)
  float* apart; size_t apart_floats;   // segment partials [rows][heads][attn_seg][66]
  int slot_busy[kMaxRows], slot_len[kMaxRows];      // host: slot in use; prompt length + tokens emitted (cache positions used)
  int max_len, steps_launched;  // host-side bound on cache positions in use
  // sampling state (smi_llm_set_sampling)
  int do_sample, top_k; float temperature, top_p; unsigned long long seed;
  float* logits; int* tok;
  float* cand_v; int* cand_i; unsigned int* cand_n;   // sampler candidate lists
  unsigned long long* stamps; int stamps_on;
  int max_steps;
  int tune[4];   // SPARKMI_TUNE block-shape selectors (diagnostics)
  int graph_steps;      // decode steps captured per graph replay (SPARKMI_GRAPH_STEPS, default 8; 1: one)
  int pf_inline;        // gate_up's work blocks touch the next layer's QKV weights (SPARKMI_PF_INLINE=0: off)
  int pf_qkv_eighths;   // SPARKMI_PF_QKV: how much of gate_up QKV's helpers prefetch, in eighths (default 2)
  int prefetch_mask, prefetch_rows;  // same-XCD L2 prefetch by helper blocks: one bit per producer kernel, up to this many live rows (smi_llm_create)
  int tune2;     // SPARKMI_TUNE2 bit mask (diagnostics)
  int pgemm_min_rows;   // prompt rows from which the prefill GEMM replaces row-grouped decode GEMMs (SPARKMI_PGEMM_MIN_ROWS)
  int pg_min[4];        // ... per kernel (QKV, o_proj, gate_up, down; SPARKMI_PGEMM_MIN_QKV / _O / _GU / _D override the common value)
  int wd_parts;         // W_down tiles are stored row-part-major (smi_llm_cfg.wd_plain == 0; include/sparkmi.h)
  int exact;            // smi_llm_cfg.weights_exact: fp32 matrices, every GEMM on k_gemm_x (verification mode)
  int gu1_lo;           // rows from which (up to 16) gate_up runs the one-batch, three-tile shape with one m-tile (SPARKMI_GU1_LO; default 4)
  int gu1_rows;         // rows up to which gate_up runs its one-batch, three-tile shape (SPARKMI_GU1_ROWS; default 32)
  hipGraphExec_t graph; int graph_B, graph_seg, graph_ident;   // the step graph in use (owned by graph_cache)
  // every exec remembers the stream it last ran on: a caller may alternate streams, and an exec is destroyed only after THAT
  // stream has drained (graphs_flush)
  std::map<hipGraphExec_t, hipStream_t> graph_last;
  // One captured decode step per (row count, context segments, slots-are-rows): in-flight batching changes the row count at
  // every admission / retirement, and re-capturing the ~100-node step each time cost more than the steps saved.  Everything
  // else a step reads is device data (row descriptors, stop ids, seed) or fixed at create; the sampler's parameters and the
  // attention-partials buffer are kernel arguments, so a change of either empties the cache (graphs_flush).
  std::map<uint32_t, hipGraphExec_t> graph_cache;
  hipEvent_t ev0, ev1;
  // host staging
  std::vector<RowDesc> host_rows;
#ifdef SMI_DIAG
  EngState eng;         // one-row decode engine (smi_eng.h); eng.enabled = 0: the launch path everywhere
  int eng_on;           // runtime switch (smi_llm_set_engine; SPARKMI_ENGINE=0 at create)
#endif
};

namespace {

// ---- paged KV cache: host-side page allocator.  A slot's row of the table lists the pages of its positions in order.
void pages_release(smi_llm* L, int slot) {
  for (int i = 0; i < L->slot_pages[slot]; ++i) L->free_pages.push_back(L->hptab[(size_t)slot * L->ppslot + i]);
  L->slot_pages[slot] = 0;
}
// Makes every listed slot hold pages for `tokens[i]` positions; all or nothing (SMI_ENOMEM when the pool is short).
int pages_ensure(smi_llm* L, const int* slots, const int* tokens, int n, hipStream_t st) {
  if (!L->paged) return SMI_OK;
  int extra = 0;
  for (int i = 0; i < n; ++i) {
    const int need = (tokens[i] + (1 << L->pshift) - 1) >> L->pshift;
    if (need > L->ppslot) { smi_set_error("KV pages: %d tokens exceed max_positions", tokens[i]); return SMI_EINVAL; }
    extra += need > L->slot_pages[slots[i]] ? need - L->slot_pages[slots[i]] : 0;
  }
  if (extra == 0) return SMI_OK;
  if (extra > (int)L->free_pages.size()) {
    smi_set_error("KV page pool exhausted: %d more pages of %d tokens needed, %zu of %d free (retire a sequence or enlarge kv_pages)",
                  extra, 1 << L->pshift, L->free_pages.size(), L->cfg.kv_pages);
    return SMI_ENOMEM;
  }
  for (int i = 0; i < n; ++i) {
    const int sl = slots[i], need = (tokens[i] + (1 << L->pshift) - 1) >> L->pshift;
    while (L->slot_pages[sl] < need) {
      L->hptab[(size_t)sl * L->ppslot + L->slot_pages[sl]++] = L->free_pages.back();
      L->free_pages.pop_back();
    }
  }
  SMI_HIP(hipMemcpyAsync(L->ptab, L->hptab.data(), L->hptab.size() * 4, hipMemcpyHostToDevice, st));
  return SMI_OK;
}
KvMap kv_map(const smi_llm* L) { return KvMap{L->paged ? L->ptab : nullptr, L->pshift, L->ppslot}; }
void graphs_flush(smi_llm* L) {
  // an exec is never destroyed while a launch of it may still be running: the stream each exec last ran on drains first
  // (one synchronise per distinct stream)
  std::vector<hipStream_t> drained;
  for (auto& gl : L->graph_last) {
    bool seen = false;
    for (hipStream_t d : drained) seen |= d == gl.second;
    if (!seen) { (void)hipStreamSynchronize(gl.second); drained.push_back(gl.second); }
  }
  L->graph_last.clear();
  for (auto& kv : L->graph_cache) if (kv.second) (void)hipGraphExecDestroy(kv.second);
  L->graph_cache.clear();
  L->graph = nullptr;
}

template <int MT, int NTB, int NW, int U, int WB, int PRO, int EPI, int H = 1, int OCC = 1>
int launch_gemm_kv(const smi_llm* L, GemmP p, hipStream_t st) {
  const int work = (p.NT + NTB - 1) / NTB * H;
  p.work_blocks = work;
  p.stamps = L->stamps_on ? L->stamps : nullptr;
  if constexpr (MT == 1 && NTB == 1 && NW == 16 && H == 4 && PRO == PRO_PLAIN && EPI == EPI_RESID) {
    // one row: four chains per wave, full load instructions (k_down1; same bits).  SPARKMI_TUNE2 bit 1048576 keeps k_gemm (A/B)
    if (p.M == 1 && !p.stamps && p.KT <= 160 && !(L->tune2 & 1048576)) {
      const int helpers = (L->prefetch_mask & 4) && p.pf.base && work < 232 ? ((L->tune2 & 4194304) ? 224 : (256 - work) / 8 * 8) : 0;   // bit 4194304: as many helper blocks as work blocks (they share the CUs)
      if (p.KT <= 32) hipLaunchKernelGGL(k_down1<2>, dim3(work + helpers), dim3(256), 0, st, p);
      else if (p.KT <= 96) hipLaunchKernelGGL(k_down1<6>, dim3(work + helpers), dim3(256), 0, st, p);
      else hipLaunchKernelGGL(k_down1<10>, dim3(work + helpers), dim3(256), 0, st, p);
      SMI_LAUNCH_CHECK();
      return SMI_OK;
    }
    if (p.M >= 2 && p.M <= 8 && !p.stamps && p.KT <= 160 && !(L->tune2 & 2097152)) {   // SPARKMI_TUNE2 bit 2097152 keeps k_gemm (A/B)
      const dim3 g(work), b(256);
      if (p.M <= 4) {
        if (p.KT <= 32) hipLaunchKernelGGL((k_downS<2, 1>), g, b, 0, st, p);
        else if (p.KT <= 96) hipLaunchKernelGGL((k_downS<6, 1>), g, b, 0, st, p);
        else hipLaunchKernelGGL((k_downS<10, 1>), g, b, 0, st, p);
      } else {
        if (p.KT <= 32) hipLaunchKernelGGL((k_downS<2, 2>), g, b, 0, st, p);
        else if (p.KT <= 96) hipLaunchKernelGGL((k_downS<6, 2>), g, b, 0, st, p);
        else hipLaunchKernelGGL((k_downS<10, 2>), g, b, 0, st, p);
      }
      SMI_LAUNCH_CHECK();
      return SMI_OK;
    }
  }
  // idle CUs warm the L2 of their own XCD for a later kernel (decode with few rows only)
  const int helpers = ((L->prefetch_mask & (EPI == EPI_QKV ? 1 : 4)) && p.pf.base && p.M <= L->prefetch_rows && work < 232) ? (256 - work) / 8 * 8 : 0;
  size_t lds = (size_t)NW * NTB * MT * 1024 + 32 * 4 + NTB * 32 * 8;
  p.ldsb = 0; p.lt_shift = 0;
  if (MT == 1 && p.M <= 5) {
    const int tw = (p.KT + NW - 1) / NW;               // k tiles per wave
    const size_t per_wave = (size_t)tw * 12 * p.M * 16;
    // (blocks of 14+ waves are alone on their CU anyway: down_proj's ten k tiles per wave then fit up to 3 rows -- 2 rows: down_proj
    // 8.7 -> 7.3 us, graph step 718 -> 690 us; at 4 rows (123 KB) a tie, so the cap stays below it.  SPARKMI_TUNE2 bit 524288: 40 KB for all)
    if (per_wave * NW <= (size_t)((NW >= 14 && !(L->tune2 & 524288)) ? 96 : 40) * 1024) {
      p.ldsb = (int)per_wave;
      int sh = 4;                                       // 16 lanes fetch 12 pieces (M = 1)
      while ((1 << sh) < 12 * p.M) ++sh;
      p.lt_shift = sh;
      lds += per_wave * NW;
    }
  }
  if (lds > 64 * 1024) {   // more than the default dynamic LDS window: opt in once per instantiation AND device
    // (the attribute belongs to the function on one device; several handles / host threads may get here together)
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    SMI_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < 64 && !done[dev]) {
      SMI_HIP(hipFuncSetAttribute((const void*)k_gemm<MT, NTB, NW, U, WB, PRO, EPI, 1, H, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      SMI_HIP(hipFuncSetAttribute((const void*)k_gemm<MT, NTB, NW, U, WB, PRO, EPI, 0, H, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      done[dev] = true;
    }
  }
  const int groups = (p.M + MT * 16 - 1) / (MT * 16);   // one block row per MT*16 rows (more than one: prefill, or 17..32 rows as 2 x 16)
  SMI_REQUIRE(groups == 1 || EPI != EPI_LM, "lm_head takes at most 32 rows per launch");
  constexpr int kLean = (MT == 1 && EPI != EPI_LM) ? 1 : 0;
  if (kLean && p.ldsb > 0 && !p.stamps && p.M == 1 && p.lt_shift == 4) {
    if constexpr (PRO == PRO_NORM && EPI == EPI_SWIGLU && H == 1 && MT == 1) {
      if (p.part_o) {   // one row behind the fused o_proj: the operand is built in the kernel from the per-head partials
#ifndef SMI_NOH16   // (A/B build: make variant NAME=noh16 VARFLAGS=-DSMI_NOH16 keeps the 16-slot form)
        if (p.n_oheads == 14) hipLaunchKernelGGL((k_gemm<MT, NTB, NW, U, WB, PRO_FUSEDO, EPI, 0, H, OCC, 2, 14>), dim3(work + helpers, groups), dim3(NW * 64), lds + 1024, st, p);
        else if (p.n_oheads == 4) hipLaunchKernelGGL((k_gemm<MT, NTB, NW, U, WB, PRO_FUSEDO, EPI, 0, H, OCC, 2, 4>), dim3(work + helpers, groups), dim3(NW * 64), lds + 1024, st, p);
        else
#endif
        hipLaunchKernelGGL((k_gemm<MT, NTB, NW, U, WB, PRO_FUSEDO, EPI, 0, H, OCC, 2>), dim3(work + helpers, groups), dim3(NW * 64), lds + 1024, st, p);
        SMI_LAUNCH_CHECK();
        return SMI_OK;
      }
    }
    if (L->cfg.kv_dtype)
      hipLaunchKernelGGL((k_gemm<MT, NTB, NW, U, WB, PRO, EPI, 1, H, OCC, 2 * kLean>), dim3(work + helpers, groups), dim3(NW * 64), lds, st, p);
    else
      hipLaunchKernelGGL((k_gemm<MT, NTB, NW, U, WB, PRO, EPI, 0, H, OCC, 2 * kLean>), dim3(work + helpers, groups), dim3(NW * 64), lds, st, p);
  } else if (kLean && p.ldsb > 0 && !p.stamps) {
    if (L->cfg.kv_dtype)
      hipLaunchKernelGGL((k_gemm<MT, NTB, NW, U, WB, PRO, EPI, 1, H, OCC, kLean>), dim3(work + helpers, groups), dim3(NW * 64), lds, st, p);
    else
      hipLaunchKernelGGL((k_gemm<MT, NTB, NW, U, WB, PRO, EPI, 0, H, OCC, kLean>), dim3(work + helpers, groups), dim3(NW * 64), lds, st, p);
  } else if (L->cfg.kv_dtype)
    hipLaunchKernelGGL((k_gemm<MT, NTB, NW, U, WB, PRO, EPI, 1, H, OCC>), dim3(work + helpers, groups), dim3(NW * 64), lds, st, p);
  else
    hipLaunchKernelGGL((k_gemm<MT, NTB, NW, U, WB, PRO, EPI, 0, H, OCC>), dim3(work + helpers, groups), dim3(NW * 64), lds, st, p);
  SMI_LAUNCH_CHECK();
  return SMI_OK;
}

// NW (the k-tile -> wave map) is fixed per kernel type for every M; only the batch depth U shrinks
// for two m-tiles (register budget), which does not change any summation order.
template <int NTB, int NW, int U, int WB, int PRO, int EPI, int H = 1, int N2 = 1, int OCC = 1>
int launch_gemm(const smi_llm* L, const GemmP& p, hipStream_t st) {
  // two m-tiles: the operand triples (12 * M pieces per k tile) outweigh the weight tile 6:1, so where there are
  // n tiles to spare (gate_up: 608) a block takes N2 of them per operand fetch.  Any NTB gives the same bits:
  // a tile's k -> wave map does not change.  (Measured at M = 32: gate_up 19.4 -> 13.7 us with N2 = 2, 14.4 with 4;
  // QKV / o_proj / down / lm_head lose with fewer blocks.)
  if (p.M > 16) {
    if constexpr (N2 > 1) {
      // 17..32 rows: three n tiles per block and ALL of a wave's k tiles in one batch -- every load of the block leaves at
      // entry (one memory round trip instead of two dependent ones; 233 VGPRs, one 8-wave block per CU, 203 blocks):
      // gate_up 12.8 -> 10.5 us, batch-32 step 1047 -> 998 us.  Beyond 32 rows (two block rows = 406 blocks, two rounds
      // at one block per CU) and with SPARKMI_TUNE2 bit 2048 the two-tile, two-batch shape stays.
      // (five n tiles per block in two batches -- 244 blocks, one round: 13.3 us against 10.3, step 947 -> 1017: not a rounds problem)
      if (p.M <= L->gu1_rows && !(L->tune2 & 2048)) return launch_gemm_kv<2, 3, NW, 4, 1, PRO, EPI>(L, p, st);
      return launch_gemm_kv<2, NTB * N2, NW, 2, 1, PRO, EPI>(L, p, st);   // two k tiles in flight: 13.9 -> 13.6 us at 32 rows, 18.5 -> 17.5 at 64
    }
    // few n tiles (N = 896 / 1152): 16-row blocks in two block rows put twice the CUs to work and halve the operand
    // bytes per CU (measured at M = 32; SPARKMI_TUNE2 bit 0 keeps 32-row blocks)
    // (WB batches of weight tiles requested at entry, as at 16 rows and fewer: down_proj's five batches then wait for L2
    // operand round trips only, not for five dependent HBM round trips; SPARKMI_TUNE2 bit 4096 keeps WB = 1 for A/B)
    if constexpr (N2 == 0) {
      if (!(L->tune2 & 1)) return (L->tune2 & 4096) ? launch_gemm_kv<1, NTB, NW, U, 1, PRO, EPI, H>(L, p, st) : launch_gemm_kv<1, NTB, NW, U, WB, PRO, EPI, H>(L, p, st);
    }
    return launch_gemm_kv<2, NTB, NW, (NW >= 16 || NTB >= 4 ? 2 : (U > 4 ? 4 : U)), 1, PRO, EPI>(L, p, st);
  }
  if constexpr (N2 > 1) {
    // gate_up at gu1_lo..16 rows: three n tiles per block and all of a wave's k tiles in one batch, as at 17..32 rows -- every load
    // of the block leaves at entry (one memory round trip).  8 rows 8.0 -> 6.9 us (graph step 801 -> 773 us), 16 rows 9.5 -> 7.9
    // (855 -> 817), 5 rows 763 -> 751, 4 rows 749 -> 742, 2 rows a tie (below 4 rows the operands staged through wave-private LDS
    // stay); same bits (the k tile -> wave map does not change).  SPARKMI_GU1_LO sets the first row count (A/B).
    if (p.M >= L->gu1_lo && !(L->tune2 & 131072)) return launch_gemm_kv<1, 3, NW, 4, 1, PRO, EPI>(L, p, st);
  }
  return launch_gemm_kv<1, NTB, NW, U, WB, PRO, EPI, H, OCC>(L, p, st);
}

const unsigned char* sec(const smi_llm* L, int s, int layer) {
  if (s >= SMI_LLM_FINAL_NORM) return L->arena + L->lay.off[s];
  return L->arena + L->lay.layers_base + (size_t)layer * L->lay.layer_stride + L->lay.off[s];
}

void* kv_layer(const smi_llm* L, void* base, int layer) {
  const size_t esz = L->cfg.kv_dtype ? 4 : 2;
  return (unsigned char*)base + (size_t)layer * L->kv_layer_elems * esz;
}

#ifndef SMI_DIAG   // product build: no engine (the launch path everywhere)
inline void eng_destroy(smi_llm*) {}
inline bool eng_usable(const smi_llm*, const RowDesc*, int) { return false; }
inline int eng_launch(smi_llm*, hipStream_t) { return SMI_OK; }
inline int eng_check(smi_llm*) { return SMI_OK; }
#else
// ---- one-row decode engine (smi_eng.h): build at create, launch in place of the 4 x layers launches of a one-row step
void eng_destroy(smi_llm* L) {
  EngState& E = L->eng;
  void* ptrs[] = {E.plan, E.stream, E.gran, E.words, E.stamps};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  E.plan = nullptr; E.stream = nullptr; E.gran = nullptr; E.words = nullptr; E.stamps = nullptr; E.enabled = 0;
}

// Not an error when the engine does not apply (other KV type, paged cache, odd shapes, small device): `why` says so and the
// launch path is used.  An allocation failure IS reported.
int eng_create(smi_llm* L) {
  EngState& E = L->eng;
  const smi_llm_cfg& c = L->cfg;
  E.enabled = 0;
  auto skip = [&](const char* m) { snprintf(E.why, sizeof(E.why), "%s", m); return SMI_OK; };
  { const char* e = smi_env("SPARKMI_ENGINE"); if (e && e[0] == '0') return skip("SPARKMI_ENGINE=0"); }
  if (c.kv_dtype != 0) return skip("f32 KV cache");
  if (L->exact) return skip("exact-weights mode");
  if (L->paged) return skip("paged KV cache");
  if (L->tune[0] || L->tune[1] || L->tune[2] || L->tune[3]) return skip("SPARKMI_TUNE set");
  int dev = 0, ncu = 0;
  SMI_HIP(hipGetDevice(&dev));
  SMI_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  { const char* e = smi_env("SPARKMI_ENGINE_CUS"); if (e && atoi(e) > 0 && atoi(e) <= ncu) ncu = atoi(e); }
  EngPlanHost P;
  char why[128];
  if (!eng_build_plan(L->H, L->Q, L->KV, L->I, c.num_heads, ncu, P, why, sizeof(why))) return skip(why);
  const EngLds lds = eng_lds(L->H, P.g.KT[EPH_DOWN], P.g.KT[EPH_QKV] > P.g.KT[EPH_O] ? P.g.KT[EPH_QKV] : P.g.KT[EPH_O]);
  if (lds.total > 160 * 1024 - 512) return skip("LDS image too large");
  E.ncu = ncu; E.maxlen = P.maxlen; E.lds = lds.total;
  E.gran_per_buf = 2 * L->H + (L->Q + 2 * L->KV) + L->Q + L->I;
  { const char* e = smi_env("SPARKMI_ENGINE_BURST"); E.ld_burst = e && atoi(e) > 0 ? atoi(e) : 16; }
  { const char* e = smi_env("SPARKMI_ENGINE_POLL"); E.poll_quiet = e ? atoi(e) : 1; }
  { const char* e = smi_env("SPARKMI_ENGINE_DELAYS"); for (int i = 0; i < 5; ++i) E.edge_delay[i] = 16; if (e) sscanf(e, "%d,%d,%d,%d,%d", &E.edge_delay[0], &E.edge_delay[1], &E.edge_delay[2], &E.edge_delay[3], &E.edge_delay[4]); }
  { const char* e = smi_env("SPARKMI_ENGINE_SLEEP"); E.ld_sleep = e && atoi(e) >= 0 ? atoi(e) : 0; }
  { const char* e = smi_env("SPARKMI_ENGINE_TIMEOUT_MS"); const double ms = e ? atof(e) : 500.0; E.timeout_ticks = (unsigned)((ms > 1.0 ? ms : 1.0) * 1e5); }
  const size_t ncw = (size_t)ncu;   // one stream per CU
  const size_t stream_bytes = (size_t)c.num_layers * ncw * P.maxlen * 1024;
  uint32_t* desc_dev = nullptr;
  uint16_t* lens_dev = nullptr;
  auto oom = [&](const char* what, size_t bytes) {
    smi_set_error("hipMalloc(engine %s, %zu bytes) failed", what, bytes);
    if (desc_dev) (void)hipFree(desc_dev);
    if (lens_dev) (void)hipFree(lens_dev);
    eng_destroy(L);
    return SMI_ENOMEM;
  };
  if (hipMalloc((void**)&E.plan, P.cu.size() * sizeof(EngCuPlan)) != hipSuccess) return oom("plan", P.cu.size() * sizeof(EngCuPlan));
  if (hipMalloc((void**)&E.stream, stream_bytes) != hipSuccess) return oom("weight stream", stream_bytes);
  if (hipMalloc((void**)&E.gran, (size_t)2 * E.gran_per_buf * 8) != hipSuccess) return oom("granules", (size_t)2 * E.gran_per_buf * 8);
  if (hipMalloc((void**)&E.words, 256) != hipSuccess) return oom("words", 256);
  if (hipMalloc((void**)&E.stamps, (size_t)3 * c.num_layers * 16 * 8) != hipSuccess) return oom("stamps", (size_t)3 * c.num_layers * 128);
  if (hipMalloc((void**)&desc_dev, P.desc.size() * 4) != hipSuccess) return oom("descriptors", P.desc.size() * 4);
  if (hipMalloc((void**)&lens_dev, P.lens.size() * 2) != hipSuccess) return oom("lengths", P.lens.size() * 2);
  // (a failing copy / memset frees the two temporaries and the engine's buffers like a failing allocation does)
#define SMI_ENG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { smi_set_error("%s: %s", #call, hipGetErrorString(e_)); \
    (void)hipFree(desc_dev); (void)hipFree(lens_dev); eng_destroy(L); return SMI_EHIP; } } while (0)
  SMI_ENG_HIP(hipMemcpy(E.plan, P.cu.data(), P.cu.size() * sizeof(EngCuPlan), hipMemcpyHostToDevice));
  SMI_ENG_HIP(hipMemcpy(desc_dev, P.desc.data(), P.desc.size() * 4, hipMemcpyHostToDevice));
  SMI_ENG_HIP(hipMemcpy(lens_dev, P.lens.data(), P.lens.size() * 2, hipMemcpyHostToDevice));
  SMI_ENG_HIP(hipMemset(E.gran, 0, (size_t)2 * E.gran_per_buf * 8));
  SMI_ENG_HIP(hipMemset(E.words, 0, 256));
  SMI_ENG_HIP(hipMemset(E.stamps, 0, (size_t)3 * c.num_layers * 128));
  EngPackP pk;
  memset(&pk, 0, sizeof(pk));
  pk.desc = desc_dev; pk.lens = lens_dev; pk.maxlen = P.maxlen; pk.ncw = (int)ncw;
  pk.arena = L->arena; pk.layers_base = L->lay.layers_base; pk.layer_stride = L->lay.layer_stride;
  const int secs[4] = {SMI_LLM_WQKV, SMI_LLM_WO, SMI_LLM_WGU, SMI_LLM_WD};
  for (int ph = 0; ph < 4; ++ph) { pk.woff[ph] = L->lay.off[secs[ph]]; pk.KT[ph] = P.g.KT[ph]; pk.NW[ph] = P.g.NW[ph]; pk.wperm[ph] = 0; }
  pk.wperm[EPH_DOWN] = L->wd_parts;
  pk.out = (uint4*)E.stream;
  const size_t imgs = ncw * P.maxlen;
  hipLaunchKernelGGL(k_eng_pack, dim3((unsigned)((imgs + 3) / 4), (unsigned)c.num_layers), dim3(256), 0, 0, pk);
  SMI_ENG_HIP(hipGetLastError());
  SMI_ENG_HIP(hipDeviceSynchronize());
#undef SMI_ENG_HIP
  (void)hipFree(desc_dev);
  (void)hipFree(lens_dev);
  if (hipFuncSetAttribute((const void*)k_engine, hipFuncAttributeMaxDynamicSharedMemorySize, E.lds) != hipSuccess) {
    (void)hipGetLastError();
    eng_destroy(L);
    return skip("hipFuncSetAttribute(LDS) refused");
  }
  snprintf(E.why, sizeof(E.why), "on: %d CUs, %d images per wave and layer at most, %d B of LDS", ncu, P.maxlen, E.lds);
  E.enabled = 1;
  return SMI_OK;
}

// the one-row step the engine stands for: one live sequence in slot 0, contiguous bf16 cache, one context segment
bool eng_usable(const smi_llm* L, const RowDesc* rows, int M) {
  return L->eng.enabled && L->eng_on && M == 1 && rows == L->rows && L->identity_slots && !L->paged && L->attn_seg <= 1 && !L->stamps_on;
}

int eng_launch(smi_llm* L, hipStream_t st) {
  const EngState& E = L->eng;
  const smi_llm_cfg& c = L->cfg;
  EngP p;
  memset(&p, 0, sizeof(p));
  p.H = L->H; p.Q = L->Q; p.KV = L->KV; p.I = L->I; p.n_heads = c.num_heads; p.n_kv = c.num_kv_heads; p.layers = c.num_layers;
  p.max_pos = c.max_positions; p.eps = c.rms_eps;
  const EngGeom g = eng_geom(L->H, L->Q, L->KV, L->I, c.num_heads);
  for (int ph = 0; ph < 4; ++ph) { p.KT[ph] = g.KT[ph]; p.NW[ph] = g.NW[ph]; }
  p.plan = E.plan; p.stream = E.stream; p.maxlen = E.maxlen; p.ncu = E.ncu;
  p.arena = L->arena; p.layers_base = L->lay.layers_base; p.layer_stride = L->lay.layer_stride;
  p.off_ln1 = L->lay.off[SMI_LLM_LN1]; p.off_bqkv = L->lay.off[SMI_LLM_BQKV]; p.off_ln2 = L->lay.off[SMI_LLM_LN2];
  p.final_norm = (const float*)(L->arena + L->lay.off[SMI_LLM_FINAL_NORM]);
  p.rope = (const float2*)(L->arena + L->lay.off[SMI_LLM_ROPE]);
  p.rows = L->rows; p.h = L->h; p.ss_in = L->sspart; p.xs_out = L->xs_h; p.ss_out = L->sspart;
  p.kcache = (uint16_t*)L->kcache; p.vcache = (uint16_t*)L->vcache; p.kv_layer_elems = L->kv_layer_elems;
  p.gran = E.gran; p.serial = E.words; p.err = E.words + 4; p.arrive = E.words + 8;
  p.timeout_ticks = E.timeout_ticks;
  p.ld_burst = E.ld_burst; p.ld_sleep = E.ld_sleep; p.poll_quiet = E.poll_quiet;
  for (int i = 0; i < 5; ++i) p.edge_delay[i] = E.edge_delay[i];
  p.stamps = smi_env("SPARKMI_ENGINE_STAMPS") ? E.stamps : nullptr;
  hipLaunchKernelGGL(k_engine, dim3(E.ncu), dim3(kEngBlock), E.lds, st, p);
  SMI_LAUNCH_CHECK();
  return SMI_OK;
}

// After a stream synchronisation: did an engine launch give up on a hand-off?  (bounded spins, smi_eng_comm.h)
int eng_check(smi_llm* L) {
  if (!L->eng.enabled) return SMI_OK;
  unsigned e[2] = {0, 0};
  SMI_HIP(hipMemcpy(e, L->eng.words + 4, 8, hipMemcpyDeviceToHost));
  if (e[0] == 0) return SMI_OK;
  SMI_HIP(hipMemset(L->eng.words + 4, 0, 8));
  L->started = 0;
  smi_set_error("decode engine: a hand-off timed out (layer %u, edge %u) -- are all %d CUs free for this stream?  SPARKMI_ENGINE=0 selects the launch path",
                (e[1] - 1) / 8, (e[1] - 1) % 8, L->eng.ncu);
  return SMI_EHIP;
}
#endif   // SMI_DIAG

enum { KQKV = 0, KATTN, KO, KGU, KD, KLM, KFIN };
enum { PF_GROUPED = 1, PF_PGEMM = 2 };   // kernel family of a pass over prompt rows (launch_layers_big)
constexpr int kPgSplitRows = 512;        // passes of up to this many rows run o_proj / down_proj in k_pgemm's split-K form
constexpr int kPgSegO = 2, kPgSegD = 8;  // K segments of the prefill o_proj / down_proj sums (fixed: part of the association)

int ensure_apart(smi_llm* L, size_t floats) {
  if (floats <= L->apart_floats) return SMI_OK;
  graphs_flush(L);   // captured steps hold the old buffer's address
  if (L->apart) (void)hipFree(L->apart);
  L->apart = nullptr; L->apart_floats = 0;
  if (hipMalloc((void**)&L->apart, floats * 4) != hipSuccess) { smi_set_error("hipMalloc(attention partials, %zu floats) failed", floats); return SMI_ENOMEM; }
  L->apart_floats = floats;
  return SMI_OK;
}

// lm_head partial columns (= persistent blocks) finalize / the sampler read for M live rows
int lm_blocks_for(const smi_llm* L, int M) {
  if (L->KTh > 32 || L->exact) return L->lm_cap;
  return M <= 16 ? L->lm_blocks : (L->lm_blocks < 256 ? L->lm_blocks : 256);
}

int segs_for(int ctx_bound) { return ctx_bound <= kAttnSeg ? 1 : (ctx_bound + kAttnSeg - 1) / kAttnSeg; }

// attention (+ the segment merge when the context bound of the current call exceeds one segment)
template <int KVF32>
int launch_attn(smi_llm* L, AttnP a, int helpers_ok, hipStream_t st) {
  a.nseg = L->attn_seg < 1 ? 1 : L->attn_seg;
  a.work_blocks = a.n_heads * a.M * a.nseg;
  const bool fuse = a.M == 1 && a.nseg == 1 && a.slot_is_row && a.part_o && !a.cnt;
  const bool fuse2 = a.nseg == 1 && a.slot_is_row && a.part_o && a.cnt;   // (a.cnt is only set by fuse2_now)   // 2 .. 8 rows: fused o_proj + last-arriver head sum
  if (fuse || fuse2) a.work_blocks *= kFuseQB;
  if (a.nseg > 1) {
    int rc = ensure_apart(L, (size_t)a.M * a.n_heads * a.nseg * 66);
    if (rc) return rc;
    a.part = L->apart;
  }
  const int helpers = (helpers_ok && (L->prefetch_mask & 2) && a.M <= L->prefetch_rows && a.work_blocks < 232) ? (256 - a.work_blocks) / 8 * 8 : 0;
  const bool pg = a.km.ptab != nullptr;   // paged cache: the slot == row kernels look the token's page up (PG)
#ifdef SMI_DIAG   // measured, not adopted: the product library does not carry these instantiations
  if (fuse2 && pg)
    hipLaunchKernelGGL((k_attn<KVF32, 2, 2, 1>), dim3(a.work_blocks), dim3(2 * kAttnWaves * 64), 0, st, a);
  else if (fuse2)
    hipLaunchKernelGGL((k_attn<KVF32, 2, 2>), dim3(a.work_blocks), dim3(2 * kAttnWaves * 64), 0, st, a);
  else
#endif
  if (fuse && pg)
    hipLaunchKernelGGL((k_attn<KVF32, 1, 1, 1>), dim3(a.work_blocks + helpers), dim3(2 * kAttnWaves * 64), 0, st, a);
  else if (fuse)
    hipLaunchKernelGGL((k_attn<KVF32, 1, 1>), dim3(a.work_blocks + helpers), dim3(2 * kAttnWaves * 64), 0, st, a);
  else if (a.M == 1 && a.nseg == 1 && a.slot_is_row && pg)
    hipLaunchKernelGGL((k_attn<KVF32, 1, 0, 1>), dim3(a.work_blocks + helpers), dim3(kAttnWaves * 64), 0, st, a);
  else if (a.M == 1 && a.nseg == 1 && a.slot_is_row)
    hipLaunchKernelGGL((k_attn<KVF32, 1>), dim3(a.work_blocks + helpers), dim3(kAttnWaves * 64), 0, st, a);
  else if (a.nseg == 1 && a.slot_is_row && pg)
    hipLaunchKernelGGL((k_attn<KVF32, 2, 0, 1>), dim3(a.work_blocks + helpers), dim3(kAttnWaves * 64), 0, st, a);
  else if (a.nseg == 1 && a.slot_is_row)
    hipLaunchKernelGGL((k_attn<KVF32, 2>), dim3(a.work_blocks + helpers), dim3(kAttnWaves * 64), 0, st, a);
  else
    hipLaunchKernelGGL((k_attn<KVF32, 0>), dim3(a.work_blocks + helpers), dim3(kAttnWaves * 64), 0, st, a);
  SMI_LAUNCH_CHECK();
  if (a.nseg > 1) {
    hipLaunchKernelGGL(k_attn_merge, dim3(a.M), dim3(1024), 0, st, a);
    SMI_LAUNCH_CHECK();
  }
  return SMI_OK;
}

// One live row in its own slot, one context segment: the attention kernel also computes its head's share of the o_proj
// (k_attn<.., FUSE>), gate_up builds its operand from those partials (PRO_FUSEDO) and down_proj adds its residual from h2;
// the o_proj kernel is not launched.  The predicate is the one launch_attn picks the one-row kernel by.
bool fuse_o_now(const smi_llm* L, const RowDesc* rows, int M) {
  return L->fuse_o && !L->fuse1_last && M == 1 && rows == L->rows && L->identity_slots && L->attn_seg <= 1;
}
// 2 .. fuse2_rows live rows, each in its own slot, one context segment: the attention kernel also computes the o_proj and the last
// block of a (row, quarter) to arrive finishes the rows (k_attn<.., FUSE = 2>); the o_proj kernel is not launched.
bool fuse2_now(const smi_llm* L, const RowDesc* rows, int M) {
  return L->fuse_o && M >= (L->fuse1_last ? 1 : 2) && M <= L->fuse2_rows && rows == L->rows && L->identity_slots && L->attn_seg <= 1;
}

// o_proj: NW = number of heads, so that wave w sums head w's two (head-interleaved) k tiles -- the per-head partial the
// fused path produces -- and the block's in-order sum over the waves is the fused consumer's in-order sum over the heads.
int launch_oproj(const smi_llm* L, const GemmP& p, int M, hipStream_t st) {
  switch (L->cfg.num_heads) {
    case 14: return M <= 8 ? launch_gemm<1, 14, 2, 1, PRO_PLAIN, EPI_RESID, 4>(L, p, st) : launch_gemm<1, 14, 2, 1, PRO_PLAIN, EPI_RESID, 1, 0>(L, p, st);
    case 4: return M <= 8 ? launch_gemm<1, 4, 2, 1, PRO_PLAIN, EPI_RESID, 4>(L, p, st) : launch_gemm<1, 4, 2, 1, PRO_PLAIN, EPI_RESID, 1, 0>(L, p, st);
    default:   // other head counts: no fused path (smi_llm_create leaves fuse_o off), eight waves as before
      return M <= 8 ? launch_gemm<1, 8, 4, 1, PRO_PLAIN, EPI_RESID, 4>(L, p, st) : launch_gemm<1, 8, 4, 1, PRO_PLAIN, EPI_RESID, 1, 0>(L, p, st);
  }
}

// down_proj with the chains split over blocks (k_downC) and the in-order combine + RESID epilogue (k_resid_comb)
template <int TPC, int MTN>
int launch_down_chains_t(const smi_llm* L, const DownCP& d, hipStream_t st) {
  const size_t lds = (size_t)TPC * 12 * (d.M < MTN * 16 ? d.M : MTN * 16) * 16;
  if (lds > 64 * 1024) {   // opt in once per instantiation and device (as in launch_gemm_kv)
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    SMI_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < 64 && !done[dev]) {
      SMI_HIP(hipFuncSetAttribute((const void*)k_downC<TPC, MTN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      done[dev] = true;
    }
  }
  hipLaunchKernelGGL((k_downC<TPC, MTN>), dim3(d.NT / 4 * 16), dim3(256), lds, st, d);
  SMI_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_resid_comb<0, 0>), dim3(d.NT * MTN), dim3(64), 0, st, d, 16, MTN);
  SMI_LAUNCH_CHECK();
  return SMI_OK;
}

int launch_down_chains(const smi_llm* L, const GemmP& p, hipStream_t st) {
  DownCP d;
  d.W = p.W; d.NT = p.NT; d.KT = p.KT; d.M = p.M; d.wperm = p.wperm; d.XS = p.XS; d.part = L->dpart;
  d.Y = p.Y; d.Yin = p.Yin; d.gamma_next = p.gamma_next; d.XSout = p.XSout; d.ssout = p.ssout;
#define SMI_DC(TPC_) (p.M <= 16 ? launch_down_chains_t<TPC_, 1>(L, d, st) : p.M <= 32 ? launch_down_chains_t<TPC_, 2>(L, d, st) : launch_down_chains_t<TPC_, 4>(L, d, st))
  if (p.KT <= 32) return SMI_DC(2);
  if (p.KT <= 96) return SMI_DC(6);
  return SMI_DC(10);
#undef SMI_DC
}

// exact-weights mode: the GEMM of launch_one's kernel `which` on k_gemm_x (p prepared as for k_gemm; W = fp32 [N][K])
template <int PRO, int EPI>
int launch_gemm_x(const smi_llm* L, const GemmP& p, hipStream_t st) {
  const dim3 grid((p.NT + 3) / 4, (p.M + 15) / 16);
  if (L->cfg.kv_dtype) hipLaunchKernelGGL((k_gemm_x<PRO, EPI, 1>), grid, dim3(256), 0, st, p, (const float*)p.W);
  else hipLaunchKernelGGL((k_gemm_x<PRO, EPI, 0>), grid, dim3(256), 0, st, p, (const float*)p.W);
  SMI_LAUNCH_CHECK();
  return SMI_OK;
}

int launch_one(smi_llm* L, int which, int layer, const RowDesc* rows, int M, float* logits, hipStream_t st) {
  const smi_llm_cfg& c = L->cfg;
  const bool fused = fuse_o_now(L, rows, M);
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.M = M; p.rows = rows; p.eps = c.rms_eps; p.sspart = L->sspart; p.npart = L->NTh * 4;
  if (L->exact && which != KATTN && which != KFIN) {
    // verification mode: same operands, scalars and epilogues, fp32 weights, one exact fp32 FMA chain per output (k_gemm_x)
    switch (which) {
      case KQKV:
        p.W = (const uint4*)sec(L, SMI_LLM_WQKV, layer); p.NT = L->NTqkv; p.KT = L->KTh; p.XS = L->xs_h;
        p.Y = L->qbuf; p.bias = (const float*)sec(L, SMI_LLM_BQKV, layer); p.rope = (const float2*)sec(L, SMI_LLM_ROPE, 0);
        p.kcache = kv_layer(L, L->kcache, layer); p.vcache = kv_layer(L, L->vcache, layer);
        p.q_dim = L->Q; p.kv_dim = L->KV; p.n_kv = c.num_kv_heads; p.max_pos = c.max_positions; p.km = kv_map(L);
        return launch_gemm_x<PRO_NORM, EPI_QKV>(L, p, st);
      case KO:
        p.W = (const uint4*)sec(L, SMI_LLM_WO, layer); p.NT = L->NTh; p.KT = L->KTq; p.XS = L->xs_attn; p.Y = L->h;
        p.XSout = L->xs_h; p.gamma_next = (const float*)sec(L, SMI_LLM_LN2, layer); p.ssout = L->sspart;
        return launch_gemm_x<PRO_PLAIN, EPI_RESID>(L, p, st);
      case KGU:
        p.W = (const uint4*)sec(L, SMI_LLM_WGU, layer); p.NT = L->NTgu; p.KT = L->KTh; p.XS = L->xs_h; p.XSout = L->xs_act;
        return launch_gemm_x<PRO_NORM, EPI_SWIGLU>(L, p, st);
      case KD:
        p.W = (const uint4*)sec(L, SMI_LLM_WD, layer); p.NT = L->NTh; p.KT = L->KTi; p.XS = L->xs_act; p.Y = L->h;
        p.XSout = L->xs_h; p.ssout = L->sspart;
        p.gamma_next = layer + 1 < c.num_layers ? (const float*)sec(L, SMI_LLM_LN1, layer + 1) : (const float*)sec(L, SMI_LLM_FINAL_NORM, 0);
        return launch_gemm_x<PRO_PLAIN, EPI_RESID>(L, p, st);
      case KLM:
        p.W = (const uint4*)sec(L, SMI_LLM_LM_HEAD, 0); p.NT = L->NTlm; p.KT = L->KTh; p.XS = L->xs_h;
        p.Y = logits ? logits : (L->do_sample ? L->logits : nullptr);
        p.V = c.vocab_size; p.pval = L->pval; p.pidx = L->pidx;
        SMI_REQUIRE((L->NTlm + 3) / 4 <= L->lm_cap, "lm_head partial buffer too small");
        return launch_gemm_x<PRO_NORM, EPI_LM>(L, p, st);
    }
  }
  switch (which) {
    case KQKV:
      p.W = (const uint4*)sec(L, SMI_LLM_WQKV, layer); p.NT = L->NTqkv; p.KT = L->KTh;
      p.XS = L->xs_h;
      p.Y = L->qbuf; p.bias = (const float*)sec(L, SMI_LLM_BQKV, layer);
      p.rope = (const float2*)sec(L, SMI_LLM_ROPE, 0);
      p.kcache = kv_layer(L, L->kcache, layer); p.vcache = kv_layer(L, L->vcache, layer);
      p.q_dim = L->Q; p.kv_dim = L->KV; p.n_kv = c.num_kv_heads; p.max_pos = c.max_positions; p.km = kv_map(L);
      // helpers: first half of this layer's gate_up slices (consumer block b reads weight tile row b)
      // helpers: the first pf_qkv_eighths / 8 of this layer's gate_up slices (consumer block b reads weight tile row b).  Measured
      // at one row, graph step, one box (profiles/r03_prefetch.txt): 8/8 600 us, 6/8 569, 4/8 564, 2/8 559 (default), none 566-571
      p.pf = PfDesc{sec(L, SMI_LLM_WGU, layer), L->KTh * 1024, L->NTgu, 0, ((L->NTgu + 7) / 8 * L->pf_qkv_eighths + 7) / 8};
      switch (L->tune[0]) {   // SPARKMI_TUNE=q,o,g,d: block-shape sweeps (diagnostics; NW changes the summation order)
        case 1: return launch_gemm<1, 8, 4, 1, PRO_NORM, EPI_QKV>(L, p, st);
        case 3: return launch_gemm<1, 4, 8, 1, PRO_NORM, EPI_QKV>(L, p, st);
        default: return launch_gemm<1, 16, 2, 1, PRO_NORM, EPI_QKV, 1, 0>(L, p, st);   // measured best (profiles/README.md)
      }
    case KATTN: {
      AttnP a;
      memset(&a, 0, sizeof(a));
      a.q = L->qbuf; a.kcache = kv_layer(L, L->kcache, layer); a.vcache = kv_layer(L, L->vcache, layer);
      a.rows = rows; a.xs_out = L->xs_attn; a.M = M; a.q_dim = L->Q; a.n_kv = c.num_kv_heads;
      a.group = c.num_heads / c.num_kv_heads; a.max_pos = c.max_positions; a.n_heads = c.num_heads; a.km = kv_map(L);
      a.slot_is_row = rows == L->rows && L->identity_slots;   // the live decode rows are (slot b, ...) in order (contiguous cache: slots contiguous; paged: through the page table)
      // helpers: second half of this layer's gate_up slices
      a.pf = PfDesc{sec(L, SMI_LLM_WGU, layer), L->KTh * 1024, L->NTgu, (L->NTgu / 8 + 1) / 2, (L->NTgu + 7) / 8};
      if (fused) { a.Wo = (const uint4*)sec(L, SMI_LLM_WO, layer); a.NTo = L->NTh; a.part_o = L->part_o; }
      if (fuse2_now(L, rows, M)) {
        a.Wo = (const uint4*)sec(L, SMI_LLM_WO, layer); a.NTo = L->NTh; a.part_o = L->part_o; a.cnt = L->fuse_cnt;
        a.h = L->h; a.gamma_next = (const float*)sec(L, SMI_LLM_LN2, layer); a.xs_next = L->xs_h; a.ssout = L->sspart;
      }
      return c.kv_dtype ? launch_attn<1>(L, a, 1, st) : launch_attn<0>(L, a, 1, st);
    }
    case KO:   // h += Wo attn; emits the post-attention norm's operand
      if (fused) return SMI_OK;   // done by the attention kernel (per-head partials) and gate_up's prologue
      if (fuse2_now(L, rows, M)) return SMI_OK;   // done by the attention kernel (per-head partials + last-arriver head sum and epilogue)
      p.W = (const uint4*)sec(L, SMI_LLM_WO, layer); p.NT = L->NTh; p.KT = L->KTq;
      p.XS = L->xs_attn; p.Y = L->h;
      p.XSout = L->xs_h; p.gamma_next = (const float*)sec(L, SMI_LLM_LN2, layer); p.ssout = L->sspart;
      // no helpers here: warming down_proj's 8.7 MB from o_proj (or from attention / gate_up) stretches the
      // producer by more than the consumer gains (measured, profiles/README.md), so down_proj stays cold
      switch (L->tune[1]) {
        case 2: return launch_gemm<1, 16, 2, 1, PRO_PLAIN, EPI_RESID>(L, p, st);
        case 3: return launch_gemm<1, 4, 8, 1, PRO_PLAIN, EPI_RESID>(L, p, st);
        case 4: return launch_gemm<1, 8, 4, 1, PRO_PLAIN, EPI_RESID, 2>(L, p, st);
        case 5: return launch_gemm<1, 8, 4, 1, PRO_PLAIN, EPI_RESID>(L, p, st);
        case 6: return launch_gemm<1, 8, 4, 1, PRO_PLAIN, EPI_RESID, 4>(L, p, st);
        default:   // four row parts up to 8 rows: 629.8 -> 626.0 us per step once the prologues were down to one round trip
          return launch_oproj(L, p, M, st);
      }
    case KGU:
      p.W = (const uint4*)sec(L, SMI_LLM_WGU, layer); p.NT = L->NTgu; p.KT = L->KTh;
      p.XS = L->xs_h; p.XSout = L->xs_act;
      if (fused) {
        p.part_o = L->part_o; p.n_oheads = c.num_heads; p.hres = L->h; p.h2out = L->h2;
        if (const char* e = smi_env("SPARKMI_FAKE_OHEADS")) p.n_oheads = atoi(e);   // TIMING ONLY (diagnostics build): gate_up reads that many partials -- wrong sums (profiles/r04_gate_up_two_tiles_and_fewer_partials.txt)
        if (const char* e = smi_env("SPARKMI_FAKE_OHEADS")) p.n_oheads = atoi(e);   // TIMING ONLY (diagnostics build): gate_up reads that many partials -- wrong sums
        if (L->pf_inline && layer + 1 < c.num_layers && L->NTqkv % 8 == 0) {   // SPARKMI_PF_INLINE=0: off (A/B)
          p.pf2_base = sec(L, SMI_LLM_WQKV, layer + 1); p.pf2_slice = L->KTh * 1024; p.pf2_nslices = L->NTqkv;
        }
        p.gam = (const float*)sec(L, SMI_LLM_LN2, layer);
      }
      switch (L->tune[2]) {
        case 2: return launch_gemm<1, 16, 2, 1, PRO_NORM, EPI_SWIGLU>(L, p, st);
        case 5: return launch_gemm<2, 4, 8, 1, PRO_NORM, EPI_SWIGLU>(L, p, st);
        case 6: return launch_gemm<1, 8, 4, 1, PRO_NORM, EPI_SWIGLU, 1, 2, 6>(L, p, st);   // spills at 80 VGPRs: 2.4x slower
        case 7: return launch_gemm<1, 8, 4, 1, PRO_NORM, EPI_SWIGLU, 1, 2>(L, p, st);      // 114 VGPRs: 2 blocks per CU, 96 of 608 wait
        case 8: return launch_gemm<2, 8, 2, 2, PRO_NORM, EPI_SWIGLU, 1, 2, 4>(L, p, st);   // two tiles per block: half the blocks re-read the fused o_proj partials
        default:   // 69 VGPRs (>= 6 waves per SIMD): all 608 blocks are resident at once, no second round (step 745 -> 705 us)
          return launch_gemm<1, 8, 2, 2, PRO_NORM, EPI_SWIGLU, 1, 2, 6>(L, p, st);
      }
    case KD:   // h += Wd act; emits the next layer's input-norm operand (or the final norm's)
      p.W = (const uint4*)sec(L, SMI_LLM_WD, layer); p.NT = L->NTh; p.KT = L->KTi; p.wperm = L->wd_parts;
      p.XS = L->xs_act; p.Y = L->h;
      if (fused) p.Yin = L->h2;   // h + o_proj, left there by gate_up's block 0
      p.XSout = L->xs_h; p.ssout = L->sspart;
      p.gamma_next = layer + 1 < c.num_layers ? (const float*)sec(L, SMI_LLM_LN1, layer + 1)
                                              : (const float*)sec(L, SMI_LLM_FINAL_NORM, 0);
      // helpers: the next layer's QKV slices (its o_proj slices ride along: same slice size, contiguous-ish)
      if (layer + 1 < c.num_layers) p.pf = PfDesc{sec(L, SMI_LLM_WQKV, layer + 1), L->KTh * 1024, L->NTqkv, 0, (L->NTqkv + 7) / 8};
      if (M >= L->dc_min && M <= kMaxRows && !L->tune[3] && !L->stamps_on && p.KT <= 160 && p.NT % 4 == 0)
        return launch_down_chains(L, p, st);   // 9 .. 64 rows: the 16 chains on 16 different blocks + an in-order combine (k_downC)
      switch (L->tune[3]) {
        case 2: return launch_gemm<1, 16, 3, 4, PRO_PLAIN, EPI_RESID>(L, p, st);
        case 4: return launch_gemm<1, 8, 10, 2, PRO_PLAIN, EPI_RESID>(L, p, st);
        case 5: return launch_gemm<1, 16, 2, 5, PRO_PLAIN, EPI_RESID, 2>(L, p, st);
        case 6: return launch_gemm<1, 16, 2, 5, PRO_PLAIN, EPI_RESID>(L, p, st);
        default:   // few rows: 4-row parts (224 blocks, -0.7 us); same bits either way, the zero-fed MFMAs cost at M > 8
          // row parts (H blocks per weight tile) while the operand re-reads they cost stay small: measured step at
          // 16 / 32 / 64 rows with H = 1 | 2 | 4: 930 | 915 | 900, 1064 | 1055 | 1161, 1318 | 1443 | 1671 us
          if (M <= 8) return launch_gemm<1, 16, 2, 5, PRO_PLAIN, EPI_RESID, 4>(L, p, st);
          if (M <= 16) return launch_gemm<1, 16, 2, 5, PRO_PLAIN, EPI_RESID, 4, 0>(L, p, st);
          if (M <= 32) return launch_gemm<1, 16, 2, 5, PRO_PLAIN, EPI_RESID, 2, 0>(L, p, st);
          return launch_gemm<1, 16, 2, 5, PRO_PLAIN, EPI_RESID, 1, 0>(L, p, st);
      }
    case KLM:
      p.W = (const uint4*)sec(L, SMI_LLM_LM_HEAD, 0); p.NT = L->NTlm; p.KT = L->KTh;
      p.XS = L->xs_h;
      p.Y = logits ? logits : (L->do_sample ? L->logits : nullptr);
      p.V = c.vocab_size; p.pval = L->pval; p.pidx = L->pidx;
      p.stamps = L->stamps_on ? L->stamps : nullptr;
      if (L->KTh <= 32) {   // persistent path: 16 rows' operand resident in the registers of a 4-wave group
        const int ngroups = (L->NTlm + 1) / 2;
        const size_t lds = (size_t)4 * 2 * 1024 + 32 * 4 + 2 * 32 * 8;
        if (M <= 16) {
          hipLaunchKernelGGL(k_lm<1>, dim3(L->lm_blocks), dim3(256), lds, st, p, ngroups, 0);
          SMI_LAUNCH_CHECK();
        } else {              // 32 rows per pass: one 4-wave block per CU, the second m-tile's operands in LDS (k_lm32);
                              // SPARKMI_TUNE2 bit 8192: the two-group kernel (k_lm<2>) for A/B
          const size_t lds32 = (size_t)4 * 2 * 2 * 1024 + 32 * 4 + 2 * 32 * 8 + (size_t)L->KTh * 12 * 16 * 16;
          for (int m0 = 0; m0 < M; m0 += 32) {
            if (!(L->tune2 & 8192) && lds32 <= 150 * 1024) {
              if (L->KTh <= 28) hipLaunchKernelGGL(k_lm32<7>, dim3(lm_blocks_for(L, M)), dim3(256), lds32, st, p, ngroups, m0);
              else hipLaunchKernelGGL(k_lm32<8>, dim3(lm_blocks_for(L, M)), dim3(256), lds32, st, p, ngroups, m0);
            } else
              hipLaunchKernelGGL(k_lm<2>, dim3(lm_blocks_for(L, M)), dim3(512), 2 * lds, st, p, ngroups, m0);
            SMI_LAUNCH_CHECK();
          }
        }
        return SMI_OK;
      }
      SMI_REQUIRE((L->NTlm + 3) / 4 <= L->lm_cap, "lm_head partial buffer too small");
      return launch_gemm<4, 4, 2, 1, PRO_NORM, EPI_LM>(L, p, st);
    case KFIN: {
      FinP f;
      f.tok = nullptr;
      if (L->do_sample) {
        SampleP sp;
        sp.logits = L->logits; sp.V = c.vocab_size; sp.top_k = L->top_k; sp.inv_temp = 1.0f / L->temperature;
        sp.top_p = L->top_p; sp.ctl = L->ctl; sp.rows = L->rows; sp.tok = L->tok;
        sp.pval = L->pval; sp.nblk = lm_blocks_for(L, M);
        sp.cand_v = L->cand_v; sp.cand_i = L->cand_i; sp.cand_n = L->cand_n;
        hipLaunchKernelGGL(k_sample_scan, dim3(kScanBlocks, M), dim3(256), 0, st, sp);
        SMI_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_sample, dim3(M), dim3(1024), 0, st, sp);
        SMI_LAUNCH_CHECK();
        f.tok = L->tok;
      }
      f.pval = L->pval; f.pidx = L->pidx; f.M = M; f.KT = L->KTh; f.V = c.vocab_size;
      f.nblk = lm_blocks_for(L, M);
      f.rows = L->rows; f.hist = L->hist; f.count = L->count; f.finished = L->finished; f.step = L->step;
      f.ctl = L->ctl; f.Wlm = (const uint16_t*)sec(L, SMI_LLM_LM_HEAD, 0); f.h = L->h; f.max_steps = L->max_steps;
      f.gamma0 = (const float*)sec(L, SMI_LLM_LN1, 0); f.xs = L->xs_h; f.sspart = L->sspart; f.npart = L->NTh * 4; f.exact = L->exact;
      hipLaunchKernelGGL(k_finalize, dim3(M), dim3(256), 0, st, f);
      SMI_LAUNCH_CHECK();
      return SMI_OK;
    }
  }
  smi_set_error("launch_one: bad kernel id %d", which);
  return SMI_EINVAL;
}

// One prefill GEMM over M rows (any M) with k_pgemm; `which` as in launch_one (GEMM kernels only).
// nsplit > 1 (RESID with XMAP only): one block per (tile, K segment) writes its segment's sums to L->pslab and k_resid_comb adds
// the segments in order and runs the epilogue -- the few-row form of the segmented sum (GemmP::kseg), same bits as the in-block form.
template <int PRO, int EPI, int NTW = 2, int WR = 1, int RING = 0, int XMAP = 0, int TWO = 0, int MTB = 8>
int launch_pgemm(const smi_llm* L, GemmP p, hipStream_t st, int nsplit = 1) {
  constexpr int cols = (4 / WR) * NTW;   // weight tiles per block
  constexpr int rows = MTB * 16;
  const dim3 grid2((p.NT + cols - 1) / cols, (p.M + rows - 1) / rows);
  dim3 grid = XMAP ? dim3(grid2.x * grid2.y) : grid2;
  const size_t lds = (RING ? (size_t)RING * (12 * rows + cols * 64) : (size_t)2 * 12 * rows) * 16 + (TWO ? 0 : rows * 4);
  if (lds > 64 * 1024) {   // opt in once per instantiation and device (as in launch_gemm_kv)
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    SMI_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < 64 && !done[dev]) {
      SMI_HIP(hipFuncSetAttribute((const void*)k_pgemm<PRO, EPI, 1, NTW, WR, RING, XMAP, TWO, MTB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      SMI_HIP(hipFuncSetAttribute((const void*)k_pgemm<PRO, EPI, 0, NTW, WR, RING, XMAP, TWO, MTB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      done[dev] = true;
    }
  }
  const int mtiles = (p.M + 15) / 16;
  if (nsplit > 1) {
    SMI_REQUIRE(EPI == EPI_RESID && XMAP && RING >= 3 && p.kseg > 0 && nsplit <= 16, "launch_pgemm: split-K is the RESID ring form's");
    SMI_REQUIRE(L->pslab && (size_t)nsplit * p.NT * mtiles * 1024 <= L->pslab_bytes, "launch_pgemm: partial slab too small");
    p.slab = L->pslab; p.slab_mt = mtiles;
    grid = dim3(grid2.x * grid2.y, nsplit);
  }
  if (L->cfg.kv_dtype) hipLaunchKernelGGL((k_pgemm<PRO, EPI, 1, NTW, WR, RING, XMAP, TWO, MTB>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((k_pgemm<PRO, EPI, 0, NTW, WR, RING, XMAP, TWO, MTB>), grid, dim3(256), lds, st, p);
  SMI_LAUNCH_CHECK();
  if (nsplit > 1) {
    DownCP d;
    d.W = nullptr; d.NT = p.NT; d.KT = p.KT; d.M = p.M; d.wperm = 0; d.XS = nullptr; d.part = L->pslab;
    d.Y = p.Y; d.Yin = p.Yin; d.gamma_next = p.gamma_next; d.XSout = p.XSout; d.ssout = p.ssout;
    hipLaunchKernelGGL((k_resid_comb<1, 1>), dim3(p.NT * mtiles), dim3(64), 0, st, d, nsplit, mtiles);
    SMI_LAUNCH_CHECK();
  }
  return SMI_OK;
}

// All layers for M (> 32) prompt rows living in the big workspace: K/V of every row appended, hidden
// states of the last layer never needed (no prompt row except each sequence's last feeds lm_head).
// Up to kPgemmMinRows rows the decode GEMM runs with one block row per 32 rows (same bits as 32-row
// chunks, one launch instead of M / 32); beyond, the LDS-shared prefill GEMM (k_pgemm) takes over.
int launch_layers_big(smi_llm* L, const RowDesc* rows, int M, int family, hipStream_t st) {
  const smi_llm_cfg& c = L->cfg;
  // Which GEMM family: decided by the CALLER from the sequences' own lengths (prefill_prompts), the same for all four kernels of
  // the pass.  Diagnostics (SPARKMI_PGEMM_MIN_* set): per kernel by the pass's row count, as in round 3 -- every mix is
  // parity-tested.  The RESID kernels leave the RMSNorm partials as [rows][NT * 4] (grouped) or [rows][NT] (prefill GEMM and
  // its split-K combine); their consumers are told which.
  const bool pg = family == PF_PGEMM;
  const bool gq = L->pg_forced ? M < L->pg_min[0] : !pg, go = L->pg_forced ? M < L->pg_min[1] : !pg,
             gg = L->pg_forced ? M < L->pg_min[2] : !pg, gd = L->pg_forced ? M < L->pg_min[3] : !pg;
  const int np_from_d = gd ? L->NTh * 4 : L->NTh, np_from_o = go ? L->NTh * 4 : L->NTh;
  // few rows: 64-column tiles (more blocks) and a deeper ring; o_proj / down_proj as one block per K segment + in-order combine
  const bool few = M <= L->pg_split_rows && !(L->tune2 & 16384);
  const int kseg_o = (L->KTq + kPgSegO - 1) / kPgSegO, kseg_d = (L->KTi + kPgSegD - 1) / kPgSegD;
  const int nseg_o = (L->KTq + kseg_o - 1) / kseg_o, nseg_d = (L->KTi + kseg_d - 1) / kseg_d;
  int rc;
  for (int l = 0; l < c.num_layers; ++l) {
    GemmP p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.rows = rows; p.eps = c.rms_eps; p.sspart = L->bss; p.npart = np_from_d;
    // QKV
    p.W = (const uint4*)sec(L, SMI_LLM_WQKV, l); p.NT = L->NTqkv; p.KT = L->KTh; p.XS = L->bxs_h;
    p.Y = L->bq; p.bias = (const float*)sec(L, SMI_LLM_BQKV, l); p.rope = (const float2*)sec(L, SMI_LLM_ROPE, 0);
    p.kcache = kv_layer(L, L->kcache, l); p.vcache = kv_layer(L, L->vcache, l);
    p.q_dim = L->Q; p.kv_dim = L->KV; p.n_kv = c.num_kv_heads; p.max_pos = c.max_positions; p.km = kv_map(L);
    if (gq) rc = launch_gemm<1, 16, 2, 1, PRO_NORM, EPI_QKV>(L, p, st);
    else if (L->tune2 & 16384) rc = launch_pgemm<PRO_NORM, EPI_QKV>(L, p, st);
    else if (few) rc = launch_pgemm<PRO_NORM, EPI_QKV, 2, 2, 6, 1, 0, 4>(L, p, st);
    else rc = launch_pgemm<PRO_NORM, EPI_QKV, 6, 2, 3, 1>(L, p, st);
    if (rc) return rc;
    if (l == c.num_layers - 1) break;
    // attention
    AttnP a;
    memset(&a, 0, sizeof(a));
    a.q = L->bq; a.kcache = p.kcache; a.vcache = p.vcache; a.rows = rows; a.xs_out = L->bxs_attn; a.M = M;
    a.q_dim = L->Q; a.n_kv = c.num_kv_heads; a.group = c.num_heads / c.num_kv_heads; a.max_pos = c.max_positions;
    a.n_heads = c.num_heads; a.slot_is_row = 0; a.km = kv_map(L);
    if (L->tune2 & 1024) {   // SPARKMI_TUNE2 bit 1024: the decode attention kernel per prompt row (the path before k_attn_pf)
      if ((rc = c.kv_dtype ? launch_attn<1>(L, a, 0, st) : launch_attn<0>(L, a, 0, st))) return rc;
    } else {
      const dim3 ag((unsigned)((M + kAttnWaves - 1) / kAttnWaves), (unsigned)c.num_heads);
      if (!c.kv_dtype && L->attn_pf2 && L->pf_ntiles > 0 && c.head_dim == 64) {   // bf16 KV: tiles of 16 rows on the matrix pipes
        const int tasks = L->pf_ntiles * c.num_heads;
        hipLaunchKernelGGL(k_attn_pf2, dim3((tasks + 3) / 4), dim3(256), 0, st, a, (const PfTile*)L->pf_tiles, L->pf_ntiles);
      } else if (c.kv_dtype) hipLaunchKernelGGL((k_attn_pf<1>), ag, dim3(kAttnWaves * 64), 0, st, a);
      else hipLaunchKernelGGL((k_attn_pf<0>), ag, dim3(kAttnWaves * 64), 0, st, a);
      SMI_LAUNCH_CHECK();
    }
    // o_proj
    GemmP o;
    memset(&o, 0, sizeof(o));
    o.M = M; o.rows = rows; o.eps = c.rms_eps; o.sspart = L->bss; o.npart = np_from_d;
    o.W = (const uint4*)sec(L, SMI_LLM_WO, l); o.NT = L->NTh; o.KT = L->KTq; o.XS = L->bxs_attn; o.Y = L->bh;
    o.XSout = L->bxs_h; o.gamma_next = (const float*)sec(L, SMI_LLM_LN2, l); o.ssout = L->bss;
    o.kseg = kseg_o;
    if (go) rc = launch_oproj(L, o, M, st);
    else if (L->tune2 & 16384) rc = launch_pgemm<PRO_PLAIN, EPI_RESID>(L, o, st);
    else if (few) rc = launch_pgemm<PRO_PLAIN, EPI_RESID, 2, 2, 6, 1, 0, 4>(L, o, st, nseg_o);
    else rc = launch_pgemm<PRO_PLAIN, EPI_RESID, 4, 2, 4, 1>(L, o, st);
    if (rc) return rc;
    // gate_up
    GemmP g;
    memset(&g, 0, sizeof(g));
    g.M = M; g.rows = rows; g.eps = c.rms_eps; g.sspart = L->bss; g.npart = np_from_o;
    g.W = (const uint4*)sec(L, SMI_LLM_WGU, l); g.NT = L->NTgu; g.KT = L->KTh; g.XS = L->bxs_h; g.XSout = L->bxs_act;
    if (gg) rc = launch_gemm<1, 8, 4, 1, PRO_NORM, EPI_SWIGLU, 1, 2>(L, g, st);
    else if (L->tune2 & 512) rc = launch_pgemm<PRO_NORM, EPI_SWIGLU>(L, g, st);
    else if (L->tune2 & 16384) rc = launch_pgemm<PRO_NORM, EPI_SWIGLU, 4>(L, g, st);
    else if (L->tune2 & 32768) rc = launch_pgemm<PRO_NORM, EPI_SWIGLU, 8, 2, 3>(L, g, st);
    else if (few) rc = launch_pgemm<PRO_NORM, EPI_SWIGLU, 2, 2, 4, 0, 0, 4>(L, g, st);   // 64 KiB ring: two blocks per CU, the 304 blocks of a 127-row prompt in one round (ring of 6: 1.2 rounds, 30 us)
    else rc = launch_pgemm<PRO_NORM, EPI_SWIGLU, 8, 2, 2, 0, 1>(L, g, st);
    if (rc) return rc;
    // down
    GemmP d;
    memset(&d, 0, sizeof(d));
    d.M = M; d.rows = rows; d.eps = c.rms_eps; d.sspart = L->bss; d.npart = np_from_o;
    d.W = (const uint4*)sec(L, SMI_LLM_WD, l); d.NT = L->NTh; d.KT = L->KTi; d.XS = L->bxs_act; d.Y = L->bh; d.wperm = L->wd_parts;
    d.XSout = L->bxs_h; d.ssout = L->bss; d.gamma_next = (const float*)sec(L, SMI_LLM_LN1, l + 1);
    d.kseg = kseg_d;
    if (gd) rc = launch_gemm<1, 16, 2, 5, PRO_PLAIN, EPI_RESID>(L, d, st);
    else if (L->tune2 & 16384) rc = launch_pgemm<PRO_PLAIN, EPI_RESID>(L, d, st);
    else if (few) rc = launch_pgemm<PRO_PLAIN, EPI_RESID, 2, 2, 6, 1, 0, 4>(L, d, st, nseg_d);
    else rc = launch_pgemm<PRO_PLAIN, EPI_RESID, 4, 2, 4, 1>(L, d, st);
    if (rc) return rc;
  }
  return SMI_OK;
}

int ensure_big(smi_llm* L, int rows) {
  if (rows <= L->big_rows) return SMI_OK;
  void* old[] = {L->bh, L->bq, L->bxs_h, L->bxs_attn, L->bxs_act, L->bss, L->pslab};
  for (void* q : old)
    if (q) (void)hipFree(q);
  L->bh = L->bq = nullptr; L->bxs_h = L->bxs_attn = L->bxs_act = nullptr; L->bss = nullptr; L->big_rows = 0;
  L->pslab = nullptr; L->pslab_bytes = 0;
  const size_t R = (size_t)rows;
  {   // split-K partial sums: up to kPgSegD segments x NTh tiles x (min(rows, kPgSplitRows) / 16) m-tiles x 1 KiB
    const size_t mt = ((size_t)(rows < kPgSplitRows ? rows : kPgSplitRows) + 15) / 16;
    const size_t bytes = (size_t)16 * L->NTh * mt * 1024;
    if (hipMalloc((void**)&L->pslab, bytes) != hipSuccess) { smi_set_error("hipMalloc(prefill split-K slab, %zu bytes) failed", bytes); return SMI_ENOMEM; }
    L->pslab_bytes = bytes;
  }
  if (hipMalloc((void**)&L->bh, R * L->H * 4) != hipSuccess || hipMalloc((void**)&L->bq, R * L->Q * 4) != hipSuccess ||
      hipMalloc((void**)&L->bxs_h, R * L->H * 6) != hipSuccess || hipMalloc((void**)&L->bxs_attn, R * L->Q * 6) != hipSuccess ||
      hipMalloc((void**)&L->bxs_act, R * L->I * 6) != hipSuccess || hipMalloc((void**)&L->bss, R * L->NTh * 4 * 4) != hipSuccess) {
    smi_set_error("hipMalloc(prefill workspace for %d rows) failed", rows);
    return SMI_ENOMEM;
  }
  L->big_rows = rows;
  return SMI_OK;
}

int launch_embed(smi_llm* L, const RowDesc* rows, int M, hipStream_t st) {
  hipLaunchKernelGGL(k_embed, dim3((M + 3) / 4), dim3(256), 0, st, (const uint16_t*)sec(L, SMI_LLM_LM_HEAD, 0), L->KTh, rows, M,
                     (const float*)sec(L, SMI_LLM_LN1, 0), L->h, L->xs_h, L->sspart, L->NTh * 4, L->exact);
  SMI_LAUNCH_CHECK();
  return SMI_OK;
}

// All layers for M rows.  kv_only_last: skip everything after the last layer's KV append
// (prefill rows whose hidden state is never read).
int launch_layers(smi_llm* L, const RowDesc* rows, int M, bool kv_only_last, hipStream_t st) {
  int rc;
  for (int l = 0; l < L->cfg.num_layers; ++l) {
    if ((rc = launch_one(L, KQKV, l, rows, M, nullptr, st))) return rc;
    if (kv_only_last && l == L->cfg.num_layers - 1) break;
    if ((rc = launch_one(L, KATTN, l, rows, M, nullptr, st))) return rc;
    if ((rc = launch_one(L, KO, l, rows, M, nullptr, st))) return rc;
    if ((rc = launch_one(L, KGU, l, rows, M, nullptr, st))) return rc;
    if ((rc = launch_one(L, KD, l, rows, M, nullptr, st))) return rc;
  }
  return SMI_OK;
}

int launch_step(smi_llm* L, int M, hipStream_t st) {
  int rc;
  if (eng_usable(L, L->rows, M)) {   // one live row: all layers in one persistent launch (smi_eng.h), same bits
    if ((rc = eng_launch(L, st))) return rc;
  } else if ((rc = launch_layers(L, L->rows, M, false, st))) return rc;
  if ((rc = launch_one(L, KLM, 0, L->rows, M, nullptr, st))) return rc;
  return launch_one(L, KFIN, 0, L->rows, M, nullptr, st);
}

int ensure_plan(smi_llm* L, size_t rows) {
  if (rows <= L->plan_cap) return SMI_OK;
  if (L->plan) (void)hipFree(L->plan);
  L->plan = nullptr; L->plan_cap = 0;
  size_t cap = rows + 1024;
  if (hipMalloc((void**)&L->plan, cap * sizeof(RowDesc)) != hipSuccess) {
    smi_set_error("hipMalloc(prefill plan, %zu rows) failed", cap);
    return SMI_ENOMEM;
  }
  L->plan_cap = cap;
  return SMI_OK;
}

}  // namespace

extern "C" {

size_t smi_llm_arena_bytes(const smi_llm_cfg* cfg) {
  if (!cfg_ok(cfg)) return 0;
  return make_layout(cfg).total;
}

int smi_llm_arena_section(const smi_llm_cfg* cfg, int section, int layer, size_t* offset, size_t* bytes) {
  SMI_REQUIRE(cfg_ok(cfg), "smi_llm_arena_section: invalid config");
  SMI_REQUIRE(section >= 0 && section < SMI_LLM_NUM_SECTIONS, "bad section %d", section);
  Layout L = make_layout(cfg);
  if (section >= SMI_LLM_FINAL_NORM) {
    SMI_REQUIRE(layer == 0, "global section takes layer 0");
    if (offset) *offset = L.off[section];
  } else {
    SMI_REQUIRE(layer >= 0 && layer < cfg->num_layers, "bad layer %d", layer);
    if (offset) *offset = L.layers_base + (size_t)layer * L.layer_stride + L.off[section];
  }
  if (bytes) *bytes = L.bytes[section];
  return SMI_OK;
}

int smi_llm_create(const smi_llm_cfg* cfg, const void* arena_dev, size_t arena_bytes, smi_llm** out) {
  SMI_REQUIRE(out, "smi_llm_create: out is null");
  *out = nullptr;
  SMI_REQUIRE(cfg_ok(cfg), "smi_llm_create: config outside the kernel contract (head_dim 64, hidden/intermediate %% 32, slots<=32)");
  Layout lay = make_layout(cfg);
  SMI_REQUIRE(arena_dev && arena_bytes >= lay.total, "smi_llm_create: arena too small (%zu < %zu)", arena_bytes, lay.total);
  SMI_REQUIRE(((uintptr_t)arena_dev & 255) == 0, "smi_llm_create: arena must be 256-byte aligned");
  {   // the arena says how it was packed (include/sparkmi.h: smi_llm_arena_tag); a mismatch with the config is an error, not wrong logits
    static_assert(sizeof(smi_llm_arena_tag) == 256, "arena tag is one 256-byte section");
    smi_llm_arena_tag tag;
    SMI_HIP(hipMemcpy(&tag, (const unsigned char*)arena_dev + lay.off[SMI_LLM_TAG], sizeof(tag), hipMemcpyDeviceToHost));
    SMI_REQUIRE(memcmp(tag.magic, "SMIARENA", 8) == 0, "smi_llm_create: the arena carries no layout tag (section SMI_LLM_TAG): not packed for this library, or for other dimensions");
    SMI_REQUIRE(tag.abi_version == SMI_ABI_VERSION, "smi_llm_create: arena packed for ABI %d, this library is ABI %d", tag.abi_version, SMI_ABI_VERSION);
    SMI_REQUIRE(tag.weights_exact == cfg->weights_exact, "smi_llm_create: the arena holds %s matrices, the config asks for %s (smi_llm_cfg.weights_exact)",
                tag.weights_exact ? "fp32 (exact-weights)" : "bf16-tiled", cfg->weights_exact ? "fp32" : "bf16 tiles");
    SMI_REQUIRE(tag.wd_plain == cfg->wd_plain, "smi_llm_create: the arena's W_down tiles are packed %s, the config says %s (smi_llm_cfg.wd_plain)",
                tag.wd_plain ? "in the plain tile order" : "row-part-major", cfg->wd_plain ? "plain" : "row-part-major");
    SMI_REQUIRE(tag.vocab_size == cfg->vocab_size && tag.hidden_size == cfg->hidden_size && tag.num_layers == cfg->num_layers &&
                tag.num_heads == cfg->num_heads && tag.num_kv_heads == cfg->num_kv_heads && tag.intermediate_size == cfg->intermediate_size &&
                tag.max_positions == cfg->max_positions, "smi_llm_create: the arena was packed for other dimensions than this config");
  }
  smi_llm* L = new smi_llm();
  L->cfg = *cfg; L->lay = lay; L->arena = (const unsigned char*)arena_dev;
  L->H = cfg->hidden_size; L->Q = cfg->num_heads * cfg->head_dim; L->KV = cfg->num_kv_heads * cfg->head_dim;
  L->I = cfg->intermediate_size;
  L->KTh = L->H / 32; L->KTq = L->Q / 32; L->KTi = L->I / 32;
  L->NTqkv = (L->Q + 2 * L->KV) / 16; L->NTh = L->H / 16; L->NTgu = 2 * L->I / 16; L->NTlm = lay.vpad / 16;
  L->lm_cap = (L->NTlm + 3) / 4;                       // partial-argmax slots (two-m-tile path: one per block)
  L->lm_blocks = L->lm_cap < 512 ? L->lm_cap : 512;    // persistent path: 2 resident blocks per CU
  L->max_steps = cfg->max_positions;
  L->do_sample = 0; L->top_k = 50; L->temperature = 0.8f; L->top_p = 0.95f; L->seed = 0; L->logits = nullptr; L->tok = nullptr; L->cand_v = nullptr; L->cand_i = nullptr; L->cand_n = nullptr; L->stamps = nullptr; L->stamps_on = 0;
  { L->tune[0] = L->tune[1] = L->tune[2] = L->tune[3] = 0; const char* e = smi_env("SPARKMI_TUNE"); if (e) sscanf(e, "%d,%d,%d,%d", &L->tune[0], &L->tune[1], &L->tune[2], &L->tune[3]); }
  L->bh = L->bq = nullptr; L->bxs_h = L->bxs_attn = L->bxs_act = nullptr; L->bss = nullptr; L->big_rows = 0;
  L->pslab = nullptr; L->pslab_bytes = 0;
  // helper-block prefetch per producer: bit 0 QKV (gate_up's first half), bit 1 attention (second half), bit 2 down_proj (the next
  // layer's QKV / o_proj); SPARKMI_PREFETCH=<mask> picks, SPARKMI_NO_PREFETCH=1 is mask 0
  // Default since k_down1 (round 3): QKV's helpers at ONE row only.  Measured on one box, alternating processes, graph step at one
  // row: mask 0 566-571 us, 1 563-564, 2 581, 3 583, 4 597, 5 602, 6 658, 7 665 (down_proj's 256-thread blocks make slow helpers);
  // at 4 rows mask 0 727 us, 3 748, 7 739-745 (profiles/r03_prefetch.txt).  SPARKMI_PREFETCH=<mask> applies up to 8 rows as before.
  L->prefetch_mask = 1; L->prefetch_rows = 1;
  if (const char* e = smi_env("SPARKMI_PREFETCH")) { L->prefetch_mask = atoi(e) & 7; L->prefetch_rows = 8; }
  if (smi_env("SPARKMI_NO_PREFETCH")) L->prefetch_mask = 0;
  { const char* e = smi_env("SPARKMI_GRAPH_STEPS"); L->graph_steps = e ? atoi(e) : 8; if (L->graph_steps < 1 || L->graph_steps > 32) L->graph_steps = 8; }
  { const char* e = smi_env("SPARKMI_PF_INLINE"); L->pf_inline = !(e && e[0] == '0'); }
  { const char* e = smi_env("SPARKMI_PF_QKV"); L->pf_qkv_eighths = e ? atoi(e) : 2; if (L->pf_qkv_eighths < 0 || L->pf_qkv_eighths > 8) L->pf_qkv_eighths = 2; }
  L->part_o = nullptr; L->h2 = nullptr; L->dpart = nullptr; L->fuse_cnt = nullptr;
  { const char* e = smi_env("SPARKMI_FUSE1_LAST"); L->fuse1_last = e && e[0] == '1'; }
  // OFF by default (diagnostics build: SPARKMI_FUSE2_ROWS=n enables it up to n <= 8 rows).  Measured at 0.5B, graph step, one box
  // (profiles/r04_fused_oproj_last_arriver.txt): 8 rows 730 -> 863 us (attention 3.9 -> 14.1 us for the 4.3-us o_proj launch it
  // replaces), 2 rows 616 -> 640, one row (SPARKMI_FUSE1_LAST=1, against the per-head partials every gate_up block sums) 543 -> 581:
  // the arrival count + the reads of write-through partials cost 3-4 us at one row, where a kernel boundary costs 1.8.
  { const char* e = smi_env("SPARKMI_FUSE2_ROWS"); L->fuse2_rows = e ? atoi(e) : 0; if (L->fuse2_rows > kFuse2Max) L->fuse2_rows = kFuse2Max; }
  // rows from which down_proj runs chain-split (k_downC): graph step at 0.5B, same box (profiles/r04_batch_ab.txt): 4 rows 641 (k_downS) vs
  // 692 us (chains), 8 rows 728 vs 713, 16 rows 790 (k_gemm<.., H = 4>) vs 744, 32 rows 892 vs 859
  // (5 / 6 rows: 731 / 728 us with the chains, the same as k_downS within noise -- the chains take over from 7)
  { const char* e = smi_env("SPARKMI_DC_MIN"); L->dc_min = e ? atoi(e) : 7; }
  L->fuse_o = !smi_env("SPARKMI_NO_FUSE_O") && (cfg->num_heads == 14 || cfg->num_heads == 4) && cfg->num_heads <= kMaxOHeads &&
              L->NTh % kFuseQB == 0 && L->NTh / kFuseQB <= kAttnWaves * kFuseOT && L->KTh * 8 <= 256;
  { const char* e = smi_env("SPARKMI_TUNE2"); L->tune2 = e ? atoi(e) : 0; }
  { const char* e = smi_env("SPARKMI_PGEMM_MIN_ROWS"); L->pgemm_min_rows = e ? atoi(e) : 1280; }
  {
    const char* names[4] = {"SPARKMI_PGEMM_MIN_QKV", "SPARKMI_PGEMM_MIN_O", "SPARKMI_PGEMM_MIN_GU", "SPARKMI_PGEMM_MIN_D"};
    // measured crossovers (tools/prefill_time.py mix / mix2, profiles/README.md): gate_up from ~300 rows (its grouped form re-reads
    // the operand triples once per 32 columns), down_proj from ~900, the two short-K GEMMs from ~1300
    const int dflt[4] = {1280, 1280, 288, 896};
    const bool common = smi_env("SPARKMI_PGEMM_MIN_ROWS") != nullptr;
    L->pg_forced = common;
    for (int i = 0; i < 4; ++i) { const char* e = smi_env(names[i]); L->pg_forced |= e != nullptr; L->pg_min[i] = e ? atoi(e) : common ? L->pgemm_min_rows : dflt[i]; }
  }
  // A sequence's prompt rows take the kernel family ITS OWN length selects (never the call's total): up to kMaxRows rows the
  // decode kernels in chunks, from pf_long rows the prefill-GEMM family (k_pgemm + k_attn_pf2), in between (empty by default)
  // the row-grouped decode GEMMs -- so its K/V rows, and with a bf16 cache its tokens, do not depend on what else is in the call.
  { const char* e = smi_env("SPARKMI_PG_SPLIT_ROWS"); L->pg_split_rows = e ? atoi(e) : kPgSplitRows; if (L->pg_split_rows > kPgSplitRows) L->pg_split_rows = kPgSplitRows; }
  { const char* e = smi_env("SPARKMI_PF_LONG"); L->pf_long = e ? atoi(e) : kMaxRows + 1; if (L->pf_long < kMaxRows + 1) L->pf_long = kMaxRows + 1; }
  { const char* e = smi_env("SPARKMI_GU1_ROWS"); L->gu1_rows = e ? atoi(e) : 32; }
  { const char* e = smi_env("SPARKMI_GU1_LO"); L->gu1_lo = e ? atoi(e) : 4; }
  L->wd_parts = cfg->wd_plain ? 0 : 1;   // the arena's W_down tile order comes with its config (never from the environment)
  L->exact = cfg->weights_exact;
  if (L->exact) L->fuse_o = 0;           // (the fused o_proj reads bf16 tiles)
  L->pf_tiles = nullptr; L->pf_tiles_cap = 0; L->pf_ntiles = 0;
  { const char* e = smi_env("SPARKMI_ATTN_PF2"); L->attn_pf2 = !(e && e[0] == '0'); }
  L->graph = nullptr; L->graph_B = 0; L->graph_seg = 1; L->graph_ident = 1; L->plan = nullptr; L->plan_cap = 0; L->B = 0; L->started = 0; L->ctl = nullptr; L->admit_seq = 0; memset(&L->hctl, 0, sizeof(L->hctl));
  L->session = 0; L->identity_slots = 1; L->attn_seg = 1; L->apart = nullptr; L->apart_floats = 0; memset(L->slot_busy, 0, sizeof(L->slot_busy)); memset(L->slot_len, 0, sizeof(L->slot_len));
  const size_t esz = cfg->kv_dtype ? 4 : 2;
  L->paged = cfg->kv_page_tokens > 0; L->pshift = 0; L->ppslot = 0; L->ptab = nullptr;
  memset(L->slot_pages, 0, sizeof(L->slot_pages)); memset(L->plen, 0, sizeof(L->plen));
  L->kv_layer_elems = (size_t)cfg->max_slots * cfg->num_kv_heads * cfg->max_positions * kHeadDim;
  if (L->paged) {
    while ((1 << L->pshift) < cfg->kv_page_tokens) ++L->pshift;
    L->ppslot = cfg->max_positions >> L->pshift;
    L->kv_layer_elems = (size_t)cfg->kv_pages * cfg->num_kv_heads * cfg->kv_page_tokens * kHeadDim;
    L->hptab.assign((size_t)cfg->max_slots * L->ppslot, 0);
    for (int pg = cfg->kv_pages - 1; pg >= 0; --pg) L->free_pages.push_back(pg);
  }
  const size_t kvbytes = L->kv_layer_elems * esz * cfg->num_layers;
#define SMI_ALLOC(ptr, bytes)                                                      \
  if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) {                          \
    smi_set_error("hipMalloc(%s, %zu bytes) failed", #ptr, (size_t)(bytes));       \
    smi_llm_destroy(L);                                                            \
    return SMI_ENOMEM;                                                             \
  }
  SMI_ALLOC(L->h, (size_t)kMaxRows * L->H * 4);
  SMI_ALLOC(L->qbuf, (size_t)kMaxRows * L->Q * 4);
  SMI_ALLOC(L->xs_h, (size_t)kMaxRows * L->H * 6);
  SMI_ALLOC(L->xs_attn, (size_t)kMaxRows * L->Q * 6);
  SMI_ALLOC(L->xs_act, (size_t)kMaxRows * L->I * 6);
  SMI_ALLOC(L->sspart, (size_t)kMaxRows * L->NTh * 4 * 4);
  SMI_ALLOC(L->part_o, (size_t)kFuse2Max * kMaxOHeads * L->H * 4);
  SMI_ALLOC(L->fuse_cnt, (size_t)kFuse2Max * kFuseQB * 4);
  SMI_HIP(hipMemset(L->fuse_cnt, 0, (size_t)kFuse2Max * kFuseQB * 4));
  SMI_ALLOC(L->h2, (size_t)L->H * 4);
  SMI_ALLOC(L->dpart, (size_t)16 * L->NTh * 4 * 64 * 16);
  SMI_ALLOC(L->rows, kMaxRows * sizeof(RowDesc));
  SMI_ALLOC(L->pval, (size_t)L->lm_cap * kMaxRows * 4);
  SMI_ALLOC(L->pidx, (size_t)L->lm_cap * kMaxRows * 4);
  SMI_ALLOC(L->hist, (size_t)L->max_steps * kMaxRows * 8);
  SMI_ALLOC(L->count, kMaxRows * 4);
  SMI_ALLOC(L->finished, kMaxRows * 4);
  SMI_ALLOC(L->step, 4);
  SMI_ALLOC(L->ctl, sizeof(Ctl));
  SMI_HIP(hipMemset(L->ctl, 0, sizeof(Ctl)));
  if (L->paged) {
    SMI_ALLOC(L->ptab, L->hptab.size() * 4);
    SMI_HIP(hipMemset(L->ptab, 0, L->hptab.size() * 4));
  }
  SMI_ALLOC(L->logits, (size_t)kMaxRows * cfg->vocab_size * 4);
  SMI_ALLOC(L->tok, kMaxRows * 4);
  SMI_ALLOC(L->cand_v, (size_t)kMaxRows * kCandCap * 4);
  SMI_ALLOC(L->cand_i, (size_t)kMaxRows * kCandCap * 4);
  SMI_ALLOC(L->cand_n, kMaxRows * 4);
  SMI_HIP(hipMemset(L->cand_n, 0, kMaxRows * 4));
  SMI_ALLOC(L->stamps, (size_t)4096 * 8 * 8);
  SMI_ALLOC(L->kcache, kvbytes);
  SMI_ALLOC(L->vcache, kvbytes);
#undef SMI_ALLOC
  if (hipMemset(L->kcache, 0, kvbytes) != hipSuccess || hipMemset(L->vcache, 0, kvbytes) != hipSuccess ||
      hipMemset(L->h, 0, (size_t)kMaxRows * L->H * 4) != hipSuccess ||
      hipMemset(L->xs_h, 0, (size_t)kMaxRows * L->H * 6) != hipSuccess ||
      hipMemset(L->xs_attn, 0, (size_t)kMaxRows * L->Q * 6) != hipSuccess ||
      hipMemset(L->xs_act, 0, (size_t)kMaxRows * L->I * 6) != hipSuccess ||
      hipMemset(L->sspart, 0, (size_t)kMaxRows * L->NTh * 4 * 4) != hipSuccess ||
      hipMemset(L->qbuf, 0, (size_t)kMaxRows * L->Q * 4) != hipSuccess ||
      hipMemset(L->rows, 0, kMaxRows * sizeof(RowDesc)) != hipSuccess ||
      hipMemset(L->count, 0, kMaxRows * 4) != hipSuccess || hipMemset(L->finished, 0, kMaxRows * 4) != hipSuccess ||
      hipMemset(L->step, 0, 4) != hipSuccess) {
    smi_set_error("hipMemset of scratch failed");
    smi_llm_destroy(L);
    return SMI_EHIP;
  }
  // k_lm32 stages 16 rows' operand triples in LDS (86 KB at K = 896): opt in to the large dynamic window here, once per
  // handle (= per device), outside any stream capture
  if (hipFuncSetAttribute((const void*)k_lm32<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
  if (hipFuncSetAttribute((const void*)k_lm32<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
  if (hipEventCreate(&L->ev0) != hipSuccess || hipEventCreate(&L->ev1) != hipSuccess) {
    smi_set_error("hipEventCreate failed");
    smi_llm_destroy(L);
    return SMI_EHIP;
  }
  // The one-row decode engine is opt-in (SPARKMI_ENGINE=1, or smi_llm_set_engine later: it is built on first request --
  // 0.8 GB of re-packed weights): measured on MI355X at the 0.5B shape a layer takes 26.3 us in the engine against 23.4 us
  // as four launches -- five in-launch hand-offs of 2.5-4.5 us cost more than the four kernel boundaries they replace
  // (profiles/r03_engine_ab.txt).
#ifdef SMI_DIAG
  L->eng_on = 0;
  snprintf(L->eng.why, sizeof(L->eng.why), "off (opt-in: SPARKMI_ENGINE=1 or smi_llm_set_engine)");
  { const char* e = smi_env("SPARKMI_ENGINE");
    if (e && e[0] == '1') {
      const int rce = eng_create(L);
      if (rce) { smi_llm_destroy(L); return rce; }
      L->eng_on = L->eng.enabled;
    }
  }
#endif
  *out = L;
  return SMI_OK;
}

int smi_llm_destroy(smi_llm* L) {
  if (!L) return SMI_OK;
  graphs_flush(L);
  eng_destroy(L);
  void* ptrs[] = {L->h, L->qbuf, L->xs_h, L->xs_attn, L->xs_act, L->sspart, L->part_o, L->h2, L->dpart, L->fuse_cnt, L->rows, L->plan, L->pf_tiles, L->pval, L->pidx, L->hist,
                  L->count, L->finished, L->step, L->ctl, L->ptab, L->kcache, L->vcache, L->logits, L->tok, L->cand_v, L->cand_i, L->cand_n, L->stamps, L->bh, L->bq, L->bxs_h, L->bxs_attn, L->bxs_act, L->bss, L->pslab, L->apart};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  if (L->ev0) (void)hipEventDestroy(L->ev0);
  if (L->ev1) (void)hipEventDestroy(L->ev1);
  delete L;
  return SMI_OK;
}

int smi_llm_set_sampling(smi_llm* L, int do_sample, float temperature, int top_k, float top_p, uint64_t seed) {
  SMI_REQUIRE(L, "smi_llm_set_sampling: null handle");
  if (do_sample) {
    SMI_REQUIRE(temperature > 0.f, "smi_llm_set_sampling: temperature must be > 0");
    SMI_REQUIRE(top_k >= 1 && top_k <= kSampleCap, "smi_llm_set_sampling: top_k must be in 1..%d", kSampleCap);
    SMI_REQUIRE(top_p > 0.f && top_p <= 1.f, "smi_llm_set_sampling: top_p must be in (0, 1]");
    if (top_k > L->cfg.vocab_size) top_k = L->cfg.vocab_size;
  }
  // mode, temperature, top-k and top-p are kernel parameters of the captured step; the seed lives in device memory (Ctl)
  const bool changed = L->do_sample != (do_sample ? 1 : 0) ||
                       (do_sample && (L->temperature != temperature || L->top_k != top_k || L->top_p != top_p));
  L->do_sample = do_sample ? 1 : 0;
  L->temperature = temperature; L->top_k = top_k; L->top_p = top_p; L->seed = seed;
  L->hctl.seed = seed;   // uploaded with the next prefill / session_begin too
  SMI_HIP(hipMemcpy(&L->ctl->seed, &seed, sizeof(seed), hipMemcpyHostToDevice));
  if (changed) graphs_flush(L);
  return SMI_OK;
}

// Runs every prompt token but each sequence's last through the layers (K/V appended at slots[b]) and leaves the n
// "last prompt token" rows at L->plan + *tail_off (device) and in L->host_rows (host) for the first step.
// Argument checks of a prefill / admission, made BEFORE anything of the handle's state (KV pages, sequence numbers, the
// `started` flag) is touched: a call that fails here changes nothing.
static int validate_prompts(const smi_llm* L, const int64_t* ids, const int32_t* lens, int B, int P_max) {
  for (int b = 0; b < B; ++b) {
    SMI_REQUIRE(lens[b] >= 1 && lens[b] <= P_max, "smi_llm_prefill: lens[%d]=%d outside 1..%d", b, lens[b], P_max);
    SMI_REQUIRE(lens[b] < L->cfg.max_positions, "smi_llm_prefill: prompt %d longer than max_positions", b);
    for (int t = 0; t < lens[b]; ++t) {
      const int64_t id = ids[(size_t)b * P_max + t];
      SMI_REQUIRE(id >= 0 && id < L->cfg.vocab_size, "smi_llm_prefill: token id %lld out of range", (long long)id);
    }
  }
  return SMI_OK;
}

static int prefill_prompts(smi_llm* L, const int64_t* ids, const int32_t* lens, int B, int P_max, const int32_t* slots,
                           size_t* tail_off, hipStream_t st) {
  size_t total = 0;
  { const int rcv = validate_prompts(L, ids, lens, B, P_max); if (rcv) return rcv; }
  for (int b = 0; b < B; ++b) total += lens[b] - 1;
  {
    int longest = 0;
    for (int b = 0; b < B; ++b) longest = lens[b] > longest ? lens[b] : longest;
    L->attn_seg = segs_for(longest);
  }
  // Every prompt token but each sequence's last is a row (slot, pos, token); the B last tokens run as the first regular step
  // (lm_head + argmax).  WHICH kernels a sequence's prompt rows run through is decided by ITS OWN row count alone:
  //   class 0  up to kMaxRows rows        the decode kernels, kMaxRows-row chunks (rows are independent there: any chunking, same bits)
  //   class 1  kMaxRows + 1 .. pf_long-1  row-grouped decode GEMMs + the prefill attention (empty by default: pf_long = kMaxRows + 1)
  //   class 2  pf_long rows and more      the prefill GEMM family (k_pgemm, segmented o_proj / down_proj sums) + the prefill attention
  // and every class is its own pass over the layers, so a sequence's K/V rows -- and, K/V being rounded to bf16, its tokens --
  // are the same bits whatever else the call holds (round 3 picked the kernels from the call's TOTAL rows).
  // Diagnostics keep the old single pass: SPARKMI_PREFILL_CHUNKS (chunks only), SPARKMI_PGEMM_MIN_* (per kernel by total rows);
  // the exact-weights mode always takes chunks.
  const bool chunks_only = smi_env("SPARKMI_PREFILL_CHUNKS") != nullptr || L->exact;
  auto cls_of = [&](int b) -> int {
    const int r = lens[b] - 1;
    if (chunks_only) return 0;
    if (L->pg_forced) return total > (size_t)kMaxRows ? 3 : 0;   // 3: one pass, kernels by the pass's row count (launch_layers_big)
    return r <= kMaxRows ? 0 : r < L->pf_long ? 1 : 2;
  };
  size_t ncls[4] = {0, 0, 0, 0};
  for (int b = 0; b < B; ++b) ncls[cls_of(b)] += lens[b] - 1;
  const size_t nchunks = (ncls[0] + kMaxRows - 1) / kMaxRows;
  // plan: [class 2 rows][class 1 rows][class 3 rows][class 0 rows, padded to whole chunks][the B last-token rows]
  const size_t off2 = 0, off1 = ncls[2], off3 = off1 + ncls[1], off0 = off3 + ncls[3], tail = off0 + nchunks * kMaxRows;
  int rc;
  if ((rc = ensure_plan(L, tail + kMaxRows))) return rc;
  L->host_rows.assign(tail + kMaxRows, RowDesc{0, 0, 0, 0});
  {
    size_t w[4] = {off0, off1, off2, off3};
    for (int b = 0; b < B; ++b) {
      size_t& r = w[cls_of(b)];
      for (int t = 0; t + 1 < lens[b]; ++t) L->host_rows[r++] = RowDesc{slots[b], t, (int32_t)ids[(size_t)b * P_max + t], 0};
    }
  }
  for (int b = 0; b < B; ++b)
    L->host_rows[tail + b] = RowDesc{slots[b], lens[b] - 1, (int32_t)ids[(size_t)b * P_max + lens[b] - 1], 0};   // flags = tokens emitted
  SMI_HIP(hipMemcpyAsync(L->plan, L->host_rows.data(), L->host_rows.size() * sizeof(RowDesc), hipMemcpyHostToDevice, st));
  // measured (tools/prefill_time.py, profiles/README.md): 32-row chunks ~1.4 ms each; row-grouped decode GEMMs
  // ~1.5 ms + 9 us/row; the prefill GEMM ~15 ms + 5 us/row (crossover near 3000 rows)
  const size_t pass_off[3] = {off2, off1, off3}, pass_rows[3] = {ncls[2], ncls[1], ncls[3]};
  const int pass_family[3] = {PF_PGEMM, PF_GROUPED, PF_PGEMM};   // (class 3: the family argument is overridden per kernel by pg_min)
  for (int ps = 0; ps < 3; ++ps) {
    if (pass_rows[ps] == 0) continue;
    // whole groups of up to kBigRows rows through the row-grouped decode GEMMs / the prefill GEMM (k_pgemm)
    constexpr size_t kBigRows = 4096;
    if ((rc = ensure_big(L, (int)(pass_rows[ps] < kBigRows ? pass_rows[ps] : kBigRows)))) return rc;
    for (size_t q0 = 0; q0 < pass_rows[ps]; q0 += kBigRows) {
      const size_t r0 = pass_off[ps] + q0;
      const int M = (int)((pass_rows[ps] - q0) < kBigRows ? (pass_rows[ps] - q0) : kBigRows);
      const RowDesc* rows = L->plan + r0;
      // prefill attention tiles of this row group: runs of up to 16 rows that are consecutive tokens of one KV slot
      L->pf_ntiles = 0;
      if (!L->cfg.kv_dtype && L->attn_pf2) {
        std::vector<PfTile>& T = L->host_tiles;
        T.clear();
        for (int i = 0; i < M;) {
          const RowDesc& a0 = L->host_rows[r0 + i];
          int n = 1;
          while (n < 16 && i + n < M && L->host_rows[r0 + i + n].slot == a0.slot && L->host_rows[r0 + i + n].pos == a0.pos + n) ++n;
          T.push_back(PfTile{i, n, a0.slot, a0.pos});
          i += n;
        }
        if (T.size() > L->pf_tiles_cap) {
          if (L->pf_tiles) (void)hipFree(L->pf_tiles);
          L->pf_tiles = nullptr; L->pf_tiles_cap = 0;
          const size_t cap = T.size() + 256;
          if (hipMalloc((void**)&L->pf_tiles, cap * sizeof(PfTile)) != hipSuccess) { smi_set_error("hipMalloc(prefill attention tiles) failed"); return SMI_ENOMEM; }
          L->pf_tiles_cap = cap;
        }
        // (pageable source: staged before the call returns; host_tiles is rebuilt only by the next group, behind this copy)
        SMI_HIP(hipMemcpyAsync(L->pf_tiles, T.data(), T.size() * sizeof(PfTile), hipMemcpyHostToDevice, st));
        L->pf_ntiles = (int)T.size();
      }
      const bool d_grouped = L->pg_forced ? M < L->pg_min[3] : pass_family[ps] == PF_GROUPED;   // who leaves the RMSNorm partials of the layer input
      hipLaunchKernelGGL(k_embed, dim3((M + 3) / 4), dim3(256), 0, st, (const uint16_t*)sec(L, SMI_LLM_LM_HEAD, 0), L->KTh, rows, M,
                         (const float*)sec(L, SMI_LLM_LN1, 0), L->bh, L->bxs_h, L->bss, d_grouped ? L->NTh * 4 : L->NTh, 0);
      SMI_LAUNCH_CHECK();
      if ((rc = launch_layers_big(L, rows, M, pass_family[ps], st))) return rc;
    }
  }
  for (size_t c = 0; c < nchunks; ++c) {
    const int M = (int)((ncls[0] - c * kMaxRows) < (size_t)kMaxRows ? (ncls[0] - c * kMaxRows) : kMaxRows);
    const RowDesc* rows = L->plan + off0 + c * kMaxRows;
    if ((rc = launch_embed(L, rows, M, st))) return rc;
    if ((rc = launch_layers(L, rows, M, true, st))) return rc;
  }
  *tail_off = tail;
  return SMI_OK;
}

// eos ids + seed + the slots' sequence numbers -> device (before anything of the generation is enqueued)
static int upload_ctl(smi_llm* L, const int64_t* eos_ids, int n_eos, hipStream_t st) {
  SMI_REQUIRE(n_eos >= 0 && n_eos <= SMI_MAX_EOS && (n_eos == 0 || eos_ids), "eos list: 0..%d ids", SMI_MAX_EOS);
  for (int e = 0; e < SMI_MAX_EOS; ++e) L->hctl.eos[e] = e < n_eos ? (long long)eos_ids[e] : -1;
  L->hctl.n_eos = n_eos;
  L->hctl.seed = L->seed;
  SMI_HIP(hipMemcpyAsync(L->ctl, &L->hctl, sizeof(Ctl), hipMemcpyHostToDevice, st));
  return SMI_OK;
}

int smi_llm_prefill(smi_llm* L, const int64_t* ids, const int32_t* lens, int B, int P_max, const int64_t* eos_ids, int n_eos,
                    void* stream) {
  SMI_REQUIRE(L && ids && lens, "smi_llm_prefill: null argument");
  SMI_REQUIRE(B >= 1 && B <= L->cfg.max_slots, "smi_llm_prefill: B=%d outside 1..%d", B, L->cfg.max_slots);
  hipStream_t st = (hipStream_t)stream;
  int32_t slots[kMaxRows];
  for (int b = 0; b < kMaxRows; ++b) slots[b] = b;
  { const int rcv = validate_prompts(L, ids, lens, B, P_max); if (rcv) return rcv; }   // nothing touched yet
  SMI_REQUIRE(n_eos >= 0 && n_eos <= SMI_MAX_EOS && (n_eos == 0 || eos_ids), "eos list: 0..%d ids", SMI_MAX_EOS);
  if (L->paged) {   // a new generation: every page back to the pool, then what the prompts need
    // the demand is checked against the WHOLE pool first (everything is about to be free), so a prompt set that cannot fit
    // fails here with the previous generation's pages, table and `started` flag untouched
    long need = 0;
    for (int b = 0; b < B; ++b) need += (lens[b] + (1 << L->pshift) - 1) >> L->pshift;
    if (need > L->cfg.kv_pages) {
      smi_set_error("KV page pool too small for these prompts: %ld pages of %d tokens needed, the pool has %d", need, 1 << L->pshift, L->cfg.kv_pages);
      return SMI_ENOMEM;
    }
    for (int b = 0; b < kMaxRows; ++b) pages_release(L, b);
    int rcp = pages_ensure(L, slots, lens, B, st);
    if (rcp) { L->started = 0; return rcp; }   // (cannot happen after the check above; the old generation is gone by now)
  }
  for (int b = 0; b < B; ++b) L->plen[b] = lens[b];
  for (int b = 0; b < kMaxRows; ++b) L->hctl.seqid[b] = b;
  L->admit_seq = B;
  { int rc0 = upload_ctl(L, eos_ids, n_eos, st); if (rc0) return rc0; }
  SMI_HIP(hipMemsetAsync(L->count, 0, kMaxRows * 4, st));
  SMI_HIP(hipMemsetAsync(L->finished, 0, kMaxRows * 4, st));
  SMI_HIP(hipMemsetAsync(L->step, 0, 4, st));
  L->B = B; L->started = 1; L->session = 0; L->identity_slots = 1;
  L->max_len = 0;
  for (int b = 0; b < B; ++b) L->max_len = lens[b] > L->max_len ? lens[b] : L->max_len;
  L->steps_launched = 1;
  // the captured decode step is kept across utterances: eos ids and the seed live in device memory (Ctl); smi_llm_decode
  // re-captures only when the row count, the context-segment count or the slot mapping differ from the captured ones
  size_t tail = 0;
  int rc;
  if ((rc = prefill_prompts(L, ids, lens, B, P_max, slots, &tail, st))) { L->started = 0; return rc; }
  if (hipMemcpyAsync(L->rows, L->plan + tail, kMaxRows * sizeof(RowDesc), hipMemcpyDeviceToDevice, st) != hipSuccess) {
    L->started = 0;
    smi_set_error("smi_llm_prefill: copying the first step's rows failed");
    return SMI_EHIP;
  }
  if ((rc = launch_embed(L, L->rows, B, st)) || (rc = launch_step(L, B, st))) { L->started = 0; return rc; }
  return SMI_OK;
}

// ---- continuous batching: sequences join (admit) and leave (retire) between decode steps ----
int smi_llm_session_begin(smi_llm* L, const int64_t* eos_ids, int n_eos, void* stream) {
  SMI_REQUIRE(L, "smi_llm_session_begin: null handle");
  hipStream_t st = (hipStream_t)stream;
  memset(L->hctl.seqid, 0, sizeof(L->hctl.seqid));
  L->admit_seq = 0;
  if (L->paged)
    for (int b = 0; b < kMaxRows; ++b) pages_release(L, b);
  { int rc0 = upload_ctl(L, eos_ids, n_eos, st); if (rc0) return rc0; }
  SMI_HIP(hipMemsetAsync(L->count, 0, kMaxRows * 4, st));
  SMI_HIP(hipMemsetAsync(L->finished, 0, kMaxRows * 4, st));
  SMI_HIP(hipMemsetAsync(L->step, 0, 4, st));
  L->B = 0; L->started = 1; L->session = 1; L->identity_slots = 1;
  L->live_order.clear();
  L->max_len = 0; L->steps_launched = 0;
  memset(L->slot_busy, 0, sizeof(L->slot_busy));
  memset(L->slot_len, 0, sizeof(L->slot_len));
  L->graph = nullptr;
  return SMI_OK;
}

// the live row set changed: refresh the device rows, the per-row residual / operand state, and drop the graph
static int session_set_rows(smi_llm* L, const std::vector<RowDesc>& live, hipStream_t st) {
  L->B = (int)live.size();
  L->identity_slots = 1;
  for (int b = 0; b < L->B; ++b) L->identity_slots &= live[b].slot == b;
  L->live_order.clear();
  for (int b = 0; b < L->B; ++b) L->live_order.push_back(live[b].slot);
  L->graph = nullptr;   // (the next decode picks the cached step of the new row count)
  if (L->B == 0) return SMI_OK;
  L->host_rows.assign(kMaxRows, RowDesc{0, 0, 0, 0});
  for (int b = 0; b < L->B; ++b) L->host_rows[b] = live[b];
  SMI_HIP(hipMemcpyAsync(L->rows, L->host_rows.data(), kMaxRows * sizeof(RowDesc), hipMemcpyHostToDevice, st));
  SMI_HIP(hipStreamSynchronize(st));    // host_rows is reused by the next call
  return launch_embed(L, L->rows, L->B, st);   // h = E[token], first-norm operand, sum of squares of every live row
}

static int session_live_rows(smi_llm* L, std::vector<RowDesc>& live, hipStream_t st) {
  live.assign((size_t)L->B, RowDesc{0, 0, 0, 0});
  SMI_HIP(hipStreamSynchronize(st));
  if (L->B) SMI_HIP(hipMemcpy(live.data(), L->rows, (size_t)L->B * sizeof(RowDesc), hipMemcpyDeviceToHost));
  return SMI_OK;
}

int smi_llm_admit(smi_llm* L, const int64_t* ids, const int32_t* lens, int n, int P_max, int32_t* slots_out, void* stream) {
  SMI_REQUIRE(L && ids && lens && slots_out, "smi_llm_admit: null argument");
  if (!L->started || !L->session) { smi_set_error("smi_llm_admit outside a session (smi_llm_session_begin first)"); return SMI_ESTATE; }
  SMI_REQUIRE(n >= 1 && L->B + n <= L->cfg.max_slots && L->B + n <= kMaxRows, "smi_llm_admit: %d new + %d live sequences exceed %d slots", n,
              L->B, L->cfg.max_slots < kMaxRows ? L->cfg.max_slots : kMaxRows);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  std::vector<RowDesc> live;
  if ((rc = session_live_rows(L, live, st))) return rc;
  int32_t slots[kMaxRows];
  int k = 0;
  for (int sl = 0; sl < L->cfg.max_slots && k < n; ++sl)
    if (!L->slot_busy[sl]) slots[k++] = sl;
  SMI_REQUIRE(k == n, "smi_llm_admit: no free KV slot");
  if ((rc = validate_prompts(L, ids, lens, n, P_max))) return rc;   // nothing touched yet
  if (L->paged && (rc = pages_ensure(L, slots, lens, n, st))) return rc;   // all or nothing: a short pool admits nobody
  const int seq0 = L->admit_seq;
  int32_t old_seqid[kMaxRows];
  for (int b = 0; b < n; ++b) { old_seqid[b] = L->hctl.seqid[slots[b]]; L->hctl.seqid[slots[b]] = L->admit_seq++; }
  auto undo = [&]() {   // nothing was admitted: sequence numbers and pages as before (the device copy is rewritten by the next admission)
    for (int b = 0; b < n; ++b) L->hctl.seqid[slots[b]] = old_seqid[b];
    L->admit_seq = seq0;
    if (L->paged)
      for (int b = 0; b < n; ++b) pages_release(L, slots[b]);
  };
  if (hipMemcpyAsync(L->ctl, &L->hctl, sizeof(Ctl), hipMemcpyHostToDevice, st) != hipSuccess) {
    undo();
    smi_set_error("smi_llm_admit: uploading the generation controls failed");
    return SMI_EHIP;
  }
  size_t tail = 0;
  if ((rc = prefill_prompts(L, ids, lens, n, P_max, slots, &tail, st))) { undo(); return rc; }
  // first token of the new sequences: one step over the new rows alone
  int32_t zeros[kMaxRows] = {0};
  for (int b = 0; b < n; ++b) {
    SMI_HIP(hipMemcpyAsync(L->count + slots[b], zeros, 4, hipMemcpyHostToDevice, st));
    SMI_HIP(hipMemcpyAsync(L->finished + slots[b], zeros, 4, hipMemcpyHostToDevice, st));
  }
  SMI_HIP(hipMemcpyAsync(L->rows, L->plan + tail, kMaxRows * sizeof(RowDesc), hipMemcpyDeviceToDevice, st));
  L->graph = nullptr;
  const int oldB = L->B;
  L->B = n;
  L->identity_slots = 1;
  for (int b = 0; b < n; ++b) L->identity_slots &= slots[b] == b;
  if ((rc = launch_embed(L, L->rows, n, st)) || (rc = launch_step(L, n, st))) { L->B = oldB; return rc; }
  std::vector<RowDesc> fresh;
  if ((rc = session_live_rows(L, fresh, st))) return rc;
  for (int b = 0; b < n; ++b) {
    live.push_back(fresh[b]);
    L->slot_busy[slots[b]] = 1;
    L->slot_len[slots[b]] = lens[b] + 1;
    slots_out[b] = slots[b];
  }
  return session_set_rows(L, live, st);
}

int smi_llm_retire(smi_llm* L, int slot, void* stream) {
  SMI_REQUIRE(L && slot >= 0 && slot < kMaxRows, "smi_llm_retire: bad argument");
  if (!L->started || !L->session) { smi_set_error("smi_llm_retire outside a session"); return SMI_ESTATE; }
  SMI_REQUIRE(L->slot_busy[slot], "smi_llm_retire: slot %d is not live", slot);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  std::vector<RowDesc> live;
  if ((rc = session_live_rows(L, live, st))) return rc;
  for (size_t i = 0; i < live.size(); ++i)
    if (live[i].slot == slot) { live.erase(live.begin() + (long)i); break; }
  L->slot_busy[slot] = 0;
  L->slot_len[slot] = 0;
  if (L->paged) pages_release(L, slot);   // its pages go back to the pool (stale table entries are never read: no live row names the slot)
  return session_set_rows(L, live, st);
}

// Several sequences leave at once, without a host round trip: the device row list is compacted in place (order kept), the
// per-row state of the rows that moved is rebuilt from their descriptors (as after any change of the live set), the slots
// (and their pages) are free again.  Their histories stay readable (smi_llm_slots_tokens) until a later admission reuses a slot.
int smi_llm_retire_many(smi_llm* L, const int32_t* slots, int n, void* stream) {
  SMI_REQUIRE(L && slots && n >= 1 && n <= kMaxRows, "smi_llm_retire_many: bad argument");
  if (!L->started || !L->session) { smi_set_error("smi_llm_retire_many outside a session"); return SMI_ESTATE; }
  unsigned long long drop = 0;
  for (int i = 0; i < n; ++i) {
    SMI_REQUIRE(slots[i] >= 0 && slots[i] < kMaxRows && L->slot_busy[slots[i]] && !((drop >> slots[i]) & 1ull), "smi_llm_retire_many: slot %d is not live (or listed twice)", slots[i]);
    drop |= 1ull << slots[i];
  }
  SMI_REQUIRE((int)L->live_order.size() == L->B, "smi_llm_retire_many: live row list out of step");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_rows_drop, dim3(1), dim3(64), 0, st, L->rows, L->B, drop);
  SMI_LAUNCH_CHECK();
  std::vector<int> keep;
  for (int sl : L->live_order) if (!((drop >> sl) & 1ull)) keep.push_back(sl);
  for (int i = 0; i < n; ++i) {
    L->slot_busy[slots[i]] = 0;
    L->slot_len[slots[i]] = 0;
    if (L->paged) pages_release(L, slots[i]);
  }
  L->live_order = keep;
  L->B = (int)keep.size();
  L->identity_slots = 1;
  for (int b = 0; b < L->B; ++b) L->identity_slots &= keep[b] == b;
  L->graph = nullptr;
  if (L->B == 0) return SMI_OK;
  return launch_embed(L, L->rows, L->B, st);
}

// Tokens of several slots in one device round trip: out_host [n][cap], n_out[n], finished[n].
int smi_llm_slots_tokens(smi_llm* L, const int32_t* slots, int n, int64_t* out_host, int cap, int32_t* n_out, int32_t* finished, void* stream) {
  SMI_REQUIRE(L && slots && out_host && n_out && finished && n >= 1 && n <= kMaxRows && cap >= 1, "smi_llm_slots_tokens: bad argument");
  for (int i = 0; i < n; ++i) SMI_REQUIRE(slots[i] >= 0 && slots[i] < kMaxRows, "smi_llm_slots_tokens: slot %d out of range", slots[i]);
  hipStream_t st = (hipStream_t)stream;
  int32_t cnt[kMaxRows], fin[kMaxRows];
  SMI_HIP(hipMemcpyAsync(cnt, L->count, kMaxRows * 4, hipMemcpyDeviceToHost, st));
  SMI_HIP(hipMemcpyAsync(fin, L->finished, kMaxRows * 4, hipMemcpyDeviceToHost, st));
  int steps = cap < L->max_steps ? cap : L->max_steps;
  std::vector<int64_t> hist((size_t)steps * kMaxRows);
  SMI_HIP(hipMemcpyAsync(hist.data(), L->hist, hist.size() * 8, hipMemcpyDeviceToHost, st));
  SMI_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) {
    int k = cnt[slots[i]] < steps ? cnt[slots[i]] : steps;
    for (int t = 0; t < k; ++t) out_host[(size_t)i * cap + t] = hist[(size_t)t * kMaxRows + slots[i]];
    n_out[i] = k;
    finished[i] = fin[slots[i]];
  }
  return SMI_OK;
}

// Tokens a live (or just finished) slot has emitted since it was admitted; *finished: it has produced eos.
int smi_llm_slot_tokens(smi_llm* L, int slot, int64_t* out, int cap, int32_t* n_out, int32_t* finished, void* stream) {
  SMI_REQUIRE(L && out && n_out && finished && slot >= 0 && slot < kMaxRows && cap >= 0, "smi_llm_slot_tokens: bad argument");
  hipStream_t st = (hipStream_t)stream;
  int32_t cnt = 0, fin = 0;
  SMI_HIP(hipMemcpyAsync(&cnt, L->count + slot, 4, hipMemcpyDeviceToHost, st));
  SMI_HIP(hipMemcpyAsync(&fin, L->finished + slot, 4, hipMemcpyDeviceToHost, st));
  SMI_HIP(hipStreamSynchronize(st));
  int n = cnt < cap ? cnt : cap;
  if (n > L->max_steps) n = L->max_steps;
  if (n > 0) SMI_HIP(hipMemcpy2D(out, 8, L->hist + slot, kMaxRows * 8, 8, (size_t)n, hipMemcpyDeviceToHost));
  *n_out = n;
  *finished = fin;
  return SMI_OK;
}

int smi_llm_decode(smi_llm* L, int n_steps, void* stream) {
  SMI_REQUIRE(L, "smi_llm_decode: null handle");
  if (!L->started) { smi_set_error("smi_llm_decode before smi_llm_prefill"); return SMI_ESTATE; }
  SMI_REQUIRE(n_steps >= 0, "smi_llm_decode: n_steps < 0");
  if (L->session) {
    if (L->B == 0 || n_steps == 0) return SMI_OK;
    for (int sl = 0; sl < kMaxRows; ++sl)
      if (L->slot_busy[sl])
        SMI_REQUIRE(L->slot_len[sl] + n_steps <= L->cfg.max_positions, "smi_llm_decode: %d more steps would take slot %d past max_positions=%d",
                    n_steps, sl, L->cfg.max_positions);
  } else {
    SMI_REQUIRE(L->max_len + L->steps_launched + n_steps <= L->cfg.max_positions,
                "smi_llm_decode: %d more steps would pass max_positions=%d (prompt %d, %d steps so far)", n_steps,
                L->cfg.max_positions, L->max_len, L->steps_launched);
  }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (L->paged && n_steps > 0) {   // every live sequence grows by n_steps positions: their pages first, all or nothing
    int sl[kMaxRows], need[kMaxRows], n = 0;
    if (L->session) {
      for (int s2 = 0; s2 < kMaxRows; ++s2)
        if (L->slot_busy[s2]) { sl[n] = s2; need[n++] = L->slot_len[s2] + n_steps; }
    } else {
      for (int b = 0; b < L->B; ++b) { sl[n] = b; need[n++] = L->plen[b] + L->steps_launched + n_steps; }
    }
    for (int i = 0; i < n; ++i) need[i] = need[i] < L->cfg.max_positions ? need[i] : L->cfg.max_positions;
    if ((rc = pages_ensure(L, sl, need, n, st))) return rc;
  }
  {   // context bound of this call -> attention segments (the partial buffer must exist before a capture starts)
    int bound = L->max_len + L->steps_launched + n_steps;
    if (L->session) {
      bound = 0;
      for (int sl = 0; sl < kMaxRows; ++sl)
        if (L->slot_busy[sl] && L->slot_len[sl] + n_steps > bound) bound = L->slot_len[sl] + n_steps;
    }
    L->attn_seg = segs_for(bound);
    if (L->attn_seg > 1 && (rc = ensure_apart(L, (size_t)kMaxRows * L->cfg.num_heads * L->attn_seg * 66))) return rc;
  }
  if (L->cfg.use_graph && n_steps > 0 && (!L->graph || L->graph_B != L->B || L->graph_seg != L->attn_seg || L->graph_ident != L->identity_slots)) {
    const uint32_t key = (uint32_t)L->B | ((uint32_t)L->attn_seg << 8) | ((uint32_t)(L->identity_slots ? 1 : 0) << 24);
    auto hit = L->graph_cache.find(key);
    L->graph = hit != L->graph_cache.end() ? hit->second : nullptr;
    L->graph_B = L->B; L->graph_seg = L->attn_seg; L->graph_ident = L->identity_slots;
  }
  if (L->cfg.use_graph && n_steps > 0 && !L->graph) {
    const uint32_t key = (uint32_t)L->B | ((uint32_t)L->attn_seg << 8) | ((uint32_t)(L->identity_slots ? 1 : 0) << 24);
    if (L->graph_cache.size() >= 192) graphs_flush(L);
    hipStream_t cs;
    SMI_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
      rc = launch_step(L, L->B, cs);
      hipError_t e2 = hipStreamEndCapture(cs, &g);
      if (rc == SMI_OK && e2 == hipSuccess && g) {
        if (hipGraphInstantiate(&L->graph, g, nullptr, nullptr, 0) != hipSuccess) L->graph = nullptr;
      }
      if (g) (void)hipGraphDestroy(g);
    }
    (void)hipStreamDestroy(cs);
    (void)hipGetLastError();
    L->graph_B = L->B; L->graph_seg = L->attn_seg; L->graph_ident = L->identity_slots;
    if (!L->graph) { smi_set_error("hipGraph capture of the decode step failed"); return SMI_EHIP; }
    L->graph_cache[key] = L->graph;
  }
  // Several steps per replay (SPARKMI_GRAPH_STEPS = K, default 8; 1: off): a step needs nothing from the host, so K of them are
  // captured back to back into one graph; the call replays it n_steps / K times and the one-step graph for the rest.  One replay
  // boundary costs ~5 us that a kernel boundary inside a graph does not: graph step at one row 548.6 -> 544.5 us (K = 4 .. 25
  // alike; profiles/r03_graph_steps.txt).  The K-step graph of a (rows, segments) key is only built by a call long enough to
  // replay it twice (a capture of 8 x 98 nodes is not free; short serving strides keep to the one-step graph).
  int s0 = 0;
  if (L->cfg.use_graph && L->graph_steps > 1 && n_steps >= L->graph_steps) {
    const int K = L->graph_steps;
    const uint32_t keyk = (uint32_t)L->B | ((uint32_t)L->attn_seg << 8) | ((uint32_t)(L->identity_slots ? 1 : 0) << 24) | ((uint32_t)K << 26);
    hipGraphExec_t gk = nullptr;
    auto hit = L->graph_cache.find(keyk);
    if (hit != L->graph_cache.end()) gk = hit->second;
    else if (n_steps >= 2 * K && L->graph_cache.size() < 192) {
      hipStream_t cs;
      SMI_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
      hipGraph_t g = nullptr;
      hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        rc = SMI_OK;
        for (int k = 0; k < K && rc == SMI_OK; ++k) rc = launch_step(L, L->B, cs);
        hipError_t e2 = hipStreamEndCapture(cs, &g);
        if (rc == SMI_OK && e2 == hipSuccess && g && hipGraphInstantiate(&gk, g, nullptr, nullptr, 0) != hipSuccess) gk = nullptr;
        if (g) (void)hipGraphDestroy(g);
      }
      (void)hipStreamDestroy(cs);
      (void)hipGetLastError();
      if (gk) L->graph_cache[keyk] = gk;
    }
    if (gk) {
      for (; s0 + K <= n_steps; s0 += K) {
        SMI_HIP(hipGraphLaunch(gk, st));
        L->graph_last[gk] = st;
      }
    }
  }
  for (int s = s0; s < n_steps; ++s) {
    if (L->cfg.use_graph) {
      SMI_HIP(hipGraphLaunch(L->graph, st));
      L->graph_last[L->graph] = st;
    } else if ((rc = launch_step(L, L->B, st))) {
      return rc;
    }
  }
  L->steps_launched += n_steps;
  if (L->session)
    for (int sl = 0; sl < kMaxRows; ++sl)
      if (L->slot_busy[sl]) L->slot_len[sl] += n_steps;
  return SMI_OK;
}

int smi_llm_all_done(smi_llm* L, int* all_done, void* stream) {
  SMI_REQUIRE(L && all_done, "smi_llm_all_done: null argument");
  int32_t fin[kMaxRows];
  SMI_HIP(hipMemcpyAsync(fin, L->finished, sizeof(fin), hipMemcpyDeviceToHost, (hipStream_t)stream));
  SMI_HIP(hipStreamSynchronize((hipStream_t)stream));
  { const int rce = eng_check(L); if (rce) return rce; }
  int d = 1;
  for (int b = 0; b < L->B; ++b) d &= fin[b] != 0;
  *all_done = d;
  return SMI_OK;
}

// Tokens emitted and eos flags of every KV slot in one round trip (continuous-batching drivers poll this every few steps)
int smi_llm_status(smi_llm* L, int32_t* count_host, int32_t* finished_host, void* stream) {
  SMI_REQUIRE(L && count_host && finished_host, "smi_llm_status: null argument");
  hipStream_t st = (hipStream_t)stream;
  SMI_HIP(hipMemcpyAsync(count_host, L->count, kMaxRows * 4, hipMemcpyDeviceToHost, st));
  SMI_HIP(hipMemcpyAsync(finished_host, L->finished, kMaxRows * 4, hipMemcpyDeviceToHost, st));
  SMI_HIP(hipStreamSynchronize(st));
  return eng_check(L);
}

int smi_llm_steps(smi_llm* L) {
  if (!L) return SMI_EINVAL;
  int32_t s = 0;
  if (hipMemcpy(&s, L->step, 4, hipMemcpyDeviceToHost) != hipSuccess) return SMI_EHIP;
  return s;
}

int smi_llm_kv_pages(smi_llm* L, int32_t* total, int32_t* free_pages) {
  SMI_REQUIRE(L && total && free_pages, "smi_llm_kv_pages: null argument");
  *total = L->paged ? L->cfg.kv_pages : 0;
  *free_pages = L->paged ? (int32_t)L->free_pages.size() : 0;
  return SMI_OK;
}

int smi_llm_get_tokens(smi_llm* L, int64_t* out, int32_t* lens, int cap, void* stream) {
  SMI_REQUIRE(L && out && lens && cap >= 0, "smi_llm_get_tokens: bad argument");
  hipStream_t st = (hipStream_t)stream;
  int32_t cnt[kMaxRows], step = 0;
  SMI_HIP(hipMemcpyAsync(cnt, L->count, sizeof(cnt), hipMemcpyDeviceToHost, st));
  SMI_HIP(hipMemcpyAsync(&step, L->step, 4, hipMemcpyDeviceToHost, st));
  SMI_HIP(hipStreamSynchronize(st));
  { const int rce = eng_check(L); if (rce) return rce; }
  if (step > L->max_steps) step = L->max_steps;
  std::vector<int64_t> hist((size_t)step * kMaxRows);
  if (step) SMI_HIP(hipMemcpy(hist.data(), L->hist, hist.size() * 8, hipMemcpyDeviceToHost));
  for (int b = 0; b < L->B; ++b) {
    int n = cnt[b] < cap ? cnt[b] : cap;
    if (n > step) n = step;
    lens[b] = n;
    for (int s = 0; s < n; ++s) out[(size_t)b * cap + s] = hist[(size_t)s * kMaxRows + b];
  }
  return SMI_OK;
}

int smi_llm_forward_logits(smi_llm* L, const int64_t* ids, int S, float* logits_dev, void* stream) {
  SMI_REQUIRE(L && ids && logits_dev, "smi_llm_forward_logits: null argument");
  SMI_REQUIRE(S >= 1 && S < L->cfg.max_positions, "smi_llm_forward_logits: S=%d outside 1..max_positions-1", S);
  hipStream_t st = (hipStream_t)stream;
  const size_t nchunks = ((size_t)S + kMaxRows - 1) / kMaxRows;
  int rc;
  L->attn_seg = segs_for(S);
  if (L->paged) {
    for (int b = 0; b < kMaxRows; ++b) pages_release(L, b);
    const int s0 = 0;
    if ((rc = pages_ensure(L, &s0, &S, 1, st))) return rc;
  }
  if ((rc = ensure_plan(L, nchunks * kMaxRows))) return rc;
  L->host_rows.assign(nchunks * kMaxRows, RowDesc{0, 0, 0, 0});
  for (int t = 0; t < S; ++t) {
    SMI_REQUIRE(ids[t] >= 0 && ids[t] < L->cfg.vocab_size, "smi_llm_forward_logits: token id out of range");
    L->host_rows[t] = RowDesc{0, t, (int32_t)ids[t], 1};
  }
  SMI_HIP(hipMemcpyAsync(L->plan, L->host_rows.data(), L->host_rows.size() * sizeof(RowDesc), hipMemcpyHostToDevice, st));
  for (size_t c = 0; c < nchunks; ++c) {
    const int M = (int)(((size_t)S - c * kMaxRows) < (size_t)kMaxRows ? ((size_t)S - c * kMaxRows) : kMaxRows);
    const RowDesc* rows = L->plan + c * kMaxRows;
    if ((rc = launch_embed(L, rows, M, st))) return rc;
    if ((rc = launch_layers(L, rows, M, false, st))) return rc;
    if ((rc = launch_one(L, KLM, 0, rows, M, logits_dev + c * kMaxRows * (size_t)L->cfg.vocab_size, st))) return rc;
  }
  L->started = 0;  // the cache now holds this sequence; a generate must prefill again
  return SMI_OK;
}

#ifdef SMI_DIAG   // ---- everything below: include/sparkmi_debug.h, exported by libsparkmi_diag.so only
// Diagnostics: one launch of a decode-step GEMM kernel with in-kernel s_memrealtime stamps
// (10 ns ticks); out[0..7) = mean over blocks of (stamp i - earliest stamp 0), out[7] = blocks.
int smi_llm_debug_stamps(smi_llm* L, int kernel, int layer, double* out) {
  SMI_REQUIRE(L && out && L->started, "smi_llm_debug_stamps: needs a started generation");
  // kernel + 32: the layer's earlier kernels run first (un-stamped, with their helper blocks), so the stamped kernel finds in
  // the caches what it finds inside a decode step -- the in-kernel evidence for (or against) the helpers' prefetch
  const bool with_producers = (kernel & 32) != 0;
  kernel &= 31;
  SMI_REQUIRE(kernel == KQKV || kernel == KO || kernel == KGU || kernel == KD || kernel == KLM, "smi_llm_debug_stamps: GEMM kernels only");
  SMI_REQUIRE(!(kernel == KLM && L->B <= 16 && L->KTh <= 32), "smi_llm_debug_stamps: the persistent lm_head has no stamps");
  SMI_HIP(hipMemset(L->stamps, 0, (size_t)4096 * 64));
  graphs_flush(L);   // (the stamped kernels are other instantiations)
  int rc = SMI_OK;
  if (with_producers) {
    if (layer > 0) rc = launch_one(L, KD, layer - 1, L->rows, L->B, nullptr, 0);   // (its helpers warm this layer's QKV / o_proj slices)
    for (int k = KQKV; k < kernel && rc == SMI_OK; ++k) rc = launch_one(L, k, layer, L->rows, L->B, nullptr, 0);
    if (rc) return rc;
  }
  L->stamps_on = 1;
  rc = launch_one(L, kernel, layer, L->rows, L->B, nullptr, 0);
  L->stamps_on = 0;
  if (rc) return rc;
  SMI_HIP(hipDeviceSynchronize());
  // the launch's grid depends on the row count: the blocks that ran are the ones that left an entry stamp
  std::vector<unsigned long long> h((size_t)4096 * 8);
  SMI_HIP(hipMemcpy(h.data(), L->stamps, h.size() * 8, hipMemcpyDeviceToHost));
  int nblk = 0;
  while (nblk < 4096 && h[(size_t)nblk * 8]) ++nblk;
  SMI_REQUIRE(nblk > 0, "smi_llm_debug_stamps: the kernel left no stamps");
  unsigned long long t0 = ~0ull;
  for (int b = 0; b < nblk; ++b) t0 = h[(size_t)b * 8] < t0 ? h[(size_t)b * 8] : t0;
  for (int i = 0; i < 7; ++i) {
    double s = 0;
    for (int b = 0; b < nblk; ++b) s += (double)(h[(size_t)b * 8 + i] - t0);
    out[i] = s / nblk * 0.01;   // microseconds
  }
  double cyc = 0, us = 0;   // shader clock during the kernel: cycles / (stamp6 - stamp0)
  for (int b = 0; b < nblk; ++b) { cyc += (double)h[(size_t)b * 8 + 7]; us += (double)(h[(size_t)b * 8 + 6] - h[(size_t)b * 8]) * 0.01; }
  out[7] = us > 0 ? cyc / us : 0;   // MHz
  return SMI_OK;
}

int smi_llm_time_kernel(smi_llm* L, int kernel, int layer, int iters, float* ms_avg, void* stream) {
  SMI_REQUIRE(L && ms_avg && iters > 0, "smi_llm_time_kernel: bad argument");
  SMI_REQUIRE((kernel >= 0 && kernel <= 8) || (kernel >= 16 && kernel <= 16 + KD), "smi_llm_time_kernel: kernel id %d", kernel);
  if (!L->started) { smi_set_error("smi_llm_time_kernel needs a started generation (prefill first)"); return SMI_ESTATE; }
  if (L->paged && kernel != 7) { smi_set_error("smi_llm_time_kernel: the per-kernel probes need a contiguous KV cache (kv_page_tokens = 0)"); return SMI_ESTATE; }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (kernel == 8) {
    // all layers of one step for the live rows: ONE launch of the one-row engine where it applies, else the layer kernels
    // back to back (each launch takes the previous one's residual row as its input: the values drift, the timing does not)
    const bool eng = eng_usable(L, L->rows, L->B);
    if ((rc = eng ? eng_launch(L, st) : launch_layers(L, L->rows, L->B, false, st))) return rc;
    SMI_HIP(hipEventRecord(L->ev0, st));
    for (int i = 0; i < iters; ++i)
      if ((rc = eng ? eng_launch(L, st) : launch_layers(L, L->rows, L->B, false, st))) return rc;
    SMI_HIP(hipEventRecord(L->ev1, st));
    L->started = 0;   // the residual row no longer belongs to the generation
  } else if (kernel == 7) {
    if ((rc = smi_llm_decode(L, 1, stream))) return rc;  // builds the graph if needed
    SMI_HIP(hipEventRecord(L->ev0, st));
    if ((rc = smi_llm_decode(L, iters, stream))) return rc;
    SMI_HIP(hipEventRecord(L->ev1, st));
  } else if (kernel >= 16) {
    // In-sequence cost of one layer kernel: (time of iters layers) - (time of the same layers without it).
    // Its producers run, so what they leave in L2 (the helper prefetch) is what it finds -- the
    // duration it has inside the decode graph, which the back-to-back loop below cannot show.
    const int k = kernel - 16, nl = L->cfg.num_layers;
    SMI_REQUIRE(k >= KQKV && k <= KD, "smi_llm_time_kernel: in-sequence timing is for the layer kernels");
    // both sequences are captured into hipGraphs and replayed, like the decode step itself: eager launches of
    // 5-us kernels can be bound by the host's launch rate, which would then be what the difference measures
    float t[2] = {0.f, 0.f};
    for (int pass = 0; pass < 2; ++pass) {
      hipStream_t cs;
      SMI_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
      hipGraph_t g = nullptr;
      hipGraphExec_t ge = nullptr;
      rc = SMI_OK;
      hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        for (int i = 0; i < iters && rc == SMI_OK; ++i) {
          const int l = (layer + i) % nl;
          for (int kk = KQKV; kk <= KD && rc == SMI_OK; ++kk)
            if (!(pass == 1 && kk == k)) rc = launch_one(L, kk, l, L->rows, L->B, nullptr, cs);
        }
        e = hipStreamEndCapture(cs, &g);
      }
      if (e == hipSuccess && rc == SMI_OK && g) e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      if (g) (void)hipGraphDestroy(g);
      (void)hipStreamDestroy(cs);
      if (e != hipSuccess || rc != SMI_OK || !ge) {
        if (ge) (void)hipGraphExecDestroy(ge);
        (void)hipGetLastError();
        if (rc != SMI_OK) return rc;
        smi_set_error("smi_llm_time_kernel: graph capture of the layer sequence failed");
        return SMI_EHIP;
      }
      hipError_t le = hipGraphLaunch(ge, st);                       // warm (also brings the L2 / MALL into steady state)
      if (le == hipSuccess) le = hipEventRecord(L->ev0, st);
      if (le == hipSuccess) le = hipGraphLaunch(ge, st);
      if (le == hipSuccess) le = hipEventRecord(L->ev1, st);
      if (le == hipSuccess) le = hipEventSynchronize(L->ev1);
      if (le == hipSuccess) le = hipEventElapsedTime(&t[pass], L->ev0, L->ev1);
      (void)hipGraphExecDestroy(ge);
      if (le != hipSuccess) { smi_set_error("smi_llm_time_kernel: graph replay failed: %s", hipGetErrorString(le)); return SMI_EHIP; }
    }
    *ms_avg = (t[0] - t[1]) / iters;
    return SMI_OK;
  } else {
    const int nl = L->cfg.num_layers;
    if (kernel == KFIN) {
      SMI_REQUIRE(L->max_len + L->steps_launched + iters + 1 <= L->cfg.max_positions, "smi_llm_time_kernel: finalize probe would pass max_positions");
      L->steps_launched += iters + 1;
    }
    if ((rc = launch_one(L, kernel, layer % nl, L->rows, L->B, nullptr, st))) return rc;
    SMI_HIP(hipEventRecord(L->ev0, st));
    for (int i = 0; i < iters; ++i) {
      // walk the layers so the weights come from HBM, not from the Infinity Cache
      const int l = kernel >= KLM ? 0 : (layer + 1 + i) % nl;
      if ((rc = launch_one(L, kernel, l, L->rows, L->B, nullptr, st))) return rc;
    }
    SMI_HIP(hipEventRecord(L->ev1, st));
  }
  SMI_HIP(hipEventSynchronize(L->ev1));
  float ms = 0.f;
  SMI_HIP(hipEventElapsedTime(&ms, L->ev0, L->ev1));
  *ms_avg = ms / iters;
  return SMI_OK;
}

// ---- one-row decode engine (smi_eng.h)
int smi_llm_engine(smi_llm* L, int32_t* enabled, int32_t* info, char* why, int n) {
  SMI_REQUIRE(L, "smi_llm_engine: null handle");
  if (enabled) *enabled = L->eng.enabled && L->eng_on;
  if (info) { info[0] = L->eng.ncu; info[1] = L->eng.maxlen; info[2] = L->eng.lds; info[3] = L->eng.enabled; }
  if (why && n > 0) { strncpy(why, L->eng.why, (size_t)n - 1); why[n - 1] = 0; }
  return SMI_OK;
}

int smi_llm_set_engine(smi_llm* L, int on) {
  SMI_REQUIRE(L, "smi_llm_set_engine: null handle");
  int v = on ? 1 : 0;
  if (v && !L->eng.enabled) {   // first request: build it (plan, 0.8 GB weight stream); stays off, with the reason, where it does not apply
    SMI_HIP(hipDeviceSynchronize());
    const int rce = eng_create(L);
    if (rce) return rce;
    v = L->eng.enabled;
  }
  if (v != L->eng_on) { SMI_HIP(hipDeviceSynchronize()); graphs_flush(L); }   // captured steps hold one path or the other
  L->eng_on = v;
  return SMI_OK;
}

int smi_llm_engine_plan(const smi_llm_cfg* cfg, int ncu, int32_t* stats) {
  SMI_REQUIRE(cfg_ok(cfg) && stats, "smi_llm_engine_plan: invalid argument");
  EngPlanHost P;
  char why[128] = "";
  const int Q = cfg->num_heads * cfg->head_dim, KV = cfg->num_kv_heads * cfg->head_dim;
  if (!eng_build_plan(cfg->hidden_size, Q, KV, cfg->intermediate_size, cfg->num_heads, ncu, P, why, sizeof(why))) {
    smi_set_error("engine plan: %s", why);
    return SMI_EINVAL;
  }
  // every (matrix, part, chain set, round) exactly once, in a stream whose order matches the jobs
  long total = 0, want = 0;
  for (int ph = 0; ph < 4; ++ph) want += (long)P.g.nparts[ph] * P.g.part_imgs[ph];
  std::vector<unsigned char> seen[4];
  for (int ph = 0; ph < 4; ++ph) seen[ph].assign((size_t)P.g.nparts[ph] * 4 * 64, 0);
  for (int c = 0; c < ncu; ++c)
    for (int w = 0; w < kEngWaves; ++w) {
      const EngWavePlan& wp = P.cu[c].w[w];
      for (int ph = 0; ph < 4; ++ph)
        for (int j = wp.jstart[ph]; j < wp.jstart[ph + 1]; ++j) {
          const EngJob& jb = wp.jobs[j];
          SMI_REQUIRE(P.cu[c].parts[P.cu[c].pstart[ph] + jb.slot] == jb.part, "engine plan: job slot does not name its part");
          SMI_REQUIRE((int)jb.goff + jb.nimg <= (int)P.cu[c].len_cu, "engine plan: job beyond the CU's stream");
          for (int r = 0; r < jb.nimg; ++r) {
            const uint32_t d = P.desc[(size_t)c * P.maxlen + jb.goff + r];
            SMI_REQUIRE(d == (((uint32_t)ph << 30) | ((uint32_t)jb.set << 26) | ((uint32_t)r << 16) | jb.part), "engine plan: stream order differs from the job order");
            unsigned char& f = seen[ph][((size_t)jb.part * 4 + jb.set) * 64 + r];
            SMI_REQUIRE(r < 64 && !f, "engine plan: image listed twice");
            f = 1;
            ++total;
          }
        }
    }
  SMI_REQUIRE(total == want, "engine plan: %ld images placed, %ld expected", total, want);
  const EngLds lds = eng_lds(cfg->hidden_size, P.g.KT[EPH_DOWN], P.g.KT[EPH_QKV] > P.g.KT[EPH_O] ? P.g.KT[EPH_QKV] : P.g.KT[EPH_O]);
  stats[0] = P.maxlen; stats[1] = P.max_wave_phase_imgs; stats[2] = P.max_slots; stats[3] = P.max_jobs;
  stats[4] = P.min_load; stats[5] = P.max_load; stats[6] = lds.total; stats[7] = (int32_t)(total);
  return SMI_OK;
}

// Tests: the residual row of sequence row 0 as the last step left it (input of the next step's first layer).
int smi_llm_debug_hidden(smi_llm* L, float* out_host, int n) {
  SMI_REQUIRE(L && out_host && n >= L->H, "smi_llm_debug_hidden: out holds %d floats, %d needed", n, L ? L->H : 0);
  SMI_HIP(hipDeviceSynchronize());
  SMI_HIP(hipMemcpy(out_host, L->h, (size_t)L->H * 4, hipMemcpyDeviceToHost));
  return eng_check(L);
}

// Tests / debugging: raw copy of one scratch buffer (what: 0 q [Q] f32, 1 attention operand triples, 2 act triples, 3 h operand
// triples, 4 h [H] f32, 5 engine granules [2][per buffer] u64, 6 partial sums of squares [H / 4], 7 layer-0 K cache of slot 0
// head 0, 8 h + o_proj [H] (fused path)); returns the bytes copied in *got.
int smi_llm_debug_read(smi_llm* L, int what, void* out_host, size_t cap, size_t* got) {
  SMI_REQUIRE(L && out_host && got, "smi_llm_debug_read: null argument");
  const void* src = nullptr;
  size_t n = 0;
  const size_t R = L->B > 0 ? (size_t)L->B : 1;   // buffers 0..4 and 6 hold one entry per live row
  switch (what) {
    case 0: src = L->qbuf; n = R * L->Q * 4; break;
    case 1: src = L->xs_attn; n = R * L->Q * 6; break;
    case 2: src = L->xs_act; n = R * L->I * 6; break;
    case 3: src = L->xs_h; n = R * L->H * 6; break;
    case 4: src = L->h; n = R * L->H * 4; break;
    case 5: src = L->eng.gran; n = L->eng.gran ? (size_t)2 * L->eng.gran_per_buf * 8 : 0; break;
    case 6: src = L->sspart; n = R * L->H; break;
    case 7: src = L->kcache; n = (size_t)L->cfg.max_positions * kHeadDim * 2; break;
    case 8: src = L->h2; n = (size_t)L->H * 4; break;
    default: smi_set_error("smi_llm_debug_read: what=%d", what); return SMI_EINVAL;
  }
  SMI_REQUIRE(src && n <= cap, "smi_llm_debug_read: buffer %d holds %zu bytes, out holds %zu", what, n, cap);
  SMI_HIP(hipDeviceSynchronize());
  SMI_HIP(hipMemcpy(out_host, src, n, hipMemcpyDeviceToHost));
  *got = n;
  return SMI_OK;
}

// ---- op-level tests (sparkmi_debug.h): one layer's kernels, stage by stage, on caller-given rows -- through launch_one, i.e. the
// launch builders, kernel choices and epilogue fusions of a real step
static int debug_kv_check(const smi_llm* L, int layer, int slot, int pos0, int n) {
  SMI_REQUIRE(L && !L->paged, "debug KV access needs a contiguous KV cache (kv_page_tokens = 0)");
  SMI_REQUIRE(layer >= 0 && layer < L->cfg.num_layers && slot >= 0 && slot < L->cfg.max_slots && pos0 >= 0 && n >= 0 &&
              pos0 + n <= L->cfg.max_positions, "debug KV access: layer %d slot %d positions %d..%d out of range", layer, slot, pos0, pos0 + n);
  return SMI_OK;
}

// k_host / v_host [n][num_kv_heads][64] fp32 in transformers' dim order, keys ALREADY rotated (what a KV cache holds); written to
// positions pos0 .. pos0 + n - 1 of `slot` in the cache's own dtype and row order (K: RoPE pairs adjacent, include/sparkmi.h)
int smi_llm_debug_set_kv(smi_llm* L, int layer, int slot, int pos0, int n, const float* k_host, const float* v_host) {
  { const int rc = debug_kv_check(L, layer, slot, pos0, n); if (rc) return rc; }
  SMI_REQUIRE(k_host && v_host, "smi_llm_debug_set_kv: null argument");
  SMI_HIP(hipDeviceSynchronize());
  const int nkv = L->cfg.num_kv_heads, f32 = L->cfg.kv_dtype;
  const size_t esz = f32 ? 4 : 2;
  std::vector<unsigned char> buf((size_t)n * kHeadDim * esz);
  for (int isv = 0; isv < 2; ++isv)
    for (int kh = 0; kh < nkv; ++kh) {
      for (int t = 0; t < n; ++t)
        for (int i = 0; i < kHeadDim; ++i) {
          const int d = isv ? i : (i >> 1) + 32 * (i & 1);   // K rows sit as RoPE pairs (0, 32, 1, 33, ...)
          const float x = (isv ? v_host : k_host)[((size_t)t * nkv + kh) * kHeadDim + d];
          if (f32) ((float*)buf.data())[(size_t)t * kHeadDim + i] = x;
          else {
            uint32_t u; memcpy(&u, &x, 4);
            ((uint16_t*)buf.data())[(size_t)t * kHeadDim + i] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
          }
        }
      unsigned char* base = (unsigned char*)kv_layer(L, isv ? L->vcache : L->kcache, layer);
      const size_t row0 = ((size_t)slot * nkv + kh) * L->cfg.max_positions + pos0;
      SMI_HIP(hipMemcpy(base + row0 * kHeadDim * esz, buf.data(), buf.size(), hipMemcpyHostToDevice));
    }
  return SMI_OK;
}

// the reverse: positions pos0 .. pos0 + n - 1 of `slot` as fp32 [n][num_kv_heads][64] in transformers' dim order
int smi_llm_debug_get_kv(smi_llm* L, int layer, int slot, int pos0, int n, float* k_host, float* v_host) {
  { const int rc = debug_kv_check(L, layer, slot, pos0, n); if (rc) return rc; }
  SMI_REQUIRE(k_host && v_host, "smi_llm_debug_get_kv: null argument");
  SMI_HIP(hipDeviceSynchronize());
  const int nkv = L->cfg.num_kv_heads, f32 = L->cfg.kv_dtype;
  const size_t esz = f32 ? 4 : 2;
  std::vector<unsigned char> buf((size_t)n * kHeadDim * esz);
  for (int isv = 0; isv < 2; ++isv)
    for (int kh = 0; kh < nkv; ++kh) {
      const unsigned char* base = (const unsigned char*)kv_layer(L, isv ? L->vcache : L->kcache, layer);
      const size_t row0 = ((size_t)slot * nkv + kh) * L->cfg.max_positions + pos0;
      SMI_HIP(hipMemcpy(buf.data(), base + row0 * kHeadDim * esz, buf.size(), hipMemcpyDeviceToHost));
      for (int t = 0; t < n; ++t)
        for (int i = 0; i < kHeadDim; ++i) {
          const int d = isv ? i : (i >> 1) + 32 * (i & 1);
          float x;
          if (f32) x = ((const float*)buf.data())[(size_t)t * kHeadDim + i];
          else { const uint32_t u = (uint32_t)((const uint16_t*)buf.data())[(size_t)t * kHeadDim + i] << 16; memcpy(&x, &u, 4); }
          (isv ? v_host : k_host)[((size_t)t * nkv + kh) * kHeadDim + d] = x;
        }
    }
  return SMI_OK;
}

// rows_host: M (slot, pos) pairs; hidden_host [M][hidden] fp32 = the residual rows entering `layer`.  Runs that layer's kernels up
// to and including `stage` (0 QKV + bias + RoPE + KV append, 1 attention, 2 o_proj + residual, 3 gate_up + SwiGLU, 4 down_proj +
// residual) for those rows through launch_one; results are then read with smi_llm_debug_read / smi_llm_debug_get_kv.  Ends the
// current generation (the handle needs a prefill before it generates again).
int smi_llm_debug_layer(smi_llm* L, int layer, int M, const int32_t* rows_host, const float* hidden_host, int stage) {
  SMI_REQUIRE(L && rows_host && hidden_host && M >= 1 && M <= kMaxRows, "smi_llm_debug_layer: bad argument");
  SMI_REQUIRE(layer >= 0 && layer < L->cfg.num_layers && stage >= 0 && stage <= 4, "smi_llm_debug_layer: layer %d stage %d", layer, stage);
  SMI_REQUIRE(!L->paged, "smi_llm_debug_layer needs a contiguous KV cache");
  SMI_HIP(hipDeviceSynchronize());
  graphs_flush(L);
  std::vector<RowDesc> rows(kMaxRows, RowDesc{0, 0, 0, 0});
  int maxpos = 0, ident = 1;
  for (int m = 0; m < M; ++m) {
    const int sl = rows_host[2 * m], ps = rows_host[2 * m + 1];
    SMI_REQUIRE(sl >= 0 && sl < L->cfg.max_slots && ps >= 0 && ps < L->cfg.max_positions, "smi_llm_debug_layer: row %d = (slot %d, pos %d)", m, sl, ps);
    rows[m] = RowDesc{sl, ps, 0, 0};
    maxpos = ps > maxpos ? ps : maxpos;
    ident &= sl == m;
  }
  SMI_HIP(hipMemcpy(L->rows, rows.data(), rows.size() * sizeof(RowDesc), hipMemcpyHostToDevice));
  L->B = M; L->identity_slots = ident; L->session = 0; L->started = 0;
  L->attn_seg = segs_for(maxpos + 1);
  float* src = nullptr;
  SMI_HIP(hipMalloc((void**)&src, (size_t)M * L->H * 4));
  int rc = SMI_OK;
  if (hipMemcpy(src, hidden_host, (size_t)M * L->H * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = SMI_EHIP; smi_set_error("smi_llm_debug_layer: upload failed"); }
  if (rc == SMI_OK) {
    hipLaunchKernelGGL(k_load_hidden, dim3((M + 3) / 4), dim3(256), 0, 0, src, L->KTh, M, (const float*)sec(L, SMI_LLM_LN1, layer), L->h, L->xs_h,
                       L->sspart, L->NTh * 4);
    if (hipGetLastError() != hipSuccess) { rc = SMI_EHIP; smi_set_error("smi_llm_debug_layer: k_load_hidden launch failed"); }
  }
  const bool fused = fuse_o_now(L, L->rows, M);
  const int last = (fused && stage == 2) ? KGU : KQKV + stage;   // one fused row: h + o_proj is formed by gate_up's prologue (read buffer 8)
  for (int k = KQKV; k <= last && rc == SMI_OK; ++k) rc = launch_one(L, k, layer, L->rows, M, nullptr, 0);
  if (hipDeviceSynchronize() != hipSuccess && rc == SMI_OK) { rc = SMI_EHIP; smi_set_error("smi_llm_debug_layer: kernels failed: %s", hipGetErrorString(hipGetLastError())); }
  (void)hipFree(src);
  return rc;
}

// Diagnostics: the stamp buffer of the last smi_llm_debug_stamps launch as it is (10 ns ticks of s_memrealtime; k_lm32 leaves
// four stamps per streamed group of its block 0: group start, MFMAs issued + partial sums written, reduce barrier passed, group done)
int smi_llm_debug_raw_stamps(smi_llm* L, unsigned long long* out, int n) {
  SMI_REQUIRE(L && out && n > 0 && n <= 4096 * 8, "smi_llm_debug_raw_stamps: bad argument");
  SMI_HIP(hipDeviceSynchronize());
  SMI_HIP(hipMemcpy(out, L->stamps, (size_t)n * 8, hipMemcpyDeviceToHost));
  return SMI_OK;
}

// Tests: the sampler alone on a caller's logits row (see sparkmi_debug.h).
int smi_llm_debug_sample(smi_llm* L, const float* logits_host, int n_rows, uint64_t seed, int use_bound, int32_t* tokens_out) {
  SMI_REQUIRE(L && tokens_out && n_rows >= 1 && n_rows <= kMaxRows, "smi_llm_debug_sample: bad argument");
  SMI_REQUIRE(L->do_sample, "smi_llm_debug_sample: smi_llm_set_sampling(do_sample = 1, ...) first");
  const int V = L->cfg.vocab_size;
  const int nblk = lm_blocks_for(L, n_rows);
  SMI_HIP(hipDeviceSynchronize());
  if (logits_host) {
    for (int m = 0; m < kMaxRows; ++m) SMI_HIP(hipMemcpy(L->logits + (size_t)m * V, logits_host, (size_t)V * 4, hipMemcpyHostToDevice));
    // what the lm_head blocks would have left: one maximum per block; here block j holds the j-th contiguous share of the row
    std::vector<float> pv((size_t)kMaxRows * L->lm_cap, -INFINITY);
    const int per = (V + nblk - 1) / nblk;
    for (int j = 0; j < nblk; ++j) {
      float mx = -INFINITY;
      for (int i = j * per; i < V && i < (j + 1) * per; ++i) mx = logits_host[i] > mx ? logits_host[i] : mx;
      for (int m = 0; m < kMaxRows; ++m) pv[(size_t)m * nblk + j] = mx;
    }
    SMI_HIP(hipMemcpy(L->pval, pv.data(), (size_t)kMaxRows * nblk * 4, hipMemcpyHostToDevice));
  }
  std::vector<RowDesc> rows(kMaxRows, RowDesc{0, 0, 0, 0});
  for (int m = 0; m < kMaxRows; ++m) { rows[m].slot = m; L->hctl.seqid[m] = m; }
  L->hctl.seed = seed;
  SMI_HIP(hipMemcpy(L->rows, rows.data(), rows.size() * sizeof(RowDesc), hipMemcpyHostToDevice));
  SMI_HIP(hipMemcpy(L->ctl, &L->hctl, sizeof(Ctl), hipMemcpyHostToDevice));
  SMI_HIP(hipMemset(L->cand_n, 0, kMaxRows * 4));
  SampleP sp;
  sp.logits = L->logits; sp.V = V; sp.top_k = L->top_k; sp.inv_temp = 1.0f / L->temperature;
  sp.top_p = L->top_p; sp.ctl = L->ctl; sp.rows = L->rows; sp.tok = L->tok;
  sp.pval = use_bound ? L->pval : nullptr; sp.nblk = nblk;
  sp.cand_v = L->cand_v; sp.cand_i = L->cand_i; sp.cand_n = L->cand_n;
  hipLaunchKernelGGL(k_sample_scan, dim3(kScanBlocks, n_rows), dim3(256), 0, 0, sp);
  SMI_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sample, dim3(n_rows), dim3(1024), 0, 0, sp);
  SMI_LAUNCH_CHECK();
  SMI_HIP(hipDeviceSynchronize());
  SMI_HIP(hipMemcpy(tokens_out, L->tok, (size_t)n_rows * 4, hipMemcpyDeviceToHost));
  L->started = 0;   // rows, controls and the lm_head partials no longer belong to a generation
  return SMI_OK;
}

// Diagnostics (SPARKMI_ENGINE_STAMPS=1): out[3][layers][16] microseconds since the first stamp of the last engine launch:
// CU 0 and the first head CU after the hand-offs A (h), B (q|k|v), C (attention), D (h_mid), E (act).
int smi_llm_engine_stamps(smi_llm* L, double* out, int cap) {
  SMI_REQUIRE(L && out && L->eng.enabled, "smi_llm_engine_stamps: engine not built");
  const int n = 3 * L->cfg.num_layers * 16;
  SMI_REQUIRE(cap >= n, "smi_llm_engine_stamps: out holds %d values, %d needed", cap, n);
  std::vector<unsigned long long> h((size_t)n);
  SMI_HIP(hipDeviceSynchronize());
  SMI_HIP(hipMemcpy(h.data(), L->eng.stamps, (size_t)n * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull;
  for (int i = 0; i < n; ++i) if (h[i] && h[i] < t0) t0 = h[i];
  for (int i = 0; i < n; ++i) out[i] = h[i] ? (double)(h[i] - t0) * 0.01 : -1.0;
  return SMI_OK;
}
#endif   // SMI_DIAG

}  // extern "C"
