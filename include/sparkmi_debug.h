/*
 * sparkmi_debug.h -- diagnostics and test entry points of the LLM half.  NOT part of the product ABI: these symbols exist
 * only in libsparkmi_diag.so (spark-tts_amd/csrc built with -DSMI_DIAG), which also exports everything sparkmi.h declares and
 * -- unlike libsparkmi.so -- honours the SPARKMI_* environment switches listed in DESIGN.md 6.1.  Loaded by tools/, by
 * bench.py's per-kernel probes and by the tests that look inside a step (tests/test_llm_ops_gpu.py, test_engine_gpu.py).
 */
#ifndef SPARKMI_DEBUG_H
#define SPARKMI_DEBUG_H

#include "sparkmi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-kernel timing probe used by bench.py: launches ONLY the named decode-step kernel of `layer`
 * `iters` times on `stream` (inputs are whatever the scratch holds), bracketed by HIP events, and
 * returns the average milliseconds per launch.  kernel: 0 qkv, 1 attn, 2 o_proj, 3 gate_up,
 * 4 down, 5 lm_head, 6 finalize, 7 = the whole decode step (graph or eager as configured), 8 = all layers of one step
 * (one launch of the one-row engine where it applies, else the layer kernels in order; needs a new prefill afterwards).
 * 16 + k (k = 0..4): layer kernel k timed in sequence -- (iters whole layers) minus (the same layers
 * without k) -- so that it finds the L2 state its producers leave, as inside the decode graph. */
int smi_llm_time_kernel(smi_llm* h, int kernel, int layer, int iters, float* ms_avg, void* stream);
/* Diagnostics: one launch of a decode-step GEMM kernel (ids as above, GEMM kernels only) with in-kernel
 * s_memrealtime phase stamps; out[0..7) = mean over blocks of (stamp i - earliest stamp 0) in microseconds,
 * out[7] = shader clock in MHz (tools/stamps.py).  kernel + 32: the layer's earlier kernels (and the previous layer's down_proj)
 * run first, un-stamped, so the stamped kernel finds the cache state it finds inside a decode step (tools/prefetch_stamps.py).
 * Needs a started generation. */
int smi_llm_debug_stamps(smi_llm* h, int kernel, int layer, double* out);
/* One-row decode engine (csrc/smi_eng.h).  With ONE live sequence in slot 0 (bf16 KV, contiguous cache, contexts up to
 * 1024 tokens) the layers of a decode step run as one persistent launch -- one workgroup per CU, weights streamed through
 * LDS rings by LDS-DMA, the five all-to-all edges of a layer handed over inside the launch -- instead of four dependent
 * launches per layer; the arithmetic (every product, accumulator chain and addition order) is the launch path's, so the
 * tokens are the same bits.  OPT-IN (SPARKMI_ENGINE=1 at create, or smi_llm_set_engine(h, 1), which builds it on first use:
 * 0.8 GB of re-packed weights): on MI355X at the 0.5B shape the five in-launch hand-offs of a layer cost more than the four
 * kernel boundaries they replace (26.3 vs 23.4 us per layer), so the launch path stays the default.
 *   smi_llm_engine: *enabled = 1 when one-row steps take the engine; info[4] = {CUs, images per wave and layer, LDS bytes,
 *                   built}; why = a one-line reason / description.
 *   smi_llm_set_engine: runtime switch between the engine and the launch path (A/B, tests); synchronises the device; on = 1
 *                   where the engine does not apply (f32 / paged KV, odd shapes, small device) leaves it off (see `why`).
 *   smi_llm_engine_plan: host-only check of the static work plan for `ncu` CUs (no GPU call): every weight image placed
 *                   exactly once, stream order = job order; stats[8] = {images per wave max, per wave and phase max, parts per
 *                   CU and phase max, jobs per wave max, images per CU min, max, LDS bytes, images per layer}.
 *   smi_llm_engine_stamps: diagnostics, SPARKMI_ENGINE_STAMPS=1: out[3][layers][16] microseconds of the last engine launch
 *                   (wave 0 of CU 0, wave 0 of the first head CU, wave 7 of CU 0; after the hand-offs h, q|k|v, attention, h_mid, act).
 * A hand-off that does not complete within SPARKMI_ENGINE_TIMEOUT_MS (default 500) ends the launch; the next call that
 * synchronises (smi_llm_get_tokens / _status / _all_done) returns SMI_EHIP. */
int smi_llm_engine(smi_llm* h, int32_t* enabled, int32_t* info, char* why, int n);
int smi_llm_set_engine(smi_llm* h, int on);
int smi_llm_engine_plan(const smi_llm_cfg* cfg, int ncu, int32_t* stats);
int smi_llm_engine_stamps(smi_llm* h, double* out, int cap);
/* Tests: synchronises and copies the residual row (hidden_size floats) of row 0 as the last step left it. */
int smi_llm_debug_hidden(smi_llm* h, float* out_host, int n);
/* Tests / debugging: synchronises and copies one scratch buffer as raw bytes (buffers 0..4 and 6 hold one entry per live row:
 * M rows after smi_llm_debug_layer).  what: 0 q [M][q_dim] f32, 1 / 2 / 3 the operand
 * triples of o_proj / down_proj / the next norm ([K / 32][3][4][M][16 B]), 4 the residual rows, 5 the engine's
 * granules [2][per buffer] u64 {tag << 32 | f32 bits}, 6 partial sums of squares [hidden / 4], 7 K rows of layer 0, slot 0,
 * kv head 0 (bf16), 8 h + o_proj of the fused one-row path. */
int smi_llm_debug_read(smi_llm* h, int what, void* out_host, size_t cap, size_t* got);

/* Op-level tests of the LLM half: ONE decoder layer's kernels, stage by stage, on caller-given rows -- launched by the functions a
 * real step launches them with (same kernel choices per row count, launch geometry, prologue / epilogue fusions), so the classes
 * of transformers' modeling_qwen2.py can be checked one at a time (tests/test_llm_ops_gpu.py on tests/golden/llm_ops.npz):
 * Qwen2RMSNorm + q/k/v_proj + apply_rotary_pos_emb (MQ:247-252, 91-135), eager_attention_forward (MQ:150-173), o_proj + residual,
 * Qwen2MLP (MQ:46-48).
 *   smi_llm_debug_set_kv / _get_kv: write / read cache rows of (layer, slot) as fp32 [n][num_kv_heads][64] in transformers' dim
 *     order (keys rotated, as a cache holds them); the cache's own dtype and row order are converted on the host.
 *   smi_llm_debug_layer: rows_host = M (slot, pos) pairs, hidden_host [M][hidden] = the residual rows entering `layer`; runs its
 *     kernels up to and including `stage`: 0 QKV (+ bias, RoPE, K/V append: read q with smi_llm_debug_read(0), K/V with _get_kv),
 *     1 attention (buffer 1: the o_proj operand triples, head-interleaved k tiles), 2 o_proj + residual (buffer 4; one fused row:
 *     buffer 8), 3 gate_up + SwiGLU (buffer 2: act triples), 4 down_proj + residual (buffer 4).  smi_llm_debug_read's buffers
 *     0..4 and 6 then hold M rows.  Contiguous KV cache only.  Ends the current generation. */
int smi_llm_debug_set_kv(smi_llm* h, int layer, int slot, int pos0, int n, const float* k_host, const float* v_host);
int smi_llm_debug_get_kv(smi_llm* h, int layer, int slot, int pos0, int n, float* k_host, float* v_host);
int smi_llm_debug_layer(smi_llm* h, int layer, int M, const int32_t* rows_host, const float* hidden_host, int stage);
/* Diagnostics: the raw stamp buffer (u64 s_memrealtime ticks, 10 ns) after smi_llm_debug_stamps; n entries. */
int smi_llm_debug_raw_stamps(smi_llm* h, unsigned long long* out, int n);
/* Tests: the sampler alone (k_sample_scan + k_sample, exactly as a decode step launches them) on a caller's logits row --
 * the reference's default decoding chain, cli/SparkTTS.py:166-168,197-204 -> transformers' TemperatureLogitsWarper ->
 * TopKLogitsWarper -> TopPLogitsWarper -> multinomial.  logits_host [vocab_size] is replicated to every row (null: the rows
 * of the previous call stay), n_rows rows draw one token each from the streams (seed; token index 0, sequence number = row),
 * tokens_out [n_rows].  use_bound = 1: the one-pass candidate collection with the top_k-th largest block maximum as the bound
 * (block j = the j-th contiguous share of the row, as many blocks as the lm_head launch of n_rows rows leaves); 0: the exact
 * radix selection.  Parameters come from smi_llm_set_sampling.  Synchronises; ends the current generation. */
int smi_llm_debug_sample(smi_llm* h, const float* logits_host, int n_rows, uint64_t seed, int use_bound, int32_t* tokens_out);

#ifdef __cplusplus
}
#endif
#endif /* SPARKMI_DEBUG_H */
