/*
 * sparkmi.h -- C ABI of libsparkmi.so: the MI355X (gfx950) Spark-TTS inference hot path.
 *
 * The reference (arghyasur1991/Spark-TTS) is pure Python and has no FFI of its own; its seams
 * for this path are Python call sites.  Each entry point below names the reference call it
 * stands behind, so a maintainer can bind it (ctypes stub in INTEGRATION.md):
 *
 *   smi_llm_*   <- AutoModelForCausalLM.generate(...)            cli/SparkTTS.py:197-204
 *                  (Qwen2 forward: transformers modeling_qwen2.py, pinned 4.46.2 at requirements.txt:12)
 *   smi_voc_*   <- BiCodec.detokenize(semantic, global)          sparktts/models/bicodec.py:171-189
 *                  via BiCodecTokenizer.detokenize               sparktts/models/audio_tokenizer.py:132-146
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types.  Every function returns an int:
 *     0 = SMI_OK, negative = SMI_E*.  smi_last_error() returns a thread-local message.
 *   - "dev" pointers are device (HBM) addresses on the current HIP device, "host" pointers are
 *     ordinary host memory.  `stream` is a hipStream_t passed as void* (0 = default stream).
 *   - The caller owns the weight arenas (they must outlive the handle).  The library owns the
 *     handle, its KV cache and scratch (allocated at create, freed at destroy).
 *   - A handle is bound to the device current at create and may be used by one host thread at
 *     a time.  All launches go to the caller's stream; nothing synchronises unless stated.
 *   - There is no CPU fallback anywhere behind this ABI.
 *   - libsparkmi.so reads NO environment variable.  Timing probes, in-kernel stamps, scratch dumps, A/B switches and the
 *     experimental one-row engine live in libsparkmi_diag.so (the same sources built with -DSMI_DIAG; it exports this
 *     header's symbols plus include/sparkmi_debug.h's) -- tools, the bench probes and the op-level tests load that one.
 */
#ifndef SPARKMI_H
#define SPARKMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMI_OK 0
#define SMI_EINVAL (-1)   /* bad argument / shape outside the kernel contract */
#define SMI_EHIP (-2)     /* a HIP runtime call failed (message has hipGetErrorString) */
#define SMI_ENOMEM (-3)   /* device allocation failed */
#define SMI_ESTATE (-4)   /* call sequence violated (e.g. decode before prefill) */

#define SMI_ABI_VERSION 4   /* 2: eos id LISTS, per-sequence sampler streams; 3: smi_llm_cfg.wd_plain (the W_down tile order is data, not environment);
                              4: diagnostics (timing probes, stamps, scratch dumps, the one-row engine) left this header for sparkmi_debug.h /
                                 libsparkmi_diag.so; the product library reads no environment variable */
#define SMI_MAX_EOS 4      /* eos ids per generation (HF stops on ANY id of generation_config.eos_token_id) */
#define SMI_MAX_ROWS 64   /* rows (= concurrent sequences, or prompt tokens per prefill chunk) per step */

int smi_version(void);
const char* smi_last_error(void);
/* Fills name[0..n) with the device's gcnArchName; fails unless it is a gfx950 part. */
int smi_device_check(char* name, int n);

/* ------------------------------------------------------------------------------------------
 * LLM: Qwen2 decoder-only LM, greedy or sampled generation with a KV cache.
 * ---------------------------------------------------------------------------------------- */
typedef struct smi_llm smi_llm;

typedef struct smi_llm_cfg {
  int32_t vocab_size;        /* config.json: vocab_size */
  int32_t hidden_size;       /* multiple of 32 */
  int32_t num_layers;
  int32_t num_heads;         /* query heads */
  int32_t num_kv_heads;
  int32_t head_dim;          /* must be 64 */
  int32_t intermediate_size; /* multiple of 32 */
  int32_t max_slots;         /* concurrent sequences, 1..SMI_MAX_ROWS */
  int32_t max_positions;     /* tokens (prompt + generated) per sequence */
  int32_t kv_dtype;          /* 0 = bf16 KV cache, 1 = f32 KV cache */
  int32_t use_graph;         /* 1 = replay the decode step as a hipGraph */
  float rms_eps;
  /* Paged KV cache (the functional analogue of TensorRT-LLM's paged KV under in-flight batching,
   * runtime/triton_trtllm/run.sh:50-65): kv_page_tokens = 0 gives every slot max_positions contiguous tokens
   * (max_slots x max_positions reserved).  A power of two in 16..1024 that divides max_positions makes the cache a
   * POOL of kv_pages pages of that many tokens; a sequence holds only the pages its length needs (allocated as it
   * grows, returned at smi_llm_retire / the next prefill), so many live sequences do not each reserve the longest
   * context.  A call that needs more pages than are free fails with SMI_ENOMEM and changes nothing.  Tokens are the
   * same either way. */
  int32_t kv_page_tokens;
  int32_t kv_pages;
  /* Layout of the WD (down_proj) tiles IN THE ARENA the caller packed: 0 (default) = row-part-major [q:4][k8:4][r:4][8],
   * 1 = the plain tile order of the other matrices (kept for A/B: sparkmi/arena.py packs it under SPARKMI_WD_PLAIN=1 and
   * sets this field).  The library reads the layout from here, never from the environment: an arena and the handle that
   * reads it cannot disagree silently. */
  int32_t wd_plain;
  /* Exact-weights verification mode: 1 = the arena holds every matrix as fp32 [N][K] row-major (same row / column orders as
   * the bf16 tiles; 2x the bytes) and every GEMM of the path runs as an exact fp32 multiply-add chain over k
   * (v_mfma_f32_16x16x4_f32), activations exact as always.  For checkpoints SAVED in fp32 -- the published Spark-TTS-0.5B
   * LLM/model.safetensors is: the reference loads it as saved (cli/SparkTTS.py:48-51) and the default bf16 arena ROUNDS it
   * (logits move ~1e-2) -- this mode reproduces the fp32 PyTorch CPU path's greedy tokens (north_star's acceptance sentence
   * on such a checkpoint).  Opt-in and slow (~4x the step time at one row); 0 (default) = bf16 weights, north_star's arithmetic. */
  int32_t weights_exact;
} smi_llm_cfg;

/* Arena sections.  The arena is one device buffer the caller fills (see sparkmi/arena.py):
 *   matrices: bf16, rows grouped into 16-row x 32-col tiles stored [n_tile][k_tile][k8:4][n:16][8]
 *             (one tile = 1 KiB = exactly one wave64 x 16-byte load = one MFMA 16x16x32 A operand);
 *   QKV rows: q heads, then k heads, then v heads; inside each q/k head the 64 rows are ordered
 *             (0,32,1,33,...) so a RoPE pair sits in adjacent rows; QKV bias in the same order;
 *   WO columns (the attention output it multiplies): 32-column k tiles head-interleaved -- tile (half * num_heads + head)
 *             holds dims 32*half .. 32*half+31 of that head (sparkmi/arena.py: o_proj_col_perm);
 *   GATE_UP rows: gate and up interleaved (g0,u0,g1,u1,...);
 *   WD (down_proj) tiles: the 64 pieces of a tile are stored row-part-major [q:4][k8:4][r:4][8] (row n = 4q + r) instead
 *             of [k8:4][n:16][8]: the row-split down_proj kernels (4 or 8 rows of a tile per block) then read whole
 *             128-byte lines (sparkmi/arena.py: pack_tiles(row_parts=True));
 *   LM_HEAD: vocab padded up to a multiple of 16 rows with zeros (also the embedding table);
 *   norms/bias: f32;  ROPE: float2 (cos,sin) [max_positions][head_dim/2].                      */
enum smi_llm_section {
  SMI_LLM_LN1 = 0, SMI_LLM_WQKV, SMI_LLM_BQKV, SMI_LLM_WO, SMI_LLM_LN2, SMI_LLM_WGU, SMI_LLM_WD, /* per layer */
  SMI_LLM_FINAL_NORM, SMI_LLM_LM_HEAD, SMI_LLM_ROPE, SMI_LLM_TAG,                               /* layer = 0 */
  SMI_LLM_NUM_SECTIONS
};
/* SMI_LLM_TAG: 256 bytes the packer fills with a smi_llm_arena_tag.  smi_llm_create reads it back from the device and refuses
 * an arena that was packed for another ABI version, other dimensions or another W_down tile order than the config it is given
 * says: a packed arena and the handle that streams it cannot disagree silently (an arena re-used under a different
 * SPARKMI_WD_PLAIN setting used to give wrong logits with no error). */
typedef struct smi_llm_arena_tag {
  char magic[8];             /* "SMIARENA" */
  int32_t abi_version;       /* SMI_ABI_VERSION of the packer */
  int32_t wd_plain;          /* = smi_llm_cfg.wd_plain the matrices were packed with */
  int32_t vocab_size, hidden_size, num_layers, num_heads, num_kv_heads, intermediate_size, max_positions;
  int32_t weights_exact;     /* = smi_llm_cfg.weights_exact: fp32 [N][K] matrices instead of bf16 tiles */
  int32_t reserved[52];
} smi_llm_arena_tag;
/* Total arena size in bytes for this config (0 on invalid config). */
size_t smi_llm_arena_bytes(const smi_llm_cfg* cfg);
/* Byte offset and byte size of one section; returns SMI_EINVAL for a bad section/layer. */
int smi_llm_arena_section(const smi_llm_cfg* cfg, int section, int layer, size_t* offset, size_t* bytes);

int smi_llm_create(const smi_llm_cfg* cfg, const void* arena_dev, size_t arena_bytes, smi_llm** out);
int smi_llm_destroy(smi_llm* h);

/* Start B sequences (slots 0..B-1): run the prompts through the model (KV cache filled) and emit
 * each sequence's first new token (generate()'s prefill forward + argmax).
 *   ids_host:  [B][P_max] int64 prompt ids, right-padded;  lens_host: [B] prompt lengths (>= 1).
 *   eos_ids_host / n_eos: a sequence stops counting tokens after emitting ANY of these ids, as HF generate()
 *              does with every id of generation_config.json's eos_token_id (cli/SparkTTS.py:197-204 passes none
 *              itself); 0 <= n_eos <= SMI_MAX_EOS, n_eos = 0: never stop.                          */
int smi_llm_prefill(smi_llm* h, const int64_t* ids_host, const int32_t* lens_host, int B, int P_max,
                    const int64_t* eos_ids_host, int n_eos, void* stream);
/* Token selection for the following prefill/decode calls.  do_sample = 0: greedy argmax (lowest id
 * on ties, like torch.argmax).  do_sample = 1: the reference's default at cli/SparkTTS.py:197-204 --
 * temperature, then top-k (1..256), then nucleus top-p, one multinomial draw per step from a
 * Philox stream keyed by (seed; the sequence's admission number, its own token index): reproducible per
 * seed whatever else is in the batch, statistically (not bitwise) equivalent to transformers' sampler. */
int smi_llm_set_sampling(smi_llm* h, int do_sample, float temperature, int top_k, float top_p, uint64_t seed);
/* Run n_steps more decode steps for all B sequences (finished ones keep stepping; their
 * tokens are not counted).  Asynchronous. */
int smi_llm_decode(smi_llm* h, int n_steps, void* stream);
/* Synchronises the stream; *all_done = 1 when every sequence has emitted eos. */
int smi_llm_all_done(smi_llm* h, int* all_done, void* stream);
/* Synchronises; copies the generated ids: out_host [B][cap] int64 (row b holds lens_host[b] ids,
 * the eos included when one was emitted). */
int smi_llm_get_tokens(smi_llm* h, int64_t* out_host, int32_t* lens_host, int cap, void* stream);
/* Continuous (in-flight) batching: sequences join and leave between decode steps, each in its own KV slot;
 * the functional analogue of the reference's Triton / TensorRT-LLM in-flight batching
 * (runtime/triton_trtllm/run.sh:50-65).  smi_llm_session_begin starts an empty session; smi_llm_admit
 * prefills n new prompts into free slots (returned in slots_out) and emits their first token without
 * touching the live sequences; smi_llm_decode then steps every live sequence; smi_llm_slot_tokens reads one
 * sequence's tokens so far and whether it has produced eos; smi_llm_retire frees its slot.  A sequence's
 * tokens do not depend on what else is live, was admitted with it or shares its call: rows are independent in every decode
 * kernel and every decode path (row-grouped GEMMs, chain-split down_proj, one-row kernels) sums a row in the same order, and the
 * kernels its PROMPT rows run through are chosen from ITS OWN length alone (up to SMI_MAX_ROWS rows: the decode kernels; more:
 * the prefill GEMM family, whose few-row and many-row shapes sum o_proj / down_proj in the same K segments in the same order),
 * each class of a call in its own pass -- bit for bit, with either KV cache type (tests/test_fullsize_gpu.py: 32 of 32 rows,
 * prompts of 3 .. 600 tokens in one call against every sequence alone, K rows compared).  Round 3 chose the prompt kernels
 * from the call's TOTAL rows; with the bf16 cache a last-bit difference between two kernels could then move a K/V element to
 * the neighbouring bf16 value and flip a near-tie arg-max between two batch compositions. */
int smi_llm_session_begin(smi_llm* h, const int64_t* eos_ids_host, int n_eos, void* stream);
int smi_llm_admit(smi_llm* h, const int64_t* ids_host, const int32_t* lens_host, int n, int P_max, int32_t* slots_out, void* stream);
int smi_llm_retire(smi_llm* h, int slot, void* stream);
int smi_llm_slot_tokens(smi_llm* h, int slot, int64_t* out_host, int cap, int32_t* n_out, int32_t* finished, void* stream);
/* Several sequences leave at once with no host round trip (the device row list is compacted in place), and the tokens of
 * several slots in one round trip: out_host [n][cap], n_out [n], finished [n].  A retired slot's history stays readable
 * until a later admission reuses the slot. */
int smi_llm_retire_many(smi_llm* h, const int32_t* slots, int n, void* stream);
int smi_llm_slots_tokens(smi_llm* h, const int32_t* slots, int n, int64_t* out_host, int cap, int32_t* n_out, int32_t* finished,
                         void* stream);
/* count_host / finished_host [SMI_MAX_ROWS]: tokens emitted and eos flag per KV slot, one round trip. */
int smi_llm_status(smi_llm* h, int32_t* count_host, int32_t* finished_host, void* stream);
/* Test/teacher-forcing entry: feeds ids_host[0..S) at positions 0..S-1 of slot 0 (cache reset) and
 * writes every position's logits to logits_dev [S][vocab_size] f32. */
int smi_llm_forward_logits(smi_llm* h, const int64_t* ids_host, int S, float* logits_dev, void* stream);
/* Steps generated so far per sequence (including the prefill token), and the KV bytes per token. */
int smi_llm_steps(smi_llm* h);
/* Paged KV cache: pages in the pool and pages currently free (both 0 when the cache is not paged). */
int smi_llm_kv_pages(smi_llm* h, int32_t* total, int32_t* free_pages);
/* ------------------------------------------------------------------------------------------
 * Vocoder: BiCodec.detokenize (codebook lookup, d-vector, ConvNeXt prenet, WaveGenerator).
 * ---------------------------------------------------------------------------------------- */
typedef struct smi_voc smi_voc;

typedef struct smi_voc_cfg {
  int32_t vq_input_dim, codebook_size, codebook_dim;
  int32_t spk_out_dim, spk_latent_dim, spk_token_num, fsq_dims; /* fsq_dims = len(fsq_levels) */
  int32_t fsq_levels[8];
  int32_t pre_input_channels, pre_dim, pre_inter, pre_layers, pre_out_channels, pre_cond_dim;
  int32_t pre_num_down;      /* len(sample_ratios); every ratio must be 1 */
  int32_t pre_tanh_final;
  int32_t dec_in, dec_channels, dec_nblocks;
  int32_t dec_rates[8], dec_ksizes[8];
  int32_t max_batch;         /* utterances per forward */
  int32_t max_frames;        /* semantic frames per utterance */
  /* 0 (default): the dense conv / linear stack runs on the bf16 matrix pipe with both operands split into two bf16 planes
   * (w x ~= w_hi x_hi + w_hi x_mid + w_mid x_hi, fp32 accumulate): 5e-5 max-abs from the fp32 waveform at the 0.5B shape,
   * north_star's bound being 1e-3.  1: every contraction on the exact-fp32 matrix pipe (verification mode, 1/16 of the rate).
   * The arena packing of the affected weights differs (smi_voc_arena_entry reports the kind). */
  int32_t exact_fp32;
} smi_voc_cfg;

/* The vocoder arena is a flat f32 buffer of tensors in the order smi_voc_arena_entry enumerates
 * (name = the reference state_dict key after remove_weight_norm, or a derived packed tensor). */
int smi_voc_arena_count(const smi_voc_cfg* cfg);
/* info: int32[6] = {packing kind (0 raw f32 copy, 1 Conv1d/Linear weight [Cout][Cin][K], 2 ConvTranspose1d
 * weight [Cin][Cout][K], 3 / 4 the same two as bf16 planes), Cout, Cin, K, stride, padding}.  Kinds 1 / 2 are laid out
 * [phase][cout_tile:32][tap][cin_group:8][lane:64][4 f32] = v_mfma_f32_32x32x2_f32 A operands; kinds 3 / 4
 * [phase][cout_tile:32][tap][cin_step:16][plane:2 (hi, mid)][lane:64][8 bf16] = v_mfma_f32_32x32x16_bf16 A operands,
 * lane l holding row l & 31, channels 8 (l >> 5) .. + 7 of the step (sparkmi/bicodec.py: pack_conv / pack_conv_b). */
int smi_voc_arena_entry(const smi_voc_cfg* cfg, int index, char* name, int name_cap,
                        size_t* offset, size_t* bytes, int32_t* info);
size_t smi_voc_arena_bytes(const smi_voc_cfg* cfg);

int smi_voc_create(const smi_voc_cfg* cfg, const void* arena_dev, size_t arena_bytes, smi_voc** out);
int smi_voc_destroy(smi_voc* h);
/* sem_dev [B][T_max] int64 semantic ids (row b valid for lens_host[b] frames), glob_dev [B][Ntok]
 * int32 global ids, wav_dev [B][hop*T_max] f32 (samples beyond hop*lens[b] are zeroed).
 * Each row's result equals an un-padded B=1 run of that row. */
int smi_voc_forward(smi_voc* h, const int64_t* sem_dev, const int32_t* lens_host, const int32_t* glob_dev,
                    int B, int T_max, float* wav_dev, void* stream);
/* Test entry: copies an internal activation (after `stage`) of the last forward to out_dev. */
int smi_voc_debug_stage(smi_voc* h, int stage, float* out_dev, size_t max_floats, size_t* n_floats, void* stream);
/* Per-kernel timing probe (see smi_llm_time_kernel): stage index into the launch list of the last
 * forward; returns avg ms and the kernel's FLOPs per launch. */
int smi_voc_num_launches(smi_voc* h);
int smi_voc_time_launch(smi_voc* h, int index, int iters, float* ms_avg, double* flops, char* name, int name_cap, void* stream);

/* One block of the vocoder on caller tensors -- the op-level test entry points.  The launches are built by the very functions
 * smi_voc_forward builds them with (same kernels, launch geometry and epilogue fusions), so the reference's own layer classes
 * can be checked one at a time: ResidualUnit (sparktts/modules/blocks/layers.py:51-67), DecoderBlock
 * (sparktts/modules/encoder_decoder/wave_generator.py:29-53), ConvNeXtBlock with LayerNorm or AdaLayerNorm
 * (sparktts/modules/blocks/vocos.py:26-110).  A block's little arena is described exactly like the vocoder's
 * (smi_voc_block_arena_entry: names are "L." + the layer's own state_dict keys, "cat:a|b" = rows concatenated). */
enum { SMI_VOC_BLOCK_RESUNIT = 0, SMI_VOC_BLOCK_DECBLOCK = 1, SMI_VOC_BLOCK_CONVNEXT = 2 };
typedef struct smi_voc_block_cfg {
  int32_t kind;        /* SMI_VOC_BLOCK_* */
  int32_t C;           /* RESUNIT: channels; DECBLOCK: input channels; CONVNEXT: model dim */
  int32_t Cout;        /* DECBLOCK: output channels */
  int32_t K, S;        /* DECBLOCK: ConvTranspose1d kernel size and stride (padding (K - S) / 2) */
  int32_t dil;         /* RESUNIT: dilation of the 7-tap conv (1, 3 or 9) */
  int32_t I;           /* CONVNEXT: intermediate dim */
  int32_t cond_dim;    /* CONVNEXT: 0 = LayerNorm; > 0 = AdaLayerNorm on a [B][cond_dim] condition */
  int32_t exact_fp32;  /* as smi_voc_cfg.exact_fp32 */
} smi_voc_block_cfg;
int smi_voc_block_arena_count(const smi_voc_block_cfg* cfg);
size_t smi_voc_block_arena_bytes(const smi_voc_block_cfg* cfg);
int smi_voc_block_arena_entry(const smi_voc_block_cfg* cfg, int index, char* name, int name_cap,
                              size_t* offset, size_t* bytes, int32_t* info);
/* x_dev [B][C][L] f32 (contiguous), y_dev [B][C or Cout][L or L * S].  The vocoder applies a block's LEADING Snake in the
 * producer's epilogue, so: RESUNIT takes x_dev and xs_dev = snake(x, block.0.alpha); DECBLOCK takes only xs_dev =
 * snake(x, block.0.alpha); CONVNEXT takes x_dev (and cond_dev [B][cond_dim] for AdaLayerNorm).  lens_host (may be null =
 * all rows full) masks ragged rows as in smi_voc_forward.  Synchronises the stream before returning. */
int smi_voc_block_run(const smi_voc_block_cfg* cfg, const void* arena_dev, size_t arena_bytes, const float* x_dev,
                      const float* xs_dev, const float* cond_dev, const int32_t* lens_host, int B, int L, float* y_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Prompt encoder (voice cloning): BiCodecTokenizer.tokenize (sparktts/models/audio_tokenizer.py:85-130)
 * = zero-mean/unit-variance wav -> wav2vec2 (layers below the last tapped hidden state) -> mean of
 * three hidden states -> BiCodec Encoder -> cosine VQ arg-max (semantic ids); reference clip -> mel
 * -> ECAPA-TDNN latent -> perceiver resampler -> FSQ (global ids)  (sparktts/models/bicodec.py:151-169).
 * One utterance per call, as the reference's tokenize().
 * ---------------------------------------------------------------------------------------- */
typedef struct smi_enc smi_enc;

typedef struct smi_enc_cfg {
  /* wav2vec2 (transformers Wav2Vec2Config; layer-norm / stable-layer-norm variant) */
  int32_t w2v_conv_dim, w2v_nconv;
  int32_t w2v_kernel[8], w2v_stride[8];
  int32_t w2v_hidden, w2v_layers /* = max(taps): layers actually run */, w2v_heads, w2v_inter;
  int32_t w2v_pos_k, w2v_pos_groups;
  int32_t w2v_taps[3];       /* hidden_states indices averaged (audio_tokenizer.py:96-98) */
  float w2v_eps;
  /* BiCodec encoder + quantizer */
  int32_t enc_in, enc_dim, enc_inter, enc_layers, enc_out, enc_num_down;
  int32_t codebook_size, codebook_dim;
  /* mel (bicodec.py:200-211) */
  int32_t n_fft, win_length, hop_length, num_mels;
  /* speaker encoder analysis side */
  int32_t ecapa_channels, ecapa_out, spk_latent, spk_tokens, fsq_dims;
  int32_t fsq_levels[8];
  int32_t perc_depth, perc_heads, perc_ff_inner;
  int32_t max_samples;       /* longest prompt wav (samples) */
  int32_t max_ref_samples;   /* longest reference clip (samples) */
  /* 0 (default): wav2vec2's transformer projections and the BiCodec encoder's ConvNeXt stack run on the bf16-split matrix
   * pipe (as smi_voc_cfg.exact_fp32 describes); 1: every contraction on the exact-fp32 matrix pipe.  Token ids are arg-max /
   * rounding decisions: they agree between the two modes wherever the decision margin exceeds the split's 2^-17 noise. */
  int32_t exact_fp32;
} smi_enc_cfg;

/* Arena: like the vocoder's (same entry info and conv packing).  Names are the reference / transformers
 * state_dict keys ("w2v." + key for wav2vec2), "cat:a|b|c" row concatenations, or derived tensors the host
 * computes: "bnscale:<bn prefix>" / "bnshift:<bn prefix>" (BatchNorm eval affine), "transpose:<key>",
 * "mel.dft" ([2*(n_fft/2+1)][n_fft] windowed cos / -sin DFT basis) and "mel.fb" ([num_mels][n_fft/2+1]). */
int smi_enc_arena_count(const smi_enc_cfg* cfg);
int smi_enc_arena_entry(const smi_enc_cfg* cfg, int index, char* name, int name_cap,
                        size_t* offset, size_t* bytes, int32_t* info);
size_t smi_enc_arena_bytes(const smi_enc_cfg* cfg);
int smi_enc_create(const smi_enc_cfg* cfg, const void* arena_dev, size_t arena_bytes, smi_enc** out);
int smi_enc_destroy(smi_enc* h);
/* wav_dev [n_samples] f32 (volume-normalised, NOT yet zero-mean/unit-var), ref_dev [n_ref] f32 reference clip;
 * sem_dev [>= frames] int64 out, glob_dev [spk_tokens] int32 out; *n_frames = wav2vec2 frames produced. */
int smi_enc_forward(smi_enc* h, const float* wav_dev, int n_samples, const float* ref_dev, int n_ref,
                    int64_t* sem_dev, int32_t* glob_dev, int* n_frames, void* stream);
/* Test entry: copies a named internal activation of the last forward ("feat", "z", "mel", "ecapa_latent",
 * "perceiver", "hs0", "conv_feats", "input_values", ...) to out_dev as [rows][cols] f32; dims[2] = {rows, cols}. */
int smi_enc_debug_stage(smi_enc* h, const char* name, float* out_dev, size_t max_floats, int32_t* dims, void* stream);
int smi_enc_num_launches(smi_enc* h);
int smi_enc_time_launch(smi_enc* h, int index, int iters, float* ms_avg, double* flops, char* name, int name_cap, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARKMI_H */
