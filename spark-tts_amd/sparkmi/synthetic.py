"""Builds a complete synthetic Spark-TTS model directory (no checkpoint exists offline):

    {dir}/config.yaml                      top-level keys (cli/SparkTTS.py:42-43)
    {dir}/LLM/config.json, model.safetensors (bf16), generation_config.json, tokenizer files
    {dir}/BiCodec/config.yaml, model.safetensors (weight_g / weight_v kept, like the real file)

so that ``SparkTTS(model_dir)`` exercises the same loading code a real checkpoint would.  The
tokenizer is byte-level (every string encodes) with the Spark control tokens as special tokens
and the ``<|bicodec_*|>`` tokens as ordinary added tokens (they must survive
``skip_special_tokens=True`` for the reference's regex at cli/SparkTTS.py:213-220 to see them).
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Tuple

import numpy as np
import torch
import yaml

from .config import BiCodecConfig, LLMConfig, tiny_bicodec
from .pipeline_text import TASK_TOKEN_MAP
from .config_tok import TokCfg, Wav2Vec2Cfg, tiny_tok, tiny_wav2vec2
from .weights import SyntheticLLM, bicodec_detok_state, bicodec_tok_state, wav2vec2_state

CONTROL_TOKENS = (
    list(TASK_TOKEN_MAP.values())
    + ["<|start_content|>", "<|end_content|>", "<|start_global_token|>", "<|end_global_token|>",
       "<|start_semantic_token|>", "<|end_semantic_token|>", "<|start_style_label|>", "<|end_style_label|>"]
    + [f"<|gender_{i}|>" for i in range(2)] + [f"<|pitch_label_{i}|>" for i in range(5)]
    + [f"<|speed_label_{i}|>" for i in range(5)])


def build_tokenizer(n_global: int, n_semantic: int):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    alphabet = pre_tokenizers.ByteLevel.alphabet()
    vocab = {ch: i for i, ch in enumerate(sorted(alphabet))}
    tok = Tokenizer(models.BPE(vocab=vocab, merges=[]))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    tok.decoder = decoders.ByteLevel()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|im_end|>", pad_token="<|endoftext|>")
    fast.add_special_tokens({"additional_special_tokens": CONTROL_TOKENS})
    fast.add_tokens([f"<|bicodec_global_{i}|>" for i in range(n_global)]
                    + [f"<|bicodec_semantic_{i}|>" for i in range(n_semantic)])
    return fast


def make_model_dir(path, llm_cfg: LLMConfig = None, voc_cfg: BiCodecConfig = None, seed: int = 0,
                   w2v_cfg: Wav2Vec2Cfg = None, tok_cfg: TokCfg = None, with_prompt_encoder: bool = True) -> Tuple[LLMConfig, BiCodecConfig]:
    from safetensors.torch import save_file
    path = Path(path)
    voc_cfg = voc_cfg or tiny_bicodec()
    n_global = int(np.prod(voc_cfg.fsq_levels))
    tok = build_tokenizer(n_global, voc_cfg.codebook_size)
    if llm_cfg is None:
        from .config import tiny_llm
        llm_cfg = tiny_llm(vocab_size=len(tok))
    if llm_cfg.vocab_size < len(tok):
        raise ValueError(f"LLM vocab {llm_cfg.vocab_size} < tokenizer size {len(tok)}")
    llm_cfg.eos_token_id = tok.eos_token_id
    (path / "LLM").mkdir(parents=True, exist_ok=True)
    (path / "BiCodec").mkdir(parents=True, exist_ok=True)
    (path / "config.yaml").write_text(yaml.safe_dump(
        {"sample_rate": 16000, "ref_segment_duration": 6, "latent_hop_length": voc_cfg.hop, "volume_normalize": True}))
    llm_cfg.to_json(path / "LLM" / "config.json")
    (path / "LLM" / "generation_config.json").write_text(json.dumps({"eos_token_id": tok.eos_token_id}))
    tok.save_pretrained(str(path / "LLM"))
    syn = SyntheticLLM(llm_cfg, seed=seed)
    save_file({n: torch.from_numpy(syn[n]).to(torch.bfloat16 if syn[n].ndim == 2 else torch.float32)
               for n in syn.names()}, str(path / "LLM" / "model.safetensors"))
    ycfg = voc_cfg.to_yaml_dict()
    state = dict(bicodec_detok_state(voc_cfg, seed=seed))
    if with_prompt_encoder:
        # the tokenize half (voice cloning): wav2vec2 directory + encoder / speaker-encoder weights in the same BiCodec file
        w2v_cfg = w2v_cfg or tiny_wav2vec2()
        tok_cfg = tok_cfg or tiny_tok()
        extra = tok_cfg.to_yaml_dict()
        ycfg["audio_tokenizer"]["encoder"] = extra["encoder"]
        ycfg["audio_tokenizer"]["mel_params"] = extra["mel_params"]
        ycfg["audio_tokenizer"]["speaker_encoder"].update(extra["speaker_encoder_extra"], input_dim=tok_cfg.num_mels)
        state.update(bicodec_tok_state(tok_cfg, voc_cfg.vq_input_dim, seed=seed))
        wdir = path / "wav2vec2-large-xlsr-53"
        wdir.mkdir(parents=True, exist_ok=True)
        w2v_cfg.to_json(wdir / "config.json")
        (wdir / "preprocessor_config.json").write_text(json.dumps({
            "do_normalize": True, "feature_extractor_type": "Wav2Vec2FeatureExtractor", "feature_size": 1,
            "padding_side": "right", "padding_value": 0, "return_attention_mask": True, "sampling_rate": 16000}))
        save_file({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in wav2vec2_state(w2v_cfg, seed=seed).items()},
                  str(wdir / "model.safetensors"))
    (path / "BiCodec" / "config.yaml").write_text(yaml.safe_dump(ycfg))
    save_file({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}, str(path / "BiCodec" / "model.safetensors"))
    return llm_cfg, voc_cfg
