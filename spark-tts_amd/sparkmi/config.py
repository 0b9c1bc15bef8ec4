"""Model configuration for the Spark-TTS hot path.

Every dimension is read from the checkpoint's own config files at load time
(``{model_dir}/LLM/config.json`` in HF format, ``{model_dir}/BiCodec/config.yaml``
with the ``audio_tokenizer`` section the reference reads at
``sparktts/models/bicodec.py:80-88``, and the top-level ``{model_dir}/config.yaml``
read at ``cli/SparkTTS.py:42-43``).  The ``spark_0p5b`` constructors hold the
published Spark-TTS-0.5B shape and are used only for synthetic-weight runs
(bench, tests) when no checkpoint is on disk.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field, asdict
from pathlib import Path
from typing import List, Optional

import yaml


@dataclass
class LLMConfig:
    """Qwen2 decoder-only LM (HF ``config.json`` key names)."""

    vocab_size: int = 166000
    hidden_size: int = 896
    num_hidden_layers: int = 24
    num_attention_heads: int = 14
    num_key_value_heads: int = 2
    intermediate_size: int = 4864
    rope_theta: float = 1000000.0
    rms_norm_eps: float = 1e-6
    tie_word_embeddings: bool = True
    max_position_embeddings: int = 32768
    eos_token_id: Optional[int] = None

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def q_dim(self) -> int:
        return self.num_attention_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.num_key_value_heads * self.head_dim

    @classmethod
    def from_json(cls, path) -> "LLMConfig":
        raw = json.loads(Path(path).read_text())
        keys = {f for f in cls.__dataclass_fields__}
        kw = {k: v for k, v in raw.items() if k in keys}
        if isinstance(kw.get("eos_token_id"), list):
            kw["eos_token_id"] = kw["eos_token_id"][0]
        return cls(**kw)

    def to_json(self, path) -> None:
        d = asdict(self)
        d["architectures"] = ["Qwen2ForCausalLM"]
        d["model_type"] = "qwen2"
        d["hidden_act"] = "silu"
        Path(path).write_text(json.dumps(d, indent=1))

    def validate(self) -> None:
        """The HIP kernels' shape contract (checked on the host before any launch)."""
        if self.head_dim != 64:
            raise ValueError(f"head_dim must be 64 (got {self.head_dim})")
        if self.hidden_size % 32 or self.intermediate_size % 32:
            raise ValueError("hidden_size and intermediate_size must be multiples of 32")
        if self.num_attention_heads % self.num_key_value_heads:
            raise ValueError("num_attention_heads must be a multiple of num_key_value_heads")

    def param_count(self) -> int:
        h, i, v = self.hidden_size, self.intermediate_size, self.vocab_size
        per_layer = (h * self.q_dim + self.q_dim) + 2 * (h * self.kv_dim + self.kv_dim) \
            + self.q_dim * h + 3 * h * i + 2 * h
        n = v * h + self.num_hidden_layers * per_layer + h
        if not self.tie_word_embeddings:
            n += v * h
        return n


def spark_0p5b_llm() -> LLMConfig:
    return LLMConfig()


def tiny_llm(vocab_size: int = 1003, layers: int = 3) -> LLMConfig:
    """Small config with the same structure (head_dim 64, GQA, odd vocab) for tests."""
    return LLMConfig(vocab_size=vocab_size, hidden_size=256, num_hidden_layers=layers,
                     num_attention_heads=4, num_key_value_heads=2, intermediate_size=608,
                     rope_theta=1000000.0, rms_norm_eps=1e-6)


@dataclass
class BiCodecConfig:
    """The detokenize half of ``BiCodec/config.yaml['audio_tokenizer']``."""

    # quantizer (FactorizedVectorQuantize)
    vq_input_dim: int = 1024
    codebook_size: int = 8192
    codebook_dim: int = 8
    # speaker_encoder (detokenize side only)
    spk_out_dim: int = 1024
    spk_latent_dim: int = 128
    spk_token_num: int = 32
    fsq_levels: List[int] = field(default_factory=lambda: [4, 4, 4, 4, 4, 4])
    fsq_num_quantizers: int = 1
    # prenet (Decoder: ConvNeXt stack)
    pre_input_channels: int = 1024
    pre_vocos_dim: int = 384
    pre_intermediate_dim: int = 2048
    pre_num_layers: int = 12
    pre_out_channels: int = 1024
    pre_condition_dim: int = 1024
    pre_sample_ratios: List[int] = field(default_factory=lambda: [1, 1])
    pre_use_tanh_at_final: bool = False
    # decoder (WaveGenerator)
    dec_input_channel: int = 1024
    dec_channels: int = 1536
    dec_rates: List[int] = field(default_factory=lambda: [8, 5, 4, 2])
    dec_kernel_sizes: List[int] = field(default_factory=lambda: [16, 11, 8, 4])

    @property
    def hop(self) -> int:
        h = 1
        for r in self.dec_rates:
            h *= r
        return h

    @classmethod
    def from_yaml(cls, path) -> "BiCodecConfig":
        raw = yaml.safe_load(Path(path).read_text())
        at = raw["audio_tokenizer"] if "audio_tokenizer" in raw else raw
        q, s, p, d = at["quantizer"], at["speaker_encoder"], at["prenet"], at["decoder"]
        return cls(
            vq_input_dim=q["input_dim"], codebook_size=q["codebook_size"],
            codebook_dim=q["codebook_dim"],
            spk_out_dim=s["out_dim"], spk_latent_dim=s["latent_dim"],
            spk_token_num=s["token_num"], fsq_levels=list(s["fsq_levels"]),
            fsq_num_quantizers=s["fsq_num_quantizers"],
            pre_input_channels=p["input_channels"], pre_vocos_dim=p["vocos_dim"],
            pre_intermediate_dim=p["vocos_intermediate_dim"],
            pre_num_layers=p["vocos_num_layers"], pre_out_channels=p["out_channels"],
            pre_condition_dim=p.get("condition_dim"),
            pre_sample_ratios=list(p.get("sample_ratios", [1, 1])),
            pre_use_tanh_at_final=bool(p.get("use_tanh_at_final", False)),
            dec_input_channel=d["input_channel"], dec_channels=d["channels"],
            dec_rates=list(d["rates"]), dec_kernel_sizes=list(d["kernel_sizes"]),
        )

    def to_yaml_dict(self) -> dict:
        """The same nested layout the reference's ``load_config`` consumer expects."""
        return {"audio_tokenizer": {
            "quantizer": {"input_dim": self.vq_input_dim, "codebook_size": self.codebook_size,
                          "codebook_dim": self.codebook_dim, "commitment": 0.25},
            "speaker_encoder": {"input_dim": 128, "out_dim": self.spk_out_dim,
                                "latent_dim": self.spk_latent_dim, "token_num": self.spk_token_num,
                                "fsq_levels": list(self.fsq_levels),
                                "fsq_num_quantizers": self.fsq_num_quantizers},
            "prenet": {"input_channels": self.pre_input_channels, "vocos_dim": self.pre_vocos_dim,
                       "vocos_intermediate_dim": self.pre_intermediate_dim,
                       "vocos_num_layers": self.pre_num_layers,
                       "out_channels": self.pre_out_channels,
                       "condition_dim": self.pre_condition_dim,
                       "sample_ratios": list(self.pre_sample_ratios),
                       "use_tanh_at_final": self.pre_use_tanh_at_final},
            "decoder": {"input_channel": self.dec_input_channel, "channels": self.dec_channels,
                        "rates": list(self.dec_rates), "kernel_sizes": list(self.dec_kernel_sizes)},
        }}

    def validate(self) -> None:
        if self.pre_sample_ratios != [1] * len(self.pre_sample_ratios):
            raise ValueError("prenet sample_ratios other than 1 are not on the hot path")
        if self.fsq_num_quantizers != 1:
            raise ValueError("speaker FSQ with more than one quantizer is not supported")
        if self.pre_out_channels != self.dec_input_channel or self.spk_out_dim != self.dec_input_channel:
            raise ValueError("prenet out / d-vector / decoder input dims must agree (bicodec.py:186)")
        if len(self.dec_rates) != len(self.dec_kernel_sizes):
            raise ValueError("decoder rates / kernel_sizes length mismatch")
        for k, s in zip(self.dec_kernel_sizes, self.dec_rates):
            if (k - s) % 2:
                raise ValueError("ConvTranspose1d needs (kernel - stride) even to upsample exactly")


def spark_0p5b_bicodec() -> BiCodecConfig:
    return BiCodecConfig()


def tiny_bicodec() -> BiCodecConfig:
    """Reduced dims, same structure: all four (k, s) transposed-conv shapes kept."""
    return BiCodecConfig(
        vq_input_dim=64, codebook_size=256, codebook_dim=8,
        spk_out_dim=64, spk_latent_dim=16, spk_token_num=8,
        pre_input_channels=64, pre_vocos_dim=32, pre_intermediate_dim=96, pre_num_layers=2,
        pre_out_channels=64, pre_condition_dim=64,
        dec_input_channel=64, dec_channels=128,
    )


@dataclass
class TopConfig:
    """``{model_dir}/config.yaml`` (keys used at cli/SparkTTS.py:43, audio_tokenizer.py:60-77)."""

    sample_rate: int = 16000
    ref_segment_duration: float = 6.0
    latent_hop_length: int = 320
    volume_normalize: bool = True

    @classmethod
    def from_yaml(cls, path) -> "TopConfig":
        raw = yaml.safe_load(Path(path).read_text()) or {}
        keys = {f for f in cls.__dataclass_fields__}
        return cls(**{k: v for k, v in raw.items() if k in keys})
