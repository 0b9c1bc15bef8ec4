"""Checkpoint access for the hot path: safetensors loading, weight-norm folding and a
deterministic synthetic-weight generator.

* Real checkpoints: ``load_llm_state`` / ``load_bicodec_state`` read the HF
  ``{model_dir}/LLM/*.safetensors`` and the reference's
  ``{model_dir}/BiCodec/model.safetensors`` (``sparktts/models/bicodec.py:78,100``).
* ``fold_weight_norm`` restates ``BiCodec.remove_weight_norm`` (``bicodec.py:213-221``):
  ``w = g * v / ||v||`` with the norm taken over every dim except dim 0 -- for a
  ``ConvTranspose1d`` weight ``(C_in, C_out, k)`` that is per *input* channel
  (``sparktts/modules/blocks/layers.py:24-29``).
* No checkpoint exists in this environment, so benches and tests draw weights from a
  counter-based numpy PRNG keyed by tensor name: both boxes regenerate identical
  tensors without shipping them and without depending on torch's RNG.
"""
from __future__ import annotations

import hashlib
from pathlib import Path
from typing import Callable, Dict, Iterable, List, Tuple

import numpy as np

from .config import BiCodecConfig, LLMConfig


# --------------------------------------------------------------------------- bf16 helpers
def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bit patterns (uint16). Inputs must be finite."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    return r.astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def round_bf16(x: np.ndarray) -> np.ndarray:
    """fp32 array whose every value is exactly representable in bf16."""
    return bf16_bits_to_f32(f32_to_bf16_bits(x)).reshape(x.shape)


# --------------------------------------------------------------------------- PRNG
def _rng(name: str, seed: int) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    key = np.frombuffer(h[:16], dtype=np.uint64)
    return np.random.Generator(np.random.Philox(key=key))


def normal(name: str, shape, std: float = 1.0, mean: float = 0.0, seed: int = 0) -> np.ndarray:
    a = _rng(name, seed).standard_normal(size=shape, dtype=np.float32)
    if std != 1.0:
        a *= np.float32(std)
    if mean != 0.0:
        a += np.float32(mean)
    return a


# --------------------------------------------------------------------------- LLM
def llm_tensor_names(cfg: LLMConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """HF Qwen2 parameter names and shapes (transformers modeling_qwen2)."""
    h, qd, kvd, it = cfg.hidden_size, cfg.q_dim, cfg.kv_dim, cfg.intermediate_size
    out = [("model.embed_tokens.weight", (cfg.vocab_size, h))]
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        out += [
            (p + "input_layernorm.weight", (h,)),
            (p + "self_attn.q_proj.weight", (qd, h)), (p + "self_attn.q_proj.bias", (qd,)),
            (p + "self_attn.k_proj.weight", (kvd, h)), (p + "self_attn.k_proj.bias", (kvd,)),
            (p + "self_attn.v_proj.weight", (kvd, h)), (p + "self_attn.v_proj.bias", (kvd,)),
            (p + "self_attn.o_proj.weight", (h, qd)),
            (p + "post_attention_layernorm.weight", (h,)),
            (p + "mlp.gate_proj.weight", (it, h)), (p + "mlp.up_proj.weight", (it, h)),
            (p + "mlp.down_proj.weight", (h, it)),
        ]
    out.append(("model.norm.weight", (h,)))
    if not cfg.tie_word_embeddings:
        out.append(("lm_head.weight", (cfg.vocab_size, h)))
    return out


class SyntheticLLM:
    """Name -> fp32 ndarray, generated on demand. Matrices are bf16-representable so the
    bf16 GPU arena and the fp32 oracle hold *identical* weight values."""

    def __init__(self, cfg: LLMConfig, seed: int = 0, embed_std: float = None,
                 w_std: float = None, bias_std: float = 0.02, bf16_exact: bool = True):
        # bf16_exact=False: full-precision fp32 matrices, i.e. what an fp32-saved checkpoint looks like (the arena
        # packer then rounds them; used to measure that rounding against the fp32 oracle)
        # 1.6/sqrt(hidden): layer outputs are O(1) per element, so greedy decoding of the random
        # net wanders over the vocabulary instead of locking onto one token (a fixed point would
        # make the KV-cache / position parity tests vacuous), and logits have std ~1.6.
        s = 1.6 / float(np.sqrt(cfg.hidden_size))
        self.cfg, self.seed = cfg, seed
        self.embed_std = s if embed_std is None else embed_std
        self.w_std = s if w_std is None else w_std
        self.bias_std = bias_std
        self.bf16_exact = bf16_exact
        self._shapes = dict(llm_tensor_names(cfg))

    def names(self) -> List[str]:
        return list(self._shapes)

    def __contains__(self, name: str) -> bool:
        return name in self._shapes

    def __getitem__(self, name: str) -> np.ndarray:
        shape = self._shapes[name]
        if name.endswith("layernorm.weight") or name == "model.norm.weight":
            return normal(name, shape, 0.1, 1.0, self.seed)
        if name.endswith(".bias"):
            return normal(name, shape, self.bias_std, 0.0, self.seed)
        std = self.embed_std if ("embed_tokens" in name or "lm_head" in name) else self.w_std
        w = normal(name, shape, std, 0.0, self.seed)
        return round_bf16(w) if self.bf16_exact else w


def load_llm_state(llm_dir) -> Dict[str, np.ndarray]:
    """All tensors of an HF Qwen2 checkpoint directory as fp32 numpy (via torch for bf16)."""
    import torch
    from safetensors import safe_open
    out = {}
    files = sorted(Path(llm_dir).glob("*.safetensors"))
    if not files:
        raise FileNotFoundError(f"no .safetensors under {llm_dir}")
    for f in files:
        with safe_open(str(f), framework="pt") as sf:
            for k in sf.keys():
                out[k] = sf.get_tensor(k).to(torch.float32).numpy()
    return out


# --------------------------------------------------------------------------- BiCodec
def _wn(sd: Dict[str, np.ndarray], prefix: str, shape, gain: float, seed: int,
        bias: bool = True, bias_std: float = 0.02) -> None:
    """weight_g / weight_v pair as torch.nn.utils.weight_norm stores them (dim=0)."""
    sd[prefix + ".weight_v"] = normal(prefix + ".weight_v", shape, 1.0, 0.0, seed)
    gshape = (shape[0],) + (1,) * (len(shape) - 1)
    g = gain * (1.0 + 0.1 * normal(prefix + ".weight_g", gshape, 1.0, 0.0, seed))
    sd[prefix + ".weight_g"] = g.astype(np.float32)
    if bias:
        sd[prefix + ".bias"] = normal(prefix + ".bias", (shape[0],), bias_std, 0.0, seed)


def bicodec_detok_state(cfg: BiCodecConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Synthetic state dict for exactly the parameters ``BiCodec.detokenize`` touches,
    under the reference module tree's own key names (checked against the reference's
    ``state_dict()`` keys by tests/golden/gen_golden.py)."""
    sd: Dict[str, np.ndarray] = {}
    N = lambda name, shape, std=1.0, mean=0.0: normal(name, shape, std, mean, seed)  # noqa: E731

    # quantizer: codebook + out_project (1x1 WNConv1d codebook_dim -> input_dim)
    sd["quantizer.codebook.weight"] = N("quantizer.codebook.weight",
                                        (cfg.codebook_size, cfg.codebook_dim), 1.0)
    _wn(sd, "quantizer.out_project", (cfg.vq_input_dim, cfg.codebook_dim, 1),
        gain=1.0, seed=seed)

    # speaker encoder: FSQ project_out (6 -> latent) + project (latent*token_num -> out_dim)
    nl = len(cfg.fsq_levels)
    sd["speaker_encoder.quantizer.project_out.weight"] = N(
        "speaker_encoder.quantizer.project_out.weight", (cfg.spk_latent_dim, nl), 0.6)
    sd["speaker_encoder.quantizer.project_out.bias"] = N(
        "speaker_encoder.quantizer.project_out.bias", (cfg.spk_latent_dim,), 0.05)
    kin = cfg.spk_latent_dim * cfg.spk_token_num
    sd["speaker_encoder.project.weight"] = N("speaker_encoder.project.weight",
                                             (cfg.spk_out_dim, kin), 1.0 / np.sqrt(kin))
    sd["speaker_encoder.project.bias"] = N("speaker_encoder.project.bias", (cfg.spk_out_dim,), 0.05)

    # prenet
    D, I, C = cfg.pre_vocos_dim, cfg.pre_intermediate_dim, cfg.pre_condition_dim

    def lin(prefix, o, i, std=None):
        sd[prefix + ".weight"] = N(prefix + ".weight", (o, i), std or 1.0 / np.sqrt(i))
        sd[prefix + ".bias"] = N(prefix + ".bias", (o,), 0.02)

    def lnorm(prefix):
        sd[prefix + ".weight"] = N(prefix + ".weight", (D,), 0.1, 1.0)
        sd[prefix + ".bias"] = N(prefix + ".bias", (D,), 0.05)

    def adanorm(prefix):
        # scale = 1 + small projection of the condition, shift = small projection
        sd[prefix + ".scale.weight"] = N(prefix + ".scale.weight", (D, C), 0.3 / np.sqrt(C))
        sd[prefix + ".scale.bias"] = N(prefix + ".scale.bias", (D,), 0.05, 1.0)
        sd[prefix + ".shift.weight"] = N(prefix + ".shift.weight", (D, C), 0.3 / np.sqrt(C))
        sd[prefix + ".shift.bias"] = N(prefix + ".shift.bias", (D,), 0.05)

    def vocos(prefix, nlayers, ada):
        sd[prefix + ".embed.weight"] = N(prefix + ".embed.weight", (D, D, 7), 1.0 / np.sqrt(7 * D))
        sd[prefix + ".embed.bias"] = N(prefix + ".embed.bias", (D,), 0.02)
        (adanorm if ada else lnorm)(prefix + ".norm")
        for j in range(nlayers):
            b = f"{prefix}.convnext.{j}"
            sd[b + ".gamma"] = N(b + ".gamma", (D,), 0.05, 0.5)
            sd[b + ".dwconv.weight"] = N(b + ".dwconv.weight", (D, 1, 7), 1.0 / np.sqrt(7))
            sd[b + ".dwconv.bias"] = N(b + ".dwconv.bias", (D,), 0.02)
            (adanorm if ada else lnorm)(b + ".norm")
            lin(b + ".pwconv1", I, D)
            lin(b + ".pwconv2", D, I)
        lnorm(prefix + ".final_layer_norm")

    lin("prenet.linear_pre", D, cfg.pre_input_channels)
    for i in range(len(cfg.pre_sample_ratios)):
        vocos(f"prenet.downsample.{i}.1", 2, ada=False)
    vocos("prenet.vocos_backbone", cfg.pre_num_layers, ada=C is not None)
    lin("prenet.linear", cfg.pre_out_channels, D)

    # decoder (WaveGenerator): model.0 conv7, model.1..n DecoderBlocks, Snake, conv7, tanh
    ch = cfg.dec_channels
    _wn(sd, "decoder.model.0", (ch, cfg.dec_input_channel, 7), gain=0.7, seed=seed)
    alpha = lambda name, c: N(name, (1, c, 1), 0.15, 1.0)  # noqa: E731
    nblk = len(cfg.dec_rates)
    for i, (k, s) in enumerate(zip(cfg.dec_kernel_sizes, cfg.dec_rates)):
        cin, cout = ch // 2 ** i, ch // 2 ** (i + 1)
        b = f"decoder.model.{i + 1}.block"
        sd[b + ".0.alpha"] = alpha(b + ".0.alpha", cin)
        # ConvTranspose1d weight (C_in, C_out, k); weight_norm dim=0 => per-input-channel norm.
        name = b + ".1"
        sd[name + ".weight_v"] = N(name + ".weight_v", (cin, cout, k), 1.0)
        g = 0.8 * np.sqrt(s / 2.0) * (1.0 + 0.1 * N(name + ".weight_g", (cin, 1, 1), 1.0))
        sd[name + ".weight_g"] = g.astype(np.float32)
        sd[name + ".bias"] = N(name + ".bias", (cout,), 0.02)
        for r in range(3):
            u = f"{b}.{r + 2}.block"
            sd[u + ".0.alpha"] = alpha(u + ".0.alpha", cout)
            _wn(sd, u + ".1", (cout, cout, 7), gain=0.7, seed=seed)
            sd[u + ".2.alpha"] = alpha(u + ".2.alpha", cout)
            _wn(sd, u + ".3", (cout, cout, 1), gain=0.35, seed=seed)
    clast = ch // 2 ** nblk
    sd[f"decoder.model.{nblk + 1}.alpha"] = alpha(f"decoder.model.{nblk + 1}.alpha", clast)
    _wn(sd, f"decoder.model.{nblk + 2}", (1, clast, 7), gain=0.35, seed=seed)
    return sd


def fold_weight_norm(sd: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Replace every ``X.weight_g`` / ``X.weight_v`` pair by ``X.weight`` exactly as
    ``torch.nn.utils.remove_weight_norm`` does (``torch._weight_norm(v, g, 0)``)."""
    import torch
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            base = k[: -len(".weight_v")]
            g = torch.from_numpy(np.ascontiguousarray(sd[base + ".weight_g"], dtype=np.float32))
            w = torch._weight_norm(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)), g, 0)
            out[base + ".weight"] = w.numpy()
        else:
            out[k] = v
    return out


def load_bicodec_state(bicodec_dir) -> Dict[str, np.ndarray]:
    """``BiCodec/model.safetensors`` as fp32 numpy (weight-norm NOT yet folded)."""
    import torch
    from safetensors import safe_open
    out = {}
    with safe_open(str(Path(bicodec_dir) / "model.safetensors"), framework="pt") as sf:
        for k in sf.keys():
            out[k] = sf.get_tensor(k).to(torch.float32).numpy()
    return out


# --------------------------------------------------------------------------- prompt encode
def wav2vec2_state(cfg, seed: int = 0) -> Dict[str, np.ndarray]:
    """Synthetic ``Wav2Vec2Model`` state dict under transformers' key names (layers beyond the last
    hidden-state tap are not generated: the tokenizer never runs them)."""
    sd: Dict[str, np.ndarray] = {}
    N = lambda name, shape, std=1.0, mean=0.0: normal("w2v." + name, shape, std, mean, seed)  # noqa: E731
    cin = 1
    for i, (c, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        p = f"feature_extractor.conv_layers.{i}"
        sd[p + ".conv.weight"] = N(p + ".conv.weight", (c, cin, k), 1.4 / np.sqrt(cin * k))
        if cfg.conv_bias:
            sd[p + ".conv.bias"] = N(p + ".conv.bias", (c,), 0.05)
        sd[p + ".layer_norm.weight"] = N(p + ".layer_norm.weight", (c,), 0.1, 1.0)
        sd[p + ".layer_norm.bias"] = N(p + ".layer_norm.bias", (c,), 0.05)
        cin = c
    H, I = cfg.hidden_size, cfg.intermediate_size
    sd["feature_projection.layer_norm.weight"] = N("feature_projection.layer_norm.weight", (cin,), 0.1, 1.0)
    sd["feature_projection.layer_norm.bias"] = N("feature_projection.layer_norm.bias", (cin,), 0.05)
    sd["feature_projection.projection.weight"] = N("feature_projection.projection.weight", (H, cin), 1.0 / np.sqrt(cin))
    sd["feature_projection.projection.bias"] = N("feature_projection.projection.bias", (H,), 0.02)
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    pc = "encoder.pos_conv_embed.conv"
    sd[pc + ".parametrizations.weight.original1"] = N(pc + ".v", (H, H // G, K), 1.0)
    g = np.sqrt(H * (H // G)) * 0.7 / np.sqrt(K * (H // G)) * (1.0 + 0.1 * N(pc + ".g", (1, 1, K), 1.0))
    sd[pc + ".parametrizations.weight.original0"] = g.astype(np.float32)
    sd[pc + ".bias"] = N(pc + ".bias", (H,), 0.02)
    for l in range(cfg.used_layers):
        p = f"encoder.layers.{l}"
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[f"{p}.attention.{nm}.weight"] = N(f"{p}.attention.{nm}.weight", (H, H), (1.6 if nm in ("q_proj", "k_proj") else 0.8) / np.sqrt(H))
            sd[f"{p}.attention.{nm}.bias"] = N(f"{p}.attention.{nm}.bias", (H,), 0.02)
        for nm in ("layer_norm", "final_layer_norm"):
            sd[f"{p}.{nm}.weight"] = N(f"{p}.{nm}.weight", (H,), 0.1, 1.0)
            sd[f"{p}.{nm}.bias"] = N(f"{p}.{nm}.bias", (H,), 0.05)
        sd[f"{p}.feed_forward.intermediate_dense.weight"] = N(f"{p}.ffn.in.weight", (I, H), 1.0 / np.sqrt(H))
        sd[f"{p}.feed_forward.intermediate_dense.bias"] = N(f"{p}.ffn.in.bias", (I,), 0.02)
        sd[f"{p}.feed_forward.output_dense.weight"] = N(f"{p}.ffn.out.weight", (H, I), 0.8 / np.sqrt(I))
        sd[f"{p}.feed_forward.output_dense.bias"] = N(f"{p}.ffn.out.bias", (H,), 0.02)
    return sd


def fold_pos_conv_weight_norm(sd: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """``encoder.pos_conv_embed.conv.weight`` from its weight-norm parametrization (dim=2:
    ``w = g * v / ||v||`` with the norm over dims (0, 1)), under either key spelling."""
    import torch
    out = dict(sd)
    pc = "encoder.pos_conv_embed.conv"
    for gk, vk in ((pc + ".parametrizations.weight.original0", pc + ".parametrizations.weight.original1"),
                   (pc + ".weight_g", pc + ".weight_v")):
        if gk in out:
            g = torch.from_numpy(np.ascontiguousarray(out.pop(gk), dtype=np.float32))
            v = torch.from_numpy(np.ascontiguousarray(out.pop(vk), dtype=np.float32))
            out[pc + ".weight"] = torch._weight_norm(v, g, 2).numpy()
    return out


def load_wav2vec2_state(w2v_dir) -> Dict[str, np.ndarray]:
    """``wav2vec2-large-xlsr-53/`` weights (safetensors or pytorch_model.bin) as fp32 numpy; a
    ``wav2vec2.`` prefix (``Wav2Vec2ForPreTraining`` checkpoints) is stripped."""
    import torch
    d = Path(w2v_dir)
    out = {}
    files = sorted(d.glob("*.safetensors"))
    if files:
        from safetensors import safe_open
        for f in files:
            with safe_open(str(f), framework="pt") as sf:
                for k in sf.keys():
                    out[k] = sf.get_tensor(k).to(torch.float32).numpy()
    elif (d / "pytorch_model.bin").exists():
        for k, v in torch.load(str(d / "pytorch_model.bin"), map_location="cpu", weights_only=True).items():
            out[k] = v.to(torch.float32).numpy()
    else:
        raise FileNotFoundError(f"no wav2vec2 weights under {d}")
    return {(k[len("wav2vec2."):] if k.startswith("wav2vec2.") else k): v for k, v in out.items()}


def bicodec_tok_state(tcfg, vq_input_dim: int, seed: int = 0) -> Dict[str, np.ndarray]:
    """Synthetic state dict for the parameters ``BiCodec.tokenize`` touches (bicodec.py:151-169),
    reference key names: ``encoder.*``, ``quantizer.in_project`` + codebook, and the analysis side
    of ``speaker_encoder.*`` (ECAPA-TDNN up to its latent, perceiver, FSQ project_in)."""
    sd: Dict[str, np.ndarray] = {}
    N = lambda name, shape, std=1.0, mean=0.0: normal(name, shape, std, mean, seed)  # noqa: E731
    D, I = tcfg.enc_vocos_dim, tcfg.enc_intermediate_dim

    def lin(prefix, o, i, std=None, bias=True):
        sd[prefix + ".weight"] = N(prefix + ".weight", (o, i), std or 1.0 / np.sqrt(i))
        if bias:
            sd[prefix + ".bias"] = N(prefix + ".bias", (o,), 0.02)

    def lnorm(prefix, d):
        sd[prefix + ".weight"] = N(prefix + ".weight", (d,), 0.1, 1.0)
        sd[prefix + ".bias"] = N(prefix + ".bias", (d,), 0.05)

    def vocos(prefix, cin, nlayers):
        sd[prefix + ".embed.weight"] = N(prefix + ".embed.weight", (D, cin, 7), 1.0 / np.sqrt(7 * cin))
        sd[prefix + ".embed.bias"] = N(prefix + ".embed.bias", (D,), 0.02)
        lnorm(prefix + ".norm", D)
        for j in range(nlayers):
            b = f"{prefix}.convnext.{j}"
            sd[b + ".gamma"] = N(b + ".gamma", (D,), 0.05, 0.5)
            sd[b + ".dwconv.weight"] = N(b + ".dwconv.weight", (D, 1, 7), 1.0 / np.sqrt(7))
            sd[b + ".dwconv.bias"] = N(b + ".dwconv.bias", (D,), 0.02)
            lnorm(b + ".norm", D)
            lin(b + ".pwconv1", I, D)
            lin(b + ".pwconv2", D, I)
        lnorm(prefix + ".final_layer_norm", D)

    vocos("encoder.encoder", tcfg.enc_input_channels, tcfg.enc_num_layers)
    for i in range(len(tcfg.enc_sample_ratios)):
        vocos(f"encoder.downsample.{i}.1", D, 2)
    lin("encoder.project", tcfg.enc_out_channels, D)
    sd["quantizer.codebook.weight"] = N("quantizer.codebook.weight", (tcfg.codebook_size, tcfg.codebook_dim), 1.0)
    _wn(sd, "quantizer.in_project", (tcfg.codebook_dim, vq_input_dim, 1), gain=1.0, seed=seed)

    # ECAPA-TDNN (ecapa_tdnn.py:152-208), only what produces `latent`
    C, F, W = tcfg.ecapa_channels, tcfg.num_mels, tcfg.ecapa_channels // 8
    se = "speaker_encoder.speaker_encoder"

    def bn(prefix, c):
        sd[prefix + ".weight"] = N(prefix + ".weight", (c,), 0.1, 1.0)
        sd[prefix + ".bias"] = N(prefix + ".bias", (c,), 0.05)
        sd[prefix + ".running_mean"] = N(prefix + ".running_mean", (c,), 0.1, 0.2)
        sd[prefix + ".running_var"] = np.abs(N(prefix + ".running_var", (c,), 0.1, 0.5)).astype(np.float32) + 0.1

    def conv(prefix, o, i, k, std=None):
        sd[prefix + ".weight"] = N(prefix + ".weight", (o, i, k), std or 1.4 / np.sqrt(i * k))
        sd[prefix + ".bias"] = N(prefix + ".bias", (o,), 0.05)

    conv(se + ".layer1.conv", C, F, 5)
    bn(se + ".layer1.bn", C)
    for li in (2, 3, 4):
        b = f"{se}.layer{li}.se_res2block"
        conv(b + ".0.conv", C, C, 1)
        bn(b + ".0.bn", C)
        for j in range(7):
            conv(f"{b}.1.convs.{j}", W, W, 3)
            bn(f"{b}.1.bns.{j}", W)
        conv(b + ".2.conv", C, C, 1)
        bn(b + ".2.bn", C)
        lin(b + ".3.linear1", 128, C)
        lin(b + ".3.linear2", C, 128)
    conv(se + ".conv", tcfg.ecapa_out, 3 * C, 1)

    # perceiver resampler (perceiver_encoder.py:297-350)
    ps = "speaker_encoder.perceiver_sampler"
    L, inner = tcfg.spk_latent_dim, tcfg.perceiver_heads * tcfg.perceiver_dim_head
    lin(ps + ".proj_context", L, tcfg.ecapa_out)
    sd[ps + ".latents"] = N(ps + ".latents", (tcfg.spk_token_num, L), 1.0)
    for i in range(tcfg.perceiver_depth):
        a = f"{ps}.layers.{i}.0"
        lin(a + ".to_q", inner, L, std=1.5 / np.sqrt(L), bias=False)
        lin(a + ".to_kv", 2 * inner, L, std=1.5 / np.sqrt(L), bias=False)
        lin(a + ".to_out", L, inner, bias=False)
        f = f"{ps}.layers.{i}.1"
        lin(f + ".0", 2 * tcfg.ff_inner, L)
        lin(f + ".2", L, tcfg.ff_inner)
    sd[ps + ".norm.gamma"] = N(ps + ".norm.gamma", (L,), 0.1, 1.0)
    nl = len(tcfg.fsq_levels)
    sd["speaker_encoder.quantizer.project_in.weight"] = N("speaker_encoder.quantizer.project_in.weight", (nl, L), 1.2 / np.sqrt(L))
    sd["speaker_encoder.quantizer.project_in.bias"] = N("speaker_encoder.quantizer.project_in.bias", (nl,), 0.1)
    return sd
