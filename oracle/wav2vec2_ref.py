"""CPU oracle (TEST INFRASTRUCTURE ONLY -- never imported by the product path) for the wav2vec2
feature path of ``BiCodecTokenizer.extract_wav2vec2_features``
(``sparktts/models/audio_tokenizer.py:85-100``): processor normalisation, ``Wav2Vec2Model`` with
``output_hidden_states`` and the mix ``(hs[11] + hs[14] + hs[16]) / 3``.

The arithmetic is third-party: ``transformers`` (pinned 4.46.2 at the reference's
``requirements.txt:12``; 5.15.0 is installed here) ``models/wav2vec2/modeling_wav2vec2.py`` (``MW``)
and ``feature_extraction_wav2vec2.py`` (``FW``).  Restated with plain torch fp32 ops for the
layer-norm / stable-layer-norm variant (xlsr-53); pinned by ``tests/golden/tok_*.npz``, generated
from ``transformers.Wav2Vec2Model`` itself (tests/golden/gen_golden_tok.py).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F


def zero_mean_unit_var(wav: np.ndarray) -> np.ndarray:
    """FW ``zero_mean_unit_var_norm`` (no attention mask): (x - mean) / sqrt(var + 1e-7), float32."""
    x = np.asarray(wav, dtype=np.float32)
    return ((x - x.mean()) / np.sqrt(x.var() + 1e-7)).astype(np.float32)


class Wav2Vec2Ref:
    def __init__(self, cfg, state: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in state.items()}
        pc = "encoder.pos_conv_embed.conv"
        if pc + ".weight" not in self.w:   # weight_norm(dim=2), MW Wav2Vec2PositionalConvEmbedding
            for gk, vk in ((pc + ".parametrizations.weight.original0", pc + ".parametrizations.weight.original1"),
                           (pc + ".weight_g", pc + ".weight_v")):
                if gk in self.w:
                    self.w[pc + ".weight"] = torch._weight_norm(self.w[vk], self.w[gk], 2)

    # MW Wav2Vec2LayerNormConvLayer.forward
    def feature_encoder(self, x: torch.Tensor) -> torch.Tensor:
        c, w = self.cfg, self.w
        h = x[:, None, :]
        for i, s in enumerate(c.conv_stride):
            p = f"feature_extractor.conv_layers.{i}"
            h = F.conv1d(h, w[p + ".conv.weight"], w.get(p + ".conv.bias"), stride=s)
            h = F.layer_norm(h.transpose(-2, -1), (h.shape[1],), w[p + ".layer_norm.weight"], w[p + ".layer_norm.bias"], 1e-5)
            h = F.gelu(h.transpose(-2, -1))
        return h                                    # (B, C, T)

    # MW Wav2Vec2FeatureProjection.forward
    def feature_projection(self, feats: torch.Tensor) -> torch.Tensor:
        c, w = self.cfg, self.w
        h = feats.transpose(1, 2)
        h = F.layer_norm(h, (h.shape[-1],), w["feature_projection.layer_norm.weight"], w["feature_projection.layer_norm.bias"], c.layer_norm_eps)
        return F.linear(h, w["feature_projection.projection.weight"], w["feature_projection.projection.bias"])

    # MW Wav2Vec2PositionalConvEmbedding.forward (+ Wav2Vec2SamePadLayer: drop the last frame for even kernels)
    def pos_conv(self, h: torch.Tensor) -> torch.Tensor:
        c, w = self.cfg, self.w
        y = F.conv1d(h.transpose(1, 2), w["encoder.pos_conv_embed.conv.weight"], w["encoder.pos_conv_embed.conv.bias"],
                     padding=c.num_conv_pos_embeddings // 2, groups=c.num_conv_pos_embedding_groups)
        if c.num_conv_pos_embeddings % 2 == 0:
            y = y[:, :, :-1]
        return F.gelu(y).transpose(1, 2)

    # MW Wav2Vec2EncoderLayerStableLayerNorm.forward, Wav2Vec2Attention.forward, eager_attention_forward
    def layer(self, h: torch.Tensor, l: int) -> torch.Tensor:
        c, w = self.cfg, self.w
        p = f"encoder.layers.{l}"
        B, T, H = h.shape
        nh, hd = c.num_attention_heads, c.head_dim
        x = F.layer_norm(h, (H,), w[p + ".layer_norm.weight"], w[p + ".layer_norm.bias"], c.layer_norm_eps)
        q = F.linear(x, w[p + ".attention.q_proj.weight"], w[p + ".attention.q_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
        k = F.linear(x, w[p + ".attention.k_proj.weight"], w[p + ".attention.k_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
        v = F.linear(x, w[p + ".attention.v_proj.weight"], w[p + ".attention.v_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
        a = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
        a = F.softmax(a, dim=-1)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, H)
        h = h + F.linear(o, w[p + ".attention.out_proj.weight"], w[p + ".attention.out_proj.bias"])
        x = F.layer_norm(h, (H,), w[p + ".final_layer_norm.weight"], w[p + ".final_layer_norm.bias"], c.layer_norm_eps)
        x = F.gelu(F.linear(x, w[p + ".feed_forward.intermediate_dense.weight"], w[p + ".feed_forward.intermediate_dense.bias"]))
        return h + F.linear(x, w[p + ".feed_forward.output_dense.weight"], w[p + ".feed_forward.output_dense.bias"])

    @torch.no_grad()
    def hidden_states(self, wav_norm, stages: dict = None) -> List[torch.Tensor]:
        """hs[i] = input of encoder layer i (MW Wav2Vec2EncoderStableLayerNorm.forward), i = 0..used_layers."""
        x = torch.as_tensor(np.asarray(wav_norm, dtype=np.float32))
        if x.ndim == 1:
            x = x[None]
        feats = self.feature_encoder(x)
        h = self.feature_projection(feats)
        if stages is not None:
            stages["conv_feats"], stages["projected"] = feats, h
        h = h + self.pos_conv(h)
        hs = [h]
        for l in range(self.cfg.used_layers):
            h = self.layer(h, l)
            hs.append(h)
        return hs

    @torch.no_grad()
    def features(self, wav: np.ndarray, stages: dict = None) -> torch.Tensor:
        """audio_tokenizer.py:85-100: (1, T, hidden) mix of the three tapped hidden states."""
        hs = self.hidden_states(zero_mean_unit_var(wav), stages)
        a, b, c = self.cfg.taps
        return (hs[a] + hs[b] + hs[c]) / 3
