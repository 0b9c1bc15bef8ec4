"""ORACLE (test infrastructure, never the product path): CPU restatement of
``BiCodec.detokenize`` (``sparktts/models/bicodec.py:171-189``) -- semantic + global token ids
to a 16 kHz waveform -- in plain fp32 torch-CPU ops.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  Every function cites the reference lines it follows.  Pinning:
``tests/golden/gen_golden.py`` ran the reference's own modules (imported from
``/root/reference`` in the build container) on seeded inputs and committed the results under
``tests/golden/``; ``tests/test_oracle_bicodec.py`` checks this file against them.

The state dict uses the reference module tree's key names after ``remove_weight_norm``
(``bicodec.py:213-221``), i.e. ``sparkmi.weights.fold_weight_norm`` output.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, name) -> torch.Tensor:
    return torch.as_tensor(np.asarray(sd[name]), dtype=torch.float32)


def snake(x: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    """blocks/layers.py:33-39: x + (alpha + 1e-9)^-1 * sin(alpha x)^2, alpha (1, C, 1)."""
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def fsq_codes(indices: torch.Tensor, levels: Sequence[int]) -> torch.Tensor:
    """fsq/finite_scalar_quantization.py:143-162: index -> per-dim level -> (lvl - L//2)/(L//2).
    indices (...,) int -> (..., len(levels)) float."""
    lv = torch.as_tensor(list(levels), dtype=torch.long)
    basis = torch.cumprod(torch.as_tensor([1] + list(levels[:-1]), dtype=torch.long), dim=0)
    lvl = (indices.to(torch.long)[..., None] // basis) % lv
    half = lv // 2
    return (lvl - half).to(torch.float32) / half.to(torch.float32)


class BiCodecDetokRef:
    def __init__(self, cfg, folded_state: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.sd = folded_state

    # ---------------------------------------------------------------- token -> feature
    def vq_detokenize(self, semantic: torch.Tensor) -> torch.Tensor:
        """vq/factorized_vector_quantize.py:154-167: raw codebook rows (not normalised),
        transpose, 1x1 conv out_project.  (B, T) -> (B, input_dim, T)."""
        z = F.embedding(semantic.to(torch.long), _t(self.sd, "quantizer.codebook.weight"))
        return F.conv1d(z.transpose(1, 2), _t(self.sd, "quantizer.out_project.weight"),
                        _t(self.sd, "quantizer.out_project.bias"))

    def d_vector(self, global_tokens: torch.Tensor) -> torch.Tensor:
        """speaker/speaker_encoder.py:107-112 + fsq/residual_fsq.py:112-199.
        global_tokens (B, 1, Ntok) (what bicodec.py:184 receives) -> (B, out_dim)."""
        idx = global_tokens.transpose(1, 2)                      # (B, Ntok, 1)
        codes = fsq_codes(idx.squeeze(-1), self.cfg.fsq_levels)   # (B, Ntok, 6); scales == 1 for q=0
        zq = F.linear(codes, _t(self.sd, "speaker_encoder.quantizer.project_out.weight"),
                      _t(self.sd, "speaker_encoder.quantizer.project_out.bias"))  # (B, Ntok, latent)
        zq = zq.transpose(1, 2)                                   # (B, latent, Ntok)
        x = zq.reshape(zq.shape[0], -1)                           # index = d * Ntok + t
        return F.linear(x, _t(self.sd, "speaker_encoder.project.weight"),
                        _t(self.sd, "speaker_encoder.project.bias"))

    # ---------------------------------------------------------------- prenet
    def _norm(self, x_btc: torch.Tensor, prefix: str, cond: Optional[torch.Tensor]) -> torch.Tensor:
        """LayerNorm(eps 1e-6) or AdaLayerNorm (blocks/vocos.py:87-110)."""
        D = x_btc.shape[-1]
        if prefix + ".scale.weight" in self.sd:
            scale = F.linear(cond, _t(self.sd, prefix + ".scale.weight"), _t(self.sd, prefix + ".scale.bias"))
            shift = F.linear(cond, _t(self.sd, prefix + ".shift.weight"), _t(self.sd, prefix + ".shift.bias"))
            x = F.layer_norm(x_btc, (D,), eps=1e-6)
            return x * scale.unsqueeze(1) + shift.unsqueeze(1)
        return F.layer_norm(x_btc, (D,), _t(self.sd, prefix + ".weight"), _t(self.sd, prefix + ".bias"), eps=1e-6)

    def _convnext(self, x: torch.Tensor, p: str, cond) -> torch.Tensor:
        """blocks/vocos.py:65-84.  x (B, C, T)."""
        sd = self.sd
        r = x
        x = F.conv1d(x, _t(sd, p + ".dwconv.weight"), _t(sd, p + ".dwconv.bias"), padding=3, groups=x.shape[1])
        x = self._norm(x.transpose(1, 2), p + ".norm", cond)
        x = F.linear(x, _t(sd, p + ".pwconv1.weight"), _t(sd, p + ".pwconv1.bias"))
        x = F.gelu(x)  # nn.GELU() default = exact erf form
        x = F.linear(x, _t(sd, p + ".pwconv2.weight"), _t(sd, p + ".pwconv2.bias"))
        x = _t(sd, p + ".gamma") * x
        return r + x.transpose(1, 2)

    def _vocos(self, x: torch.Tensor, p: str, nlayers: int, cond) -> torch.Tensor:
        """blocks/vocos.py:324-335.  (B, C, T) -> (B, T, C)."""
        sd = self.sd
        x = F.conv1d(x, _t(sd, p + ".embed.weight"), _t(sd, p + ".embed.bias"), padding=3)
        x = self._norm(x.transpose(1, 2), p + ".norm", cond).transpose(1, 2)
        for j in range(nlayers):
            x = self._convnext(x, f"{p}.convnext.{j}", cond)
        return F.layer_norm(x.transpose(1, 2), (x.shape[1],), _t(sd, p + ".final_layer_norm.weight"),
                            _t(sd, p + ".final_layer_norm.bias"), eps=1e-6)

    def prenet(self, z_q: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
        """encoder_decoder/feat_decoder.py:78-94.  SamplingBlock with both ratios 1
        (blocks/samper.py:79-100) returns conv_res + skip1 + skip2 = 3 * x."""
        sd, cfg = self.sd, self.cfg
        x = F.linear(z_q.transpose(1, 2), _t(sd, "prenet.linear_pre.weight"), _t(sd, "prenet.linear_pre.bias"))
        for i in range(len(cfg.pre_sample_ratios)):
            x = x.transpose(1, 2)          # SamplingBlock.forward transposes to (B, C, T)
            x = x + x + x                  # conv_res + skip1_res + skip2_res, all equal to x
            x = self._vocos(x, f"prenet.downsample.{i}.1", 2, None)   # -> (B, T, C)
        x = x.transpose(1, 2)
        x = self._vocos(x, "prenet.vocos_backbone", cfg.pre_num_layers, d)
        x = F.linear(x, _t(sd, "prenet.linear.weight"), _t(sd, "prenet.linear.bias")).transpose(1, 2)
        if cfg.pre_use_tanh_at_final:
            x = torch.tanh(x)
        return x

    # ---------------------------------------------------------------- wave generator
    def _res_unit(self, x: torch.Tensor, p: str, dil: int) -> torch.Tensor:
        """blocks/layers.py:51-67 (length-preserving padding; the crop branch is never taken)."""
        sd = self.sd
        y = snake(x, _t(sd, p + ".0.alpha"))
        y = F.conv1d(y, _t(sd, p + ".1.weight"), _t(sd, p + ".1.bias"), dilation=dil, padding=3 * dil)
        y = snake(y, _t(sd, p + ".2.alpha"))
        y = F.conv1d(y, _t(sd, p + ".3.weight"), _t(sd, p + ".3.bias"))
        return x + y

    def _decoder_block(self, x: torch.Tensor, b: str, k: int, s: int) -> torch.Tensor:
        """encoder_decoder/wave_generator.py:29-53: Snake -> ConvTranspose1d(k, stride s, padding (k - s) // 2) -> three
        ResidualUnits (dilations 1, 3, 9)."""
        sd = self.sd
        x = snake(x, _t(sd, b + ".0.alpha"))
        x = F.conv_transpose1d(x, _t(sd, b + ".1.weight"), _t(sd, b + ".1.bias"), stride=s, padding=(k - s) // 2)
        for r, dil in enumerate((1, 3, 9)):
            x = self._res_unit(x, f"{b}.{r + 2}.block", dil)
        return x

    def wave_generator(self, x: torch.Tensor, stages: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        """encoder_decoder/wave_generator.py:56-88."""
        sd, cfg = self.sd, self.cfg
        x = F.conv1d(x, _t(sd, "decoder.model.0.weight"), _t(sd, "decoder.model.0.bias"), padding=3)
        if stages is not None:
            stages.append(x)
        for i, (k, s) in enumerate(zip(cfg.dec_kernel_sizes, cfg.dec_rates)):
            x = self._decoder_block(x, f"decoder.model.{i + 1}.block", k, s)
            if stages is not None:
                stages.append(x)
        n = len(cfg.dec_rates)
        x = snake(x, _t(sd, f"decoder.model.{n + 1}.alpha"))
        x = F.conv1d(x, _t(sd, f"decoder.model.{n + 2}.weight"), _t(sd, f"decoder.model.{n + 2}.bias"), padding=3)
        return torch.tanh(x)

    # ---------------------------------------------------------------- top level
    @torch.no_grad()
    def detokenize(self, semantic: torch.Tensor, global_tokens: torch.Tensor,
                   stages: Optional[dict] = None) -> torch.Tensor:
        """bicodec.py:183-187.  semantic (B, T) int, global_tokens (B, 1, Ntok) int -> (B, 1, hop*T)."""
        z_q = self.vq_detokenize(semantic)
        d = self.d_vector(global_tokens)
        x = self.prenet(z_q, d)
        x = x + d.unsqueeze(-1)
        st = [] if stages is not None else None
        wav = self.wave_generator(x, st)
        if stages is not None:
            stages.update(z_q=z_q, d_vector=d, prenet_plus_d=x, wavegen=st)
        return wav

    def detokenize_numpy(self, global_tokens, semantic) -> np.ndarray:
        """``BiCodecTokenizer.detokenize`` (models/audio_tokenizer.py:132-146): (B, Ntok), (B, T)
        -> squeezed float32 numpy."""
        g = torch.as_tensor(np.asarray(global_tokens)).unsqueeze(1)
        s = torch.as_tensor(np.asarray(semantic))
        return self.detokenize(s, g).squeeze().cpu().numpy()
