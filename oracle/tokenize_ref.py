"""ORACLE (test infrastructure, never the product path): CPU restatement of ``BiCodec.tokenize``
(``sparktts/models/bicodec.py:151-169``) in plain fp32 torch-CPU ops -- wav2vec2 features +
reference clip -> semantic token ids (B, T) and global token ids (B, 1, Ntok).

Pinning: ``tests/golden/gen_golden_tok.py`` ran the reference's own ``Encoder``,
``FactorizedVectorQuantize`` and ``SpeakerEncoder`` sub-modules (ECAPA-TDNN, PerceiverResampler,
ResidualFSQ), imported from ``/root/reference`` in the build container, on seeded inputs and
committed inputs/outputs under ``tests/golden/tok_*.npz``; ``tests/test_oracle_tokenize.py`` checks
this file against them.  NOT pinned: the mel spectrogram -- the reference calls
``torchaudio.transforms.MelSpectrogram`` (``bicodec.py:198-211``) and torchaudio is absent from this
image, so ``mel_spectrogram`` below restates torchaudio's published algorithm (``torch.stft`` +
slaney mel filterbank) with **parity unpinned** for that one function; everything downstream of the
mel is pinned.  (The test file also cross-checks it against ``transformers.audio_utils``, an
independent third-party implementation of the same algorithm: a second opinion, not a pin.)

State dict keys are the reference module tree's (weight-norm folded).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .bicodec_ref import _t


# --------------------------------------------------------------------------- host-side audio prep
def audio_volume_normalize(audio: np.ndarray, coeff: float = 0.2) -> np.ndarray:
    """sparktts/utils/audio.py:34-75."""
    temp = np.sort(np.abs(audio))
    if temp[-1] < 0.1:
        audio = audio / max(temp[-1], 1e-3) * 0.1
    temp = temp[temp > 0.01]
    L = temp.shape[0]
    if L <= 10:
        return audio
    volume = np.mean(temp[int(0.9 * L): int(0.99 * L)])
    audio = audio * np.clip(coeff / volume, a_min=0.1, a_max=10)
    max_value = np.max(np.abs(audio))
    if max_value > 1:
        audio = audio / max_value
    return audio


def get_ref_clip(wav: np.ndarray, sample_rate: int, ref_segment_duration: float, latent_hop_length: int) -> np.ndarray:
    """models/audio_tokenizer.py:57-72."""
    n = int(sample_rate * ref_segment_duration) // latent_hop_length * latent_hop_length
    if n > len(wav):
        wav = np.tile(wav, n // len(wav) + 1)
    return wav[:n]


# --------------------------------------------------------------------------- mel (parity unpinned)
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> np.ndarray:
    """torchaudio.functional.melscale_fbanks(norm="slaney", mel_scale="slaney"): (n_freqs, n_mels) fp32."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_pts = np.linspace(_hz_to_mel_slaney(f_min), _hz_to_mel_slaney(f_max), n_mels + 2)
    f_pts = _mel_to_hz_slaney(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (f_pts[2: n_mels + 2] - f_pts[:n_mels])
    return (fb * enorm[None, :]).astype(np.float32)


def mel_spectrogram(wav: torch.Tensor, cfg) -> torch.Tensor:
    """TT.MelSpectrogram(sample_rate, n_fft, win_length, hop_length, f_min, f_max, n_mels, power=1,
    norm="slaney", mel_scale="slaney") as constructed at bicodec.py:200-211: periodic Hann window
    (zero-padded to n_fft by stft), center=True / reflect padding, magnitude, filterbank matmul.
    wav (B, L) -> (B, n_mels, L // hop + 1)."""
    win = torch.hann_window(cfg.win_length, periodic=True, dtype=torch.float32)
    spec = torch.stft(wav, cfg.n_fft, cfg.hop_length, cfg.win_length, win, center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True).abs()
    fmax = cfg.mel_fmax if cfg.mel_fmax is not None else cfg.sample_rate / 2
    fb = torch.from_numpy(melscale_fbanks(cfg.n_fft // 2 + 1, cfg.mel_fmin, fmax, cfg.num_mels, cfg.sample_rate))
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)


# --------------------------------------------------------------------------- model
class BiCodecTokRef:
    def __init__(self, tcfg, folded_state: Dict[str, np.ndarray]):
        self.cfg = tcfg
        self.sd = folded_state

    # ---- encoder (encoder_decoder/feat_encoder.py:76-87; Vocos pieces blocks/vocos.py)
    def _ln(self, x_btc, prefix):
        return F.layer_norm(x_btc, (x_btc.shape[-1],), _t(self.sd, prefix + ".weight"), _t(self.sd, prefix + ".bias"), eps=1e-6)

    def _convnext(self, x, p):
        sd = self.sd
        r = x
        x = F.conv1d(x, _t(sd, p + ".dwconv.weight"), _t(sd, p + ".dwconv.bias"), padding=3, groups=x.shape[1])
        x = self._ln(x.transpose(1, 2), p + ".norm")
        x = F.gelu(F.linear(x, _t(sd, p + ".pwconv1.weight"), _t(sd, p + ".pwconv1.bias")))
        x = F.linear(x, _t(sd, p + ".pwconv2.weight"), _t(sd, p + ".pwconv2.bias"))
        return r + (_t(sd, p + ".gamma") * x).transpose(1, 2)

    def _vocos(self, x, p, nlayers):
        sd = self.sd
        x = F.conv1d(x, _t(sd, p + ".embed.weight"), _t(sd, p + ".embed.bias"), padding=3)
        x = self._ln(x.transpose(1, 2), p + ".norm").transpose(1, 2)
        for j in range(nlayers):
            x = self._convnext(x, f"{p}.convnext.{j}")
        return self._ln(x.transpose(1, 2), p + ".final_layer_norm")       # (B, T, C)

    def encoder(self, feat_bct: torch.Tensor) -> torch.Tensor:
        """(B, C_in, T) -> z (B, out_channels, T).  SamplingBlock with ratio 1 = 3x (samper.py:79-100)."""
        x = self._vocos(feat_bct, "encoder.encoder", self.cfg.enc_num_layers)
        for i in range(len(self.cfg.enc_sample_ratios)):
            x = x.transpose(1, 2)
            x = x + x + x
            x = self._vocos(x, f"encoder.downsample.{i}.1", 2)
        x = F.linear(x, _t(self.sd, "encoder.project.weight"), _t(self.sd, "encoder.project.bias"))
        return x.transpose(1, 2)

    def vq_tokenize(self, z: torch.Tensor, margins: Optional[list] = None) -> torch.Tensor:
        """vq/factorized_vector_quantize.py:148-152,169-187: in_project, L2-normalise both sides,
        arg-max of -(|e|^2 - 2 e.c + |c|^2)."""
        z_e = F.conv1d(z, _t(self.sd, "quantizer.in_project.weight"), _t(self.sd, "quantizer.in_project.bias"))
        B, D, T = z_e.shape
        enc = F.normalize(z_e.transpose(1, 2).reshape(B * T, D))
        cb = F.normalize(_t(self.sd, "quantizer.codebook.weight"))
        dist = enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cb.t() + cb.pow(2).sum(1, keepdim=True).t()
        if margins is not None:
            top2 = (-dist).topk(2, dim=1).values
            margins.append((top2[:, 0] - top2[:, 1]).reshape(B, T))
        return (-dist).max(1)[1].reshape(B, T)

    # ---- speaker encoder analysis side
    def _bn(self, x, p):
        sd = self.sd
        return F.batch_norm(x, _t(sd, p + ".running_mean"), _t(sd, p + ".running_var"), _t(sd, p + ".weight"), _t(sd, p + ".bias"),
                            False, 0.0, 1e-5)

    def _conv_relu_bn(self, x, p, **kw):
        """ecapa_tdnn.py:72-93: bn(relu(conv(x)))."""
        return self._bn(F.relu(F.conv1d(x, _t(self.sd, p + ".conv.weight"), _t(self.sd, p + ".conv.bias"), **kw)), p + ".bn")

    def _se_res2block(self, x, p, dil):
        """ecapa_tdnn.py:14-69,96-134 (scale 8, kernel 3, padding = dilation)."""
        sd = self.sd
        y = self._conv_relu_bn(x, p + ".0")
        w = y.shape[1] // 8
        spx = torch.split(y, w, 1)
        out, sp = [], spx[0]
        for i in range(7):
            if i >= 1:
                sp = sp + spx[i]
            sp = F.conv1d(sp, _t(sd, f"{p}.1.convs.{i}.weight"), _t(sd, f"{p}.1.convs.{i}.bias"), padding=dil, dilation=dil)
            sp = self._bn(F.relu(sp), f"{p}.1.bns.{i}")
            out.append(sp)
        out.append(spx[7])
        y = self._conv_relu_bn(torch.cat(out, 1), p + ".2")
        s = y.mean(dim=2)
        s = F.relu(F.linear(s, _t(sd, p + ".3.linear1.weight"), _t(sd, p + ".3.linear1.bias")))
        s = torch.sigmoid(F.linear(s, _t(sd, p + ".3.linear2.weight"), _t(sd, p + ".3.linear2.bias")))
        return x + y * s.unsqueeze(2)

    def ecapa_latent(self, mel_bft: torch.Tensor) -> torch.Tensor:
        """ecapa_tdnn.py:186-197 up to ``latent`` (the x-vector branch is not used by tokenize)."""
        se = "speaker_encoder.speaker_encoder"
        o1 = self._conv_relu_bn(mel_bft, se + ".layer1", padding=2)
        o2 = self._se_res2block(o1, se + ".layer2.se_res2block", 2)
        o3 = self._se_res2block(o2, se + ".layer3.se_res2block", 3)
        o4 = self._se_res2block(o3, se + ".layer4.se_res2block", 4)
        return F.relu(F.conv1d(torch.cat([o2, o3, o4], 1), _t(self.sd, se + ".conv.weight"), _t(self.sd, se + ".conv.bias")))

    def perceiver(self, feats_btc: torch.Tensor) -> torch.Tensor:
        """perceiver_encoder.py:297-350 (Attention :246-294 with cross_attn_include_queries, Attend
        :125-160, FeedForward/GEGLU :213-234, RMSNorm :180-198).  (B, T, 1536) -> (B, Ntok, latent)."""
        sd, c = self.sd, self.cfg
        ps = "speaker_encoder.perceiver_sampler"
        x = F.linear(feats_btc, _t(sd, ps + ".proj_context.weight"), _t(sd, ps + ".proj_context.bias"))
        B = x.shape[0]
        lat = _t(sd, ps + ".latents")[None].expand(B, -1, -1)
        h, dh = c.perceiver_heads, c.perceiver_dim_head
        for i in range(c.perceiver_depth):
            a = f"{ps}.layers.{i}.0"
            ctx = torch.cat((lat, x), dim=-2)
            q = F.linear(lat, _t(sd, a + ".to_q.weight"))
            k, v = F.linear(ctx, _t(sd, a + ".to_kv.weight")).chunk(2, dim=-1)
            sh = lambda t: t.reshape(B, t.shape[1], h, dh).transpose(1, 2)   # noqa: E731
            q, k, v = sh(q), sh(k), sh(v)
            sim = torch.einsum("bhid,bhjd->bhij", q, k) * (dh ** -0.5)
            o = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
            o = o.transpose(1, 2).reshape(B, -1, h * dh)
            lat = F.linear(o, _t(sd, a + ".to_out.weight")) + lat
            f = f"{ps}.layers.{i}.1"
            y = F.linear(lat, _t(sd, f + ".0.weight"), _t(sd, f + ".0.bias"))
            val, gate = y.chunk(2, dim=-1)
            lat = F.linear(F.gelu(gate) * val, _t(sd, f + ".2.weight"), _t(sd, f + ".2.bias")) + lat
        return F.normalize(lat, dim=-1) * (lat.shape[-1] ** 0.5) * _t(sd, ps + ".norm.gamma")

    def fsq_indices(self, x_bnd: torch.Tensor, bounded: Optional[list] = None) -> torch.Tensor:
        """fsq/residual_fsq.py:211-276 with one quantizer (scale 1) + finite_scalar_quantization.py:
        bound (:113-118), round, codes_to_indices (:133-137).  (B, Ntok, latent) -> (B, Ntok) int32."""
        sd = self.sd
        z = F.linear(x_bnd, _t(sd, "speaker_encoder.quantizer.project_in.weight"), _t(sd, "speaker_encoder.quantizer.project_in.bias"))
        lv = torch.tensor(self.cfg.fsq_levels, dtype=torch.int32)
        half_l = (lv - 1) * (1 + 1e-3) / 2
        offset = torch.where(lv % 2 == 0, 0.5, 0.0)
        shift = (offset / half_l).atanh()
        b = (z + shift).tanh() * half_l - offset
        if bounded is not None:
            bounded.append(b)
        q = b.round()
        half_w = lv // 2
        zhat = q / half_w
        basis = torch.cumprod(torch.tensor([1] + list(self.cfg.fsq_levels[:-1])), dim=0).to(torch.int32)
        return ((zhat * half_w + half_w) * basis).sum(dim=-1).to(torch.int32)

    # ---- top level
    @torch.no_grad()
    def tokenize_from_mel(self, feat_btc: torch.Tensor, mel_bft: torch.Tensor, stages: Optional[dict] = None):
        """bicodec.py:162-169 after the mel transform.  feat (B, T, C) wav2vec2 mix; mel (B, n_mels, Tm).
        Returns (semantic (B, T) int64, global (B, 1, Ntok) int32)."""
        z = self.encoder(feat_btc.transpose(1, 2))
        mg = [] if stages is not None else None
        sem = self.vq_tokenize(z, mg)
        latent = self.ecapa_latent(mel_bft)
        x = self.perceiver(latent.transpose(1, 2))
        bd = [] if stages is not None else None
        idx = self.fsq_indices(x, bd)
        if stages is not None:
            stages.update(z=z, vq_margin=mg[0], ecapa_latent=latent, perceiver=x, fsq_bounded=bd[0])
        return sem, idx.unsqueeze(1)

    @torch.no_grad()
    def tokenize(self, feat_btc: torch.Tensor, ref_wav: torch.Tensor, stages: Optional[dict] = None):
        mel = mel_spectrogram(ref_wav, self.cfg)
        if stages is not None:
            stages["mel"] = mel
        return self.tokenize_from_mel(feat_btc, mel, stages)
