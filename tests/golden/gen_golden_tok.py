#!/usr/bin/env python3
"""Golden vectors of the voice-clone prompt encode, build container only.

* wav2vec2 half: ``transformers.Wav2Vec2Model`` (third-party arithmetic the reference calls at
  ``sparktts/models/audio_tokenizer.py:53-55,87-98``) + ``Wav2Vec2FeatureExtractor``, random-init
  architecture loaded with the build's synthetic weights -> tapped hidden states and their mix.
* BiCodec half: the reference's own ``Encoder``, ``FactorizedVectorQuantize`` and ``SpeakerEncoder``
  sub-modules (``sparktts.modules.*`` imported from /root/reference), assembled as
  ``BiCodec.tokenize`` does (``bicodec.py:162-169``).  ``bicodec.py`` itself and the mel transform
  need torchaudio/omegaconf (absent: ordinary ModuleNotFoundError), so the speaker branch is fed
  the oracle's mel; the mel function itself stays unpinned (oracle/tokenize_ref.py header).

    python tests/golden/gen_golden_tok.py [--full]
"""
from __future__ import annotations

import argparse
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

from sparkmi import config as C, config_tok as T, weights as W  # noqa: E402
from oracle.tokenize_ref import mel_spectrogram  # noqa: E402


def synth_wav(n, seed):
    """Deterministic speech-like test signal: a few drifting harmonics + noise bursts."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n) / 16000.0
    f0 = 110 + 40 * np.sin(2 * np.pi * 0.7 * t)
    ph = 2 * np.pi * np.cumsum(f0) / 16000.0
    x = sum(a * np.sin(k * ph) for k, a in ((1, 0.5), (2, 0.3), (3, 0.2), (5, 0.1)))
    env = 0.5 + 0.5 * np.sin(2 * np.pi * 2.3 * t) ** 2
    return (0.3 * env * x + 0.02 * rng.standard_normal(n)).astype(np.float32)


def hf_wav2vec2(cfg, sd):
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    hc = Wav2Vec2Config(conv_dim=cfg.conv_dim, conv_kernel=cfg.conv_kernel, conv_stride=cfg.conv_stride, conv_bias=cfg.conv_bias,
                        hidden_size=cfg.hidden_size, num_hidden_layers=cfg.used_layers + 1,   # one more than the last tap: hidden_states[-1] alone is layer-normed
                        num_attention_heads=cfg.num_attention_heads,
                        intermediate_size=cfg.intermediate_size, num_conv_pos_embeddings=cfg.num_conv_pos_embeddings,
                        num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups, feat_extract_norm="layer",
                        do_stable_layer_norm=True, layer_norm_eps=cfg.layer_norm_eps, hidden_dropout=0.0, attention_dropout=0.0,
                        activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0)
    m = Wav2Vec2Model(hc).eval()
    res = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    miss = [k for k in res.missing_keys if not (k.startswith(("encoder.layer_norm", f"encoder.layers.{cfg.used_layers}.")) or k == "masked_spec_embed")]
    assert not miss, miss
    return m


def reference_tok_modules(tcfg, vcfg, sd_unfolded):
    from sparktts.modules.encoder_decoder.feat_encoder import Encoder
    from sparktts.modules.vq.factorized_vector_quantize import FactorizedVectorQuantize
    from sparktts.modules.speaker.ecapa_tdnn import ECAPA_TDNN
    from sparktts.modules.speaker.perceiver_encoder import PerceiverResampler
    from sparktts.modules.fsq.residual_fsq import ResidualFSQ
    y = tcfg.to_yaml_dict()

    class Spk(torch.nn.Module):   # the sub-modules SpeakerEncoder.__init__ builds (speaker_encoder.py:55-69), at this config's dims
        def __init__(self):
            super().__init__()
            self.speaker_encoder = ECAPA_TDNN(channels=tcfg.ecapa_channels, feat_dim=tcfg.num_mels, embed_dim=vcfg.spk_out_dim,
                                              pooling_func="ASTP", global_context_att=True)
            self.perceiver_sampler = PerceiverResampler(dim=tcfg.spk_latent_dim, dim_context=512 * 3, num_latents=tcfg.spk_token_num,
                                                        heads=tcfg.perceiver_heads, dim_head=tcfg.perceiver_dim_head,
                                                        depth=tcfg.perceiver_depth, ff_mult=tcfg.perceiver_ff_mult)
            self.quantizer = ResidualFSQ(levels=list(tcfg.fsq_levels), num_quantizers=1, dim=tcfg.spk_latent_dim,
                                         is_channel_first=True, quantize_dropout=False)

        def tokenize(self, mels):   # speaker_encoder.py:100-105
            _, features = self.speaker_encoder(mels, True)
            x = self.perceiver_sampler(features.transpose(1, 2)).transpose(1, 2)
            zq, indices = self.quantizer(x)
            return indices, features, x

    mods = torch.nn.ModuleDict(dict(
        encoder=Encoder(**y["encoder"]),
        quantizer=FactorizedVectorQuantize(input_dim=vcfg.vq_input_dim, codebook_size=tcfg.codebook_size,
                                           codebook_dim=tcfg.codebook_dim, commitment=0.25),
        speaker_encoder=Spk()))
    state = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_unfolded.items()}
    res = mods.load_state_dict(state, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    touched = ("encoder.", "quantizer.in_project", "quantizer.codebook", "speaker_encoder.speaker_encoder.layer",
               "speaker_encoder.speaker_encoder.conv.", "speaker_encoder.perceiver_sampler", "speaker_encoder.quantizer.project_in")
    miss = [k for k in res.missing_keys if k.startswith(touched) and "num_batches_tracked" not in k]
    assert not miss, miss
    mods.eval()

    def _rm(m):
        try:
            torch.nn.utils.remove_weight_norm(m)
        except ValueError:
            pass
    mods.apply(_rm)
    return mods


@torch.no_grad()
def run(tag, wcfg, tcfg, vcfg, seconds, ref_seconds, full_stages):
    wav = synth_wav(int(16000 * seconds), 77)
    from transformers import Wav2Vec2FeatureExtractor
    fe = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True, return_attention_mask=True)
    iv = fe(wav, sampling_rate=16000, return_tensors="pt", padding=True).input_values
    wsd = W.wav2vec2_state(wcfg)
    m = hf_wav2vec2(wcfg, wsd)
    out = m(iv, output_hidden_states=True)
    a, b, c = wcfg.taps
    hs = out.hidden_states
    feat = (hs[a] + hs[b] + hs[c]) / 3                     # audio_tokenizer.py:96-98
    tsd = W.bicodec_tok_state(tcfg, vcfg.vq_input_dim)
    mods = reference_tok_modules(tcfg, vcfg, tsd)
    z = mods["encoder"](feat.transpose(1, 2))              # bicodec.py:165
    sem = mods["quantizer"].tokenize(z)                    # bicodec.py:166
    n_ref = int(16000 * ref_seconds) // tcfg.hop_length * tcfg.hop_length
    ref_wav = torch.from_numpy(np.tile(wav, n_ref // len(wav) + 1)[:n_ref])[None]
    mel = mel_spectrogram(ref_wav, tcfg)                   # oracle mel (torchaudio absent)
    glob, latent, perc = mods["speaker_encoder"].tokenize(mel.transpose(1, 2))   # bicodec.py:167
    d = dict(wav=wav, input_values=iv[0].numpy(), feat=feat[0].numpy(), sem=sem.numpy(), mel=mel[0].numpy(),
             glob=glob.numpy().astype(np.int32), perceiver=perc[0].numpy(),
             hs_first=hs[0][0].numpy(), z_abs_sum=np.float64(z.double().abs().sum()), latent_abs_sum=np.float64(latent.double().abs().sum()))
    if full_stages:
        d.update(z=z[0].numpy(), ecapa_latent=latent[0].numpy(), **{f"hs{i}": hs[i][0].numpy() for i in (a, b, c)})
    else:   # keep the committed file small: strided samples of the big tensors
        d.update(z_sample=z[0, ::16, ::4].numpy(), ecapa_latent_sample=latent[0, ::32, ::8].numpy(),
                 **{f"hs{i}_sample": hs[i][0, ::4, ::16].numpy() for i in (a, b, c)})
        d["feat"] = feat[0, ::2, ::8].numpy()
        d["hs_first"] = hs[0][0, ::4, ::16].numpy()
        del d["input_values"]
    path = os.path.join(HERE, f"tok_{tag}.npz")
    np.savez_compressed(path, **d)
    print(tag, "frames", feat.shape[1], "sem[:8]", sem[0, :8].tolist(), "glob[:8]", glob.reshape(-1)[:8].tolist(),
          "distinct sem", len(set(sem[0].tolist())), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    a = ap.parse_args()
    run("tiny", T.tiny_wav2vec2(), T.tiny_tok(), C.tiny_bicodec(), 1.3, 0.5, True)
    if a.full:
        run("full", T.xlsr53(), T.spark_0p5b_tok(), C.spark_0p5b_bicodec(), 3.0, 6.0, False)
