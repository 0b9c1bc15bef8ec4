"""GPU parity of the prompt encoder (smi_enc_*, through the C ABI) against the CPU oracles and the
golden vectors from transformers' Wav2Vec2Model + the reference's BiCodec sub-modules.

Tolerances: every contraction is exact-fp32 MFMA / fp32 VALU with fp32 accumulation, so activations
agree with the fp32 oracle to summation-order noise (observed ~1e-5 relative after 16 transformer
layers); bounds below are absolute on activations of magnitude O(1-10).  Token ids must be EQUAL
wherever the decision margin exceeds that noise (checked frame by frame with the oracle's margins)."""
import os

import numpy as np
import pytest
import torch

from oracle.tokenize_ref import BiCodecTokRef, get_ref_clip, mel_spectrogram
from oracle.wav2vec2_ref import Wav2Vec2Ref
from sparkmi import config as C, config_tok as T, weights as W

pytestmark = pytest.mark.gpu


def _build(wcfg, tcfg, vcfg, **kw):
    from sparkmi.encoder import BiCodecEncoder
    wsd = W.wav2vec2_state(wcfg)
    tsd = W.fold_weight_norm(W.bicodec_tok_state(tcfg, vcfg.vq_input_dim))
    kw.setdefault("diag", any(k.startswith("SPARKMI_") for k in os.environ))   # SPARKMI_* switches exist in the diagnostics build only
    enc = BiCodecEncoder(wcfg, tcfg, W.fold_pos_conv_weight_norm(wsd), tsd, "cuda:0", **kw)
    return enc, wsd, tsd


@pytest.fixture(scope="module")
def tiny():
    wcfg, tcfg, vcfg = T.tiny_wav2vec2(), T.tiny_tok(), C.tiny_bicodec()
    enc, wsd, tsd = _build(wcfg, tcfg, vcfg, max_seconds=3.0, ref_seconds=1.0)
    return enc, wcfg, tcfg, Wav2Vec2Ref(wcfg, wsd), BiCodecTokRef(tcfg, tsd)


def test_tiny_stages_and_tokens_match_oracle_and_golden(tiny, golden_dir):
    enc, wcfg, tcfg, w2v, tok = tiny
    g = np.load(os.path.join(golden_dir, "tok_tiny.npz"))
    wav = g["wav"]
    n_ref = int(16000 * 0.5) // tcfg.hop_length * tcfg.hop_length
    ref = np.tile(wav, n_ref // len(wav) + 1)[:n_ref]
    glob, sem = enc.tokenize_arrays(wav, ref)
    st = {}
    feat = w2v.features(wav, st)
    hs = w2v.hidden_states(g["input_values"])
    got = {k: enc.debug_stage(k).cpu().numpy() for k in ("input_values", "conv_feats", "hs0", "feat", "z", "mel",
                                                         "ecapa_latent", "perceiver", "fsq_bounded")}
    assert np.abs(got["input_values"][0] - g["input_values"]).max() < 2e-6
    assert np.abs(got["conv_feats"] - st["conv_feats"][0].numpy()).max() < 2e-5
    assert np.abs(got["hs0"] - hs[0][0].numpy().T).max() < 5e-5
    assert np.abs(got["hs0"] - g["hs_first"].T).max() < 1e-4
    assert np.abs(got["feat"] - feat[0].numpy().T).max() < 2e-4
    assert np.abs(got["feat"] - g["feat"].T).max() < 2e-4          # transformers' own output
    ost = {}
    mel = mel_spectrogram(torch.from_numpy(ref)[None], tcfg)
    osem, oglob = tok.tokenize_from_mel(feat, mel, ost)
    assert np.abs(got["mel"] - mel[0].numpy()).max() < 2e-4 * max(1.0, float(mel.abs().max()))
    assert np.abs(got["z"] - ost["z"][0].numpy()).max() < 5e-4
    assert np.abs(got["z"] - g["z"]).max() < 5e-4                   # the reference Encoder's output
    assert np.abs(got["ecapa_latent"] - ost["ecapa_latent"][0].numpy()).max() < 5e-4
    assert np.abs(got["perceiver"] - ost["perceiver"][0].numpy().T).max() < 5e-4
    assert np.abs(got["fsq_bounded"] - ost["fsq_bounded"][0].numpy()).max() < 5e-4
    # tokens: equal to the reference modules' wherever the decision is not a numerical coin flip
    sem, glob = sem.cpu().numpy(), glob.cpu().numpy()
    assert sem.shape == g["sem"].shape and glob.shape == g["glob"].shape and glob.dtype == np.int32
    safe = ost["vq_margin"].numpy() > 1e-4
    assert safe.mean() > 0.9
    np.testing.assert_array_equal(sem[safe], g["sem"][safe])
    assert (sem == g["sem"]).mean() > 0.97
    bd = ost["fsq_bounded"][0].numpy()
    safe_g = (np.abs(bd - np.floor(bd) - 0.5) > 1e-3).all(axis=1)
    np.testing.assert_array_equal(glob[0, 0][safe_g], g["glob"][0, 0][safe_g])


def test_tiny_other_lengths_and_determinism(tiny):
    enc, wcfg, tcfg, w2v, tok = tiny
    rng = np.random.default_rng(9)
    for n in (2000, 7777, 16000 * 2 + 123):
        wav = (0.1 * rng.standard_normal(n)).astype(np.float32)
        ref = get_ref_clip(wav, 16000, 1.0, tcfg.hop_length).astype(np.float32)
        glob, sem = enc.tokenize_arrays(wav, ref)
        feat = w2v.features(wav)
        assert sem.shape == (1, wcfg.frames(n))
        assert np.abs(enc.debug_stage("feat").cpu().numpy() - feat[0].numpy().T).max() < 3e-4
        ost = {}
        osem, oglob = tok.tokenize(feat, torch.from_numpy(ref)[None], ost)
        safe = ost["vq_margin"].numpy() > 1e-4
        np.testing.assert_array_equal(sem.cpu().numpy()[safe], osem.numpy()[safe])
        glob2, sem2 = enc.tokenize_arrays(wav, ref)
        assert torch.equal(sem, sem2) and torch.equal(glob, glob2)
    from sparkmi._lib import SparkMIError
    with pytest.raises(SparkMIError):
        enc.tokenize_arrays(np.zeros(100, np.float32), np.zeros(8000, np.float32))     # too short for one frame
    with pytest.raises(ValueError):
        enc.tokenize_arrays(np.zeros(16000 * 4, np.float32), np.zeros(8000, np.float32))  # beyond max_samples


def test_full_size_xlsr53_against_golden(golden_dir):
    """wav2vec2-large-xlsr-53 shape (16 of 24 layers run) + the 0.5B BiCodec tokenizer dims, synthetic
    weights: transformers' feature mix and the reference modules' tokens (tests/golden/tok_full.npz)."""
    wcfg, tcfg, vcfg = T.xlsr53(), T.spark_0p5b_tok(), C.spark_0p5b_bicodec()
    g = np.load(os.path.join(golden_dir, "tok_full.npz"))
    enc, wsd, tsd = _build(wcfg, tcfg, vcfg, max_seconds=6.0, ref_seconds=6.0)
    wav = g["wav"]
    ref = get_ref_clip(wav, 16000, 6.0, tcfg.hop_length).astype(np.float32)
    glob, sem = enc.tokenize_arrays(wav, ref)
    a, b, c = wcfg.taps
    feat = enc.debug_stage("feat").cpu().numpy()                    # [1024][T]
    scale = float(np.abs(g["feat"]).max())
    assert np.abs(feat.T[::2, ::8] - g["feat"]).max() < 2e-4 * max(1.0, scale)
    hs0 = enc.debug_stage("hs0").cpu().numpy()
    assert np.abs(hs0.T[::4, ::16] - g["hs_first"]).max() < 1e-4 * max(1.0, float(np.abs(g["hs_first"]).max()))
    z = enc.debug_stage("z").cpu().numpy()
    assert np.abs(z[::16, ::4] - g["z_sample"]).max() < 1e-3
    assert abs(float(np.abs(z.astype(np.float64)).sum()) - float(g["z_abs_sum"])) < 1e-4 * float(g["z_abs_sum"])
    lat = enc.debug_stage("ecapa_latent").cpu().numpy()
    assert np.abs(lat[::32, ::8] - g["ecapa_latent_sample"]).max() < 1e-3
    assert np.abs(enc.debug_stage("perceiver").cpu().numpy() - g["perceiver"]).max() < 1e-3
    sem, glob = sem.cpu().numpy(), glob.cpu().numpy()
    assert sem.shape == g["sem"].shape
    assert (sem == g["sem"]).mean() > 0.97, f"semantic token agreement {(sem == g['sem']).mean()}"
    assert (glob == g["glob"]).mean() > 0.9, f"global token agreement {(glob == g['glob']).mean()}"


def test_tokenize_many_on_parallel_streams_equals_one_by_one(tiny):
    """Independent prompts encoded side by side on their own HIP streams (own handle, shared arena) give exactly the ids of
    one-by-one calls, whatever the number of lanes and with more prompts than lanes."""
    enc = tiny[0]
    rng = np.random.default_rng(9)
    wavs = [(0.2 * rng.standard_normal(int(16000 * s))).astype(np.float32) for s in (1.0, 2.3, 0.7, 1.9, 1.1)]
    refs = [get_ref_clip(w.astype(np.float64), 16000, 1.0, 320).astype(np.float32) for w in wavs]
    one = [enc.tokenize_arrays(w, r) for w, r in zip(wavs, refs)]
    torch.cuda.synchronize()
    for lanes in (1, 3, 8):
        many = enc.tokenize_many(wavs, refs, lanes=lanes)
        torch.cuda.synchronize()
        for (g1, s1), (g2, s2) in zip(one, many):
            assert torch.equal(g1, g2) and torch.equal(s1, s2)


def test_graph_replay_equals_eager_launches(tiny, monkeypatch):
    """A (n_samples, n_ref) shape seen twice is captured and replayed as one hipGraph (prompt copied into / ids copied out of
    handle-owned buffers): other prompts of that shape, and the debug views, must equal the eager launches'."""
    enc, wcfg, tcfg, _, _ = tiny
    rng = np.random.default_rng(21)
    n = 16000 + 321
    wavs = [(0.15 * rng.standard_normal(n)).astype(np.float32) for _ in range(4)]
    refs = [get_ref_clip(w.astype(np.float64), 16000, 1.0, tcfg.hop_length).astype(np.float32) for w in wavs]
    got = [enc.tokenize_arrays(w, r) for w, r in zip(wavs, refs)]             # eager, capture + replay, replay, replay
    feat_last = enc.debug_stage("feat").cpu().numpy()
    short = enc.tokenize_arrays(wavs[0][:9000], refs[0])                      # another shape in between
    again = enc.tokenize_arrays(wavs[1], refs[1])
    monkeypatch.setenv("SPARKMI_ENC_GRAPH", "0")
    eager, _, _ = _build(wcfg, tcfg, C.tiny_bicodec(), max_seconds=3.0, ref_seconds=1.0)
    for (g, s_), w, r in zip(got, wavs, refs):
        ge, se = eager.tokenize_arrays(w, r)
        assert torch.equal(g, ge) and torch.equal(s_, se)
    assert np.array_equal(eager.debug_stage("feat").cpu().numpy(), feat_last)
    ge, se = eager.tokenize_arrays(wavs[0][:9000], refs[0])
    assert torch.equal(short[0], ge) and torch.equal(short[1], se)
    assert torch.equal(again[0], got[1][0]) and torch.equal(again[1], got[1][1])


@pytest.mark.timeout(900)
def test_config5_eight_prompts_at_full_size_streams_and_graph_replay(golden_dir):
    """BASELINE configs[4]'s prompt-encode leg as the bench runs it (audio_tokenizer.py:85-130, bicodec.py:151-169): eight
    six-second prompts at wav2vec2-large-xlsr-53 size through `tokenize_many` -- eight HIP streams, per-lane handles -- twice
    (the second pass replays the hipGraph captured per shape).  The ids of both passes equal eight one-by-one eager calls bit
    for bit; prompt 0 is the golden prompt and reproduces tests/golden/tok_full.npz (the reference modules' own tokens)."""
    wcfg, tcfg, vcfg = T.xlsr53(), T.spark_0p5b_tok(), C.spark_0p5b_bicodec()
    g = np.load(os.path.join(golden_dir, "tok_full.npz"))
    enc, wsd, tsd = _build(wcfg, tcfg, vcfg, max_seconds=6.0, ref_seconds=6.0)
    base = g["wav"].astype(np.float32)
    wavs = [np.roll(base, int(8000 * i)) for i in range(8)]               # SURVEY 8d cfg 5: the prompt shifted by i * 0.5 s
    refs = [get_ref_clip(w.astype(np.float64), 16000, 6.0, tcfg.hop_length).astype(np.float32) for w in wavs]
    first = enc.tokenize_many(wavs, refs, lanes=8)                        # eager on every lane (shape seen once per lane)
    torch.cuda.synchronize()
    second = enc.tokenize_many(wavs, refs, lanes=8)                       # capture + replay
    third = enc.tokenize_many(wavs, refs, lanes=8)                        # replay
    torch.cuda.synchronize()
    one = [enc.tokenize_arrays(w, r) for w, r in zip(wavs, refs)]
    for i in range(8):
        for many in (first, second, third):
            assert torch.equal(many[i][0], one[i][0]) and torch.equal(many[i][1], one[i][1]), f"prompt {i}"
    sem0, glob0 = one[0][1].cpu().numpy(), one[0][0].cpu().numpy()
    assert sem0.shape == g["sem"].shape
    assert (sem0 == g["sem"]).mean() > 0.97 and (glob0 == g["glob"]).mean() > 0.9
    # the shifted prompts are different inputs: their ids must not all collapse onto prompt 0's
    assert any(not torch.equal(one[i][1], one[0][1]) for i in range(1, 8))
