"""Host side of the prompt encoder (sparkmi/encoder.py, no GPU): audio preparation against the oracle's
restatement of sparktts/utils/audio.py, WAV reading, the derived mel tensors, arena layout."""
import ctypes as C
import wave

import numpy as np
import pytest
import torch

from oracle import tokenize_ref as O
from sparkmi import _lib, config as Cf, config_tok as T, encoder as E, weights as W


def test_volume_normalize_and_ref_clip_equal_the_oracle():
    rng = np.random.default_rng(0)
    for scale in (0.003, 0.05, 0.4, 3.0):
        a = rng.standard_normal(30000) * scale
        np.testing.assert_array_equal(E.audio_volume_normalize(a.copy()), O.audio_volume_normalize(a.copy()))
    few = np.zeros(1000); few[:5] = 0.5
    np.testing.assert_array_equal(E.audio_volume_normalize(few.copy()), O.audio_volume_normalize(few.copy()))
    for n in (500, 95999, 96000, 200000):
        x = rng.standard_normal(n)
        np.testing.assert_array_equal(E.get_ref_clip(x, 16000, 6, 320), O.get_ref_clip(x, 16000, 6, 320))


def test_wav_reading_and_load_audio(tmp_path):
    x = np.sin(np.arange(16000) * 0.05) * 0.5
    p = tmp_path / "a.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
        st = np.stack([x, -x], 1)
        w.writeframes((st * 32767).astype("<i2").tobytes())
    a, sr = E.read_audio(p)
    assert sr == 16000 and a.shape == (16000, 2)
    got = E.load_audio(p, sampling_rate=16000, volume_normalize=False)
    assert got.shape == (16000,) and np.abs(got - (x * 32767).astype(np.int16) / 32768.0).max() < 1e-9   # first channel
    n = E.load_audio(p, sampling_rate=16000, volume_normalize=True)
    np.testing.assert_array_equal(n, O.audio_volume_normalize(got))
    p8 = tmp_path / "b.wav"
    with wave.open(str(p8), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
        w.writeframes((x[::2] * 32767).astype("<i2").tobytes())
    r = E.load_audio(p8, sampling_rate=16000)
    assert abs(len(r) - 16000) <= 2            # resampled to the model rate (host-side; soxr in the reference)
    with pytest.raises(FileNotFoundError):
        E.load_audio(tmp_path / "missing.wav")


@pytest.mark.parametrize("tcfg", [T.tiny_tok(), T.spark_0p5b_tok()])
def test_derived_mel_tensors(tcfg):
    fb = E.mel_filterbank(tcfg)
    fmax = tcfg.mel_fmax if tcfg.mel_fmax is not None else tcfg.sample_rate / 2
    np.testing.assert_array_equal(fb, O.melscale_fbanks(tcfg.n_fft // 2 + 1, tcfg.mel_fmin, fmax, tcfg.num_mels, tcfg.sample_rate).T)
    basis = E.dft_basis(tcfg)
    nf = tcfg.n_fft // 2 + 1
    assert basis.shape == (2 * nf, tcfg.n_fft)
    # basis @ frame == rfft(window * frame): the DFT-as-GEMM the GPU runs equals torch.stft's per-frame transform
    rng = np.random.default_rng(1)
    frame = rng.standard_normal(tcfg.n_fft)
    win = torch.hann_window(tcfg.win_length, periodic=True, dtype=torch.float64).numpy()
    wpad = np.zeros(tcfg.n_fft); lo = (tcfg.n_fft - tcfg.win_length) // 2; wpad[lo:lo + tcfg.win_length] = win
    want = np.fft.rfft(frame * wpad)
    got = basis.astype(np.float64) @ frame
    np.testing.assert_allclose(got[:nf], want.real, atol=2e-4)
    np.testing.assert_allclose(got[nf:], want.imag, atol=2e-4)
    # and the whole mel of the oracle = filterbank @ |basis @ frames|
    wav = torch.from_numpy(rng.standard_normal(tcfg.hop_length * 12).astype(np.float32))[None]
    mel = O.mel_spectrogram(wav, tcfg)[0].numpy()
    x = np.pad(wav[0].numpy().astype(np.float64), tcfg.n_fft // 2, mode="reflect")
    frames = np.stack([x[i * tcfg.hop_length: i * tcfg.hop_length + tcfg.n_fft] for i in range(mel.shape[1])], 1)
    d = basis.astype(np.float64) @ frames
    np.testing.assert_allclose(fb.astype(np.float64) @ np.sqrt(d[:nf] ** 2 + d[nf:] ** 2), mel, rtol=2e-3, atol=2e-4)


def test_arena_layout_and_packing_tiny():
    wcfg, tcfg, vcfg = T.tiny_wav2vec2(), T.tiny_tok(), Cf.tiny_bicodec()
    cs = E.enc_cfg_struct(wcfg, tcfg, 48000, 16000 + tcfg.n_fft)
    lib = _lib.lib()
    n = lib.smi_enc_arena_count(C.byref(cs))
    assert n > 100
    names, end = [], 0
    name = C.create_string_buffer(8192)
    for i in range(n):
        off, nb = C.c_size_t(), C.c_size_t()
        info = (C.c_int32 * 6)()
        assert lib.smi_enc_arena_entry(C.byref(cs), i, name, 8192, C.byref(off), C.byref(nb), info) == 0
        assert off.value >= end and off.value % 256 == 0
        end = off.value + nb.value
        names.append(name.value.decode())
    assert end <= lib.smi_enc_arena_bytes(C.byref(cs))
    assert len(set(names)) == n and "mel.dft" in names and any(k.startswith("bnscale:") for k in names)
    arena = E.pack_enc_arena(tcfg, W.fold_pos_conv_weight_norm(W.wav2vec2_state(wcfg)),
                             W.fold_weight_norm(W.bicodec_tok_state(tcfg, vcfg.vq_input_dim)), cs)
    assert arena.dtype == np.float32 and np.isfinite(arena).all() and arena.size * 4 == lib.smi_enc_arena_bytes(C.byref(cs))
    bad = E.enc_cfg_struct(wcfg, tcfg, 48000, 16000 + tcfg.n_fft)
    bad.w2v_heads = 3
    assert lib.smi_enc_arena_count(C.byref(bad)) < 0           # hidden != heads * 64: outside the kernel contract
    with pytest.raises(ValueError):
        T.Wav2Vec2Cfg(feat_extract_norm="group").validate()
    with pytest.raises(_lib.SparkMIError):
        E.BiCodecEncoder(wcfg, tcfg, None, None, "cpu")        # no CPU path
