"""Prompt encoder: ``BiCodecTokenizer.tokenize`` (``sparktts/models/audio_tokenizer.py:85-130``)
on the HIP kernels of ``smi_enc.hip`` -- wav2vec2 features, BiCodec encoder + cosine VQ (semantic
ids) and mel -> ECAPA-TDNN -> perceiver -> FSQ (global ids).

Host side of the row (cheap, numpy): reading the file, channel selection, volume normalisation and
the reference clip (``sparktts/utils/audio.py:34-110``, ``audio_tokenizer.py:57-83``).  Resampling
uses ``scipy.signal.resample_poly`` where the reference uses soxr VHQ (absent offline): prompts that
are already at the model's 16 kHz -- the shipped examples -- take exactly the reference's path.
"""
from __future__ import annotations

import ctypes as C
import math
from pathlib import Path
from typing import Sequence, Dict, Mapping, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib
from .bicodec import PACK_CONV_B, PACK_CONVT_B, PACK_RAW, pack_conv, pack_conv_b
from .config import BiCodecConfig
from .config_tok import TokCfg, Wav2Vec2Cfg


# --------------------------------------------------------------------------- host audio prep
def audio_volume_normalize(audio: np.ndarray, coeff: float = 0.2) -> np.ndarray:
    """sparktts/utils/audio.py:34-75 (same arithmetic, numpy)."""
    temp = np.sort(np.abs(audio))
    if temp[-1] < 0.1:
        audio = audio / max(temp[-1], 1e-3) * 0.1
    temp = temp[temp > 0.01]
    n = temp.shape[0]
    if n <= 10:
        return audio
    volume = np.mean(temp[int(0.9 * n): int(0.99 * n)])
    audio = audio * np.clip(coeff / volume, a_min=0.1, a_max=10)
    peak = np.max(np.abs(audio))
    if peak > 1:
        audio = audio / peak
    return audio


def read_audio(path) -> Tuple[np.ndarray, int]:
    """(float64 samples in [-1, 1], sample rate): soundfile when present, else PCM / float WAV via ``wave``."""
    try:
        import soundfile
        return soundfile.read(str(path))
    except ImportError:
        pass
    import wave
    with wave.open(str(path), "rb") as w:
        sr, nch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    elif sw == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0
    elif sw == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported sample width {sw}")
    return (a.reshape(-1, nch) if nch > 1 else a), sr


def load_audio(path, sampling_rate: int = None, volume_normalize: bool = False) -> np.ndarray:
    """sparktts/utils/audio.py:78-122 without the training-only segment / length options."""
    audio, sr = read_audio(path)
    if audio.ndim > 1:
        audio = audio[:, 0]
    if sampling_rate is not None and sr != sampling_rate:
        from scipy.signal import resample_poly
        g = math.gcd(int(sr), int(sampling_rate))
        audio = resample_poly(audio, sampling_rate // g, sr // g)
    if volume_normalize:
        audio = audio_volume_normalize(audio)
    return audio


def get_ref_clip(wav: np.ndarray, sample_rate: int, ref_segment_duration: float, latent_hop_length: int) -> np.ndarray:
    """audio_tokenizer.py:57-72: fixed-length reference clip, tiling short prompts."""
    n = int(sample_rate * ref_segment_duration) // latent_hop_length * latent_hop_length
    if n > len(wav):
        wav = np.tile(wav, n // len(wav) + 1)
    return wav[:n]


# --------------------------------------------------------------------------- derived tensors
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(t: TokCfg) -> np.ndarray:
    """[num_mels][n_fft/2+1]: torchaudio melscale_fbanks(norm="slaney", mel_scale="slaney"), transposed."""
    nf = t.n_fft // 2 + 1
    fmax = t.mel_fmax if t.mel_fmax is not None else t.sample_rate / 2
    all_freqs = np.linspace(0, t.sample_rate // 2, nf)
    f_pts = _mel_to_hz(np.linspace(_hz_to_mel(t.mel_fmin), _hz_to_mel(fmax), t.num_mels + 2))
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))
    fb = fb * (2.0 / (f_pts[2: t.num_mels + 2] - f_pts[: t.num_mels]))[None, :]
    return np.ascontiguousarray(fb.astype(np.float32).T)


def dft_basis(t: TokCfg) -> np.ndarray:
    """[2*(n_fft/2+1)][n_fft]: rows f = win[k] cos(2 pi f k / N), rows nf + f = -win[k] sin(2 pi f k / N), with the
    periodic Hann window of win_length centred in the n_fft frame (what torch.stft does)."""
    N, wl = t.n_fft, t.win_length
    nf = N // 2 + 1
    win = np.zeros(N)
    lo = (N - wl) // 2
    win[lo: lo + wl] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(wl) / wl)
    k = np.arange(N)[None, :]
    f = np.arange(nf)[:, None]
    ang = 2 * np.pi * ((f * k) % N) / N
    return np.concatenate([np.cos(ang) * win, -np.sin(ang) * win], axis=0).astype(np.float32)


def enc_cfg_struct(w: Wav2Vec2Cfg, t: TokCfg, max_samples: int, max_ref_samples: int,
                   exact_fp32: Optional[bool] = None) -> _lib.EncCfg:
    """``exact_fp32``: every contraction on the exact-fp32 matrix pipe (verification mode) instead of the bf16-split pipe for
    the transformer projections and the ConvNeXt stack; None reads SPARKMI_ENC_EXACT=1 from the environment."""
    import os
    if exact_fp32 is None:
        exact_fp32 = os.environ.get("SPARKMI_ENC_EXACT") == "1"
    w.validate()
    t.validate()
    s = _lib.EncCfg(
        w2v_conv_dim=w.conv_dim[0], w2v_nconv=len(w.conv_dim), w2v_hidden=w.hidden_size, w2v_layers=w.used_layers,
        w2v_heads=w.num_attention_heads, w2v_inter=w.intermediate_size, w2v_pos_k=w.num_conv_pos_embeddings,
        w2v_pos_groups=w.num_conv_pos_embedding_groups, w2v_eps=w.layer_norm_eps,
        enc_in=t.enc_input_channels, enc_dim=t.enc_vocos_dim, enc_inter=t.enc_intermediate_dim, enc_layers=t.enc_num_layers,
        enc_out=t.enc_out_channels, enc_num_down=len(t.enc_sample_ratios), codebook_size=t.codebook_size,
        codebook_dim=t.codebook_dim, n_fft=t.n_fft, win_length=t.win_length, hop_length=t.hop_length, num_mels=t.num_mels,
        ecapa_channels=t.ecapa_channels, ecapa_out=t.ecapa_out, spk_latent=t.spk_latent_dim, spk_tokens=t.spk_token_num,
        fsq_dims=len(t.fsq_levels), perc_depth=t.perceiver_depth, perc_heads=t.perceiver_heads, perc_ff_inner=t.ff_inner,
        max_samples=max_samples, max_ref_samples=max_ref_samples, exact_fp32=int(bool(exact_fp32)))
    for i, (k, st) in enumerate(zip(w.conv_kernel, w.conv_stride)):
        s.w2v_kernel[i], s.w2v_stride[i] = k, st
    for i, v in enumerate(w.taps):
        s.w2v_taps[i] = v
    for i, v in enumerate(t.fsq_levels):
        s.fsq_levels[i] = v
    return s


def pack_enc_arena(t: TokCfg, w2v_state: Mapping[str, np.ndarray], tok_state_folded: Mapping[str, np.ndarray],
                   cs: _lib.EncCfg) -> np.ndarray:
    """``w2v_state``: transformers keys with the positional-conv weight norm folded
    (``weights.fold_pos_conv_weight_norm``); ``tok_state_folded``: BiCodec keys after ``fold_weight_norm``."""
    lib = _lib.lib()
    n = lib.smi_enc_arena_count(C.byref(cs))
    total = lib.smi_enc_arena_bytes(C.byref(cs))
    if n <= 0 or total == 0:
        raise _lib.SparkMIError("smi_enc_arena_count: config outside the kernel contract")

    def tensor(key: str) -> np.ndarray:
        if key.startswith("cat:"):
            return np.concatenate([tensor(k) for k in key[4:].split("|")], axis=0)
        if key.startswith("w2v."):
            return np.asarray(w2v_state[key[4:]], np.float32)
        if key.startswith(("bnscale:", "bnshift:")):
            p = key.split(":", 1)[1]
            g, b = tok_state_folded[p + ".weight"].astype(np.float32), tok_state_folded[p + ".bias"].astype(np.float32)
            m, v = tok_state_folded[p + ".running_mean"].astype(np.float32), tok_state_folded[p + ".running_var"].astype(np.float32)
            scale = (g / np.sqrt(v + np.float32(1e-5))).astype(np.float32)
            return scale if key.startswith("bnscale:") else (b - m * scale).astype(np.float32)
        if key.startswith("transpose:"):
            return np.ascontiguousarray(np.asarray(tok_state_folded[key[10:]], np.float32).T)
        if key == "mel.dft":
            return dft_basis(t)
        if key == "mel.fb":
            return mel_filterbank(t)
        return np.asarray(tok_state_folded[key], np.float32)

    arena = np.zeros(total // 4, dtype=np.float32)
    name = C.create_string_buffer(8192)
    for i in range(n):
        off, nb = C.c_size_t(), C.c_size_t()
        info = (C.c_int32 * 6)()
        _lib.check(lib.smi_enc_arena_entry(C.byref(cs), i, name, 8192, C.byref(off), C.byref(nb), info), "smi_enc_arena_entry")
        key = name.value.decode()
        arr = tensor(key)
        kind, cout, cin, K, S, pad = list(info)
        data = (arr.reshape(-1) if kind == PACK_RAW else
                pack_conv_b(arr, kind, S, pad) if kind in (PACK_CONV_B, PACK_CONVT_B) else pack_conv(arr, kind, S, pad))
        if data.size * 4 != nb.value:
            raise ValueError(f"{key}: packed {data.size * 4} bytes, library expects {nb.value}")
        arena[off.value // 4: off.value // 4 + data.size] = data
    return arena


class BiCodecEncoder:
    """wav (+ reference clip) -> (global ids (1, 1, Ntok) int32, semantic ids (1, T) int64) on one MI355X."""

    def __init__(self, wcfg: Wav2Vec2Cfg, tcfg: TokCfg, w2v_state: Optional[Mapping[str, np.ndarray]],
                 tok_state_folded: Optional[Mapping[str, np.ndarray]], device: Union[str, torch.device] = "cuda:0",
                 max_seconds: float = 30.0, ref_seconds: float = 6.0, arena: Optional[torch.Tensor] = None,
                 exact_fp32: Optional[bool] = None, diag: bool = False):
        self.wcfg, self.tcfg = wcfg, tcfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SparkMIError("BiCodecEncoder runs on an MI355X only (device must be cuda:N); there is no CPU path")
        self._lib = _lib.pick(diag)
        torch.cuda.set_device(self.device)
        _lib.require_gfx950()
        self.max_samples = int(max_seconds * tcfg.sample_rate)
        self.max_ref = int(ref_seconds * tcfg.sample_rate) + tcfg.n_fft
        self._cs = enc_cfg_struct(wcfg, tcfg, self.max_samples, self.max_ref, exact_fp32)
        self.exact_fp32 = bool(self._cs.exact_fp32)
        if arena is None:
            arena = torch.from_numpy(pack_enc_arena(tcfg, w2v_state, tok_state_folded, self._cs)).to(self.device)
        self.arena = arena
        self._h = C.c_void_p()
        self._lib.check(self._lib.smi_enc_create(C.byref(self._cs), C.c_void_p(arena.data_ptr()), arena.numel() * 4,
                                            C.byref(self._h)), "smi_enc_create")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.smi_enc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @torch.no_grad()
    def tokenize_arrays(self, wav: np.ndarray, ref_wav: np.ndarray) -> Tuple[torch.Tensor, torch.Tensor]:
        """``wav``: the whole prompt (after load_audio); ``ref_wav``: the reference clip.  Returns
        (global ids (1, 1, Ntok) int32, semantic ids (1, T) int64) on the device, like
        ``BiCodecTokenizer.tokenize`` (audio_tokenizer.py:119-130)."""
        w = torch.from_numpy(np.ascontiguousarray(wav, dtype=np.float32)).to(self.device)
        r = torch.from_numpy(np.ascontiguousarray(np.asarray(ref_wav).reshape(-1), dtype=np.float32)).to(self.device)
        if w.numel() > self.max_samples:
            raise ValueError(f"prompt of {w.numel()} samples exceeds max_samples={self.max_samples}")
        frames = self.wcfg.frames(w.numel())
        sem = torch.empty((1, max(frames, 1)), dtype=torch.int64, device=self.device)
        glob = torch.empty((1, 1, self.tcfg.spk_token_num), dtype=torch.int32, device=self.device)
        n = C.c_int(0)
        self._lib.check(self._lib.smi_enc_forward(self._h, C.c_void_p(w.data_ptr()), w.numel(), C.c_void_p(r.data_ptr()), r.numel(),
                                             C.c_void_p(sem.data_ptr()), C.c_void_p(glob.data_ptr()), C.byref(n), self._stream()),
                   "smi_enc_forward")
        assert n.value == frames
        return glob, sem

    @torch.no_grad()
    def tokenize_many(self, wavs: Sequence[np.ndarray], refs: Sequence[np.ndarray], lanes: int = 8):
        """Several prompts at once: prompt i runs on HIP stream i mod ``lanes`` with its own handle (own scratch; the weight
        arena is shared).  One encode is ~260 small launches (a 6 s prompt is 299 frames: grids of tens of blocks), so
        independent encodes fill the chip side by side -- 8 prompts: 68 ms one after the other, 35 ms on 8 streams
        (tools/enc_streams.py); the ids are those of ``tokenize_arrays``, prompt by prompt.  Returns [(global, semantic)];
        the current stream waits for all lanes."""
        n = max(1, min(int(lanes), len(wavs)))
        if not hasattr(self, "_lanes"):
            self._lanes = [(self, torch.cuda.Stream(self.device))]
        while len(self._lanes) < n:
            sib = BiCodecEncoder(self.wcfg, self.tcfg, None, None, self.device, max_seconds=self.max_samples / self.tcfg.sample_rate,
                                 ref_seconds=(self.max_ref - self.tcfg.n_fft) / self.tcfg.sample_rate, arena=self.arena,
                                 exact_fp32=self.exact_fp32)
            self._lanes.append((sib, torch.cuda.Stream(self.device)))
        cur = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(cur)
        out = []
        for i, (w, r) in enumerate(zip(wavs, refs)):
            enc, st = self._lanes[i % n]
            if i < n:
                st.wait_event(start)
            with torch.cuda.stream(st):
                g, sm = enc.tokenize_arrays(w, r)
            g.record_stream(cur)      # produced on the lane's stream, consumed on the caller's: keep the allocator from
            sm.record_stream(cur)     # handing the blocks to a later lane-stream allocation while the caller still reads them
            out.append((g, sm))
        for enc, st in self._lanes[:n]:
            ev = torch.cuda.Event()
            ev.record(st)
            cur.wait_event(ev)
        return out

    def debug_stage(self, name: str) -> torch.Tensor:
        out = torch.empty(64 * 1024 * 1024 // 4, dtype=torch.float32, device=self.device)
        dims = (C.c_int32 * 2)()
        self._lib.check(self._lib.smi_enc_debug_stage(self._h, name.encode(), C.c_void_p(out.data_ptr()), out.numel(), dims, self._stream()),
                   "smi_enc_debug_stage")
        return out[: dims[0] * dims[1]].reshape(dims[0], dims[1]).clone()

    def launches(self) -> int:
        return self._lib.smi_enc_num_launches(self._h)

    def time_launch(self, index: int, iters: int = 5):
        ms, fl = C.c_float(0), C.c_double(0)
        name = C.create_string_buffer(512)
        self._lib.check(self._lib.smi_enc_time_launch(self._h, index, iters, C.byref(ms), C.byref(fl), name, 512, self._stream()),
                   "smi_enc_time_launch")
        return name.value.decode(), float(ms.value), float(fl.value)
