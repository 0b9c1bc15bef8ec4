"""BiCodec detokenize on the HIP vocoder (``smi_voc.hip``), behind the reference's own seams:

* ``BiCodecVocoder.detokenize(semantic_tokens, global_tokens)`` mirrors ``BiCodec.detokenize``
  (``sparktts/models/bicodec.py:171-189``): (B, T) + (B, 1, Ntok) -> (B, 1, hop*T) tensor.
* ``BiCodecTokenizer.detokenize(global_tokens, semantic_tokens)`` mirrors the facade
  (``sparktts/models/audio_tokenizer.py:132-146``): (B, Ntok) + (B, T) -> squeezed numpy.

Weight packing for the implicit-GEMM kernel lives here; section offsets and packing kinds come
from the library (``smi_voc_arena_entry``).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Tuple,  Dict, List, Mapping, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .config import BiCodecConfig, TopConfig
from .weights import fold_weight_norm, load_bicodec_state

PACK_RAW, PACK_CONV, PACK_CONVT, PACK_CONV_B, PACK_CONVT_B = 0, 1, 2, 3, 4


def voc_cfg_struct(cfg: BiCodecConfig, max_batch: int, max_frames: int, exact_fp32: Optional[bool] = None) -> _lib.VocCfg:
    """``exact_fp32``: every contraction on the exact-fp32 matrix pipe (verification mode) instead of the bf16-split pipe;
    None reads SPARKMI_VOC_EXACT=1 from the environment."""
    import os
    if exact_fp32 is None:
        exact_fp32 = os.environ.get("SPARKMI_VOC_EXACT") == "1"
    cfg.validate()
    s = _lib.VocCfg(
        vq_input_dim=cfg.vq_input_dim, codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
        spk_out_dim=cfg.spk_out_dim, spk_latent_dim=cfg.spk_latent_dim, spk_token_num=cfg.spk_token_num,
        fsq_dims=len(cfg.fsq_levels),
        pre_input_channels=cfg.pre_input_channels, pre_dim=cfg.pre_vocos_dim, pre_inter=cfg.pre_intermediate_dim,
        pre_layers=cfg.pre_num_layers, pre_out_channels=cfg.pre_out_channels,
        pre_cond_dim=cfg.pre_condition_dim or 0, pre_num_down=len(cfg.pre_sample_ratios),
        pre_tanh_final=int(cfg.pre_use_tanh_at_final),
        dec_in=cfg.dec_input_channel, dec_channels=cfg.dec_channels, dec_nblocks=len(cfg.dec_rates),
        max_batch=max_batch, max_frames=max_frames, exact_fp32=int(bool(exact_fp32)))
    for i, v in enumerate(cfg.fsq_levels):
        s.fsq_levels[i] = v
    for i, (r, k) in enumerate(zip(cfg.dec_rates, cfg.dec_kernel_sizes)):
        s.dec_rates[i], s.dec_ksizes[i] = r, k
    return s


def conv_phases(K: int, S: int, pad: int):
    """Kernel indices j of each output phase r of a ConvTranspose1d (stride S, padding pad):
    out[q*S + r] takes taps j = j0 + S*i with j0 = (r + pad) mod S.  For S == 1: all taps."""
    if S == 1:
        return [list(range(K))]
    return [list(range((r + pad) % S, K, S)) for r in range(S)]


def pack_conv(w: np.ndarray, kind: int, S: int, pad: int) -> np.ndarray:
    """Weights -> [phase][cout_tile][tap][cin_group][lane:64][4] floats, the A operand of
    v_mfma_f32_32x32x2_f32: lane l, slot j holds A[co = 32*tile + (l & 31)][ci = 8*group + 2*j + (l >> 5)]."""
    w = np.asarray(w, dtype=np.float32)
    if w.ndim == 2:
        w = w[:, :, None]
    if kind == PACK_CONVT:
        w = w.transpose(1, 0, 2)          # (Cin, Cout, K) -> (Cout, Cin, K)
    cout, cin, K = w.shape
    cop, cip = (cout + 31) // 32 * 32, (cin + 7) // 8 * 8
    wp = np.zeros((cop, cip, K), np.float32)
    wp[:cout, :cin] = w
    out = []
    for taps in conv_phases(K, S, pad):
        a = wp[:, :, taps]                                    # (cop, cip, ntap)
        a = a.reshape(cop // 32, 32, cip // 8, 4, 2, len(taps))   # [ct][row][g][j][half][tap]
        a = a.transpose(0, 5, 2, 4, 1, 3)                     # [ct][tap][g][half][row][j]
        out.append(np.ascontiguousarray(a).reshape(-1))       # lane = half*32 + row
    return np.concatenate(out)


def pack_conv_b(w: np.ndarray, kind: int, S: int, pad: int) -> np.ndarray:
    """Weights -> two bf16 planes (hi = bf16(w), mid = bf16(w - hi)) in the A-operand order of v_mfma_f32_32x32x16_bf16:
    [phase][cout_tile][tap][16-channel step][plane][lane:64][8 bf16], lane l holding A[co = 32*tile + (l & 31)]
    [ci = 16*step + 8*(l >> 5) + 0..7].  Returned as float32 words (the arena's element type; the bytes are bf16)."""
    from .weights import bf16_bits_to_f32, f32_to_bf16_bits
    w = np.asarray(w, dtype=np.float32)
    if w.ndim == 2:
        w = w[:, :, None]
    if kind == PACK_CONVT_B:
        w = w.transpose(1, 0, 2)          # (Cin, Cout, K) -> (Cout, Cin, K)
    cout, cin, K = w.shape
    cop, cip = (cout + 31) // 32 * 32, (cin + 15) // 16 * 16
    wp = np.zeros((cop, cip, K), np.float32)
    wp[:cout, :cin] = w
    hi = f32_to_bf16_bits(wp).reshape(wp.shape)
    mid = f32_to_bf16_bits(wp - bf16_bits_to_f32(hi).reshape(wp.shape)).reshape(wp.shape)
    planes = np.stack([hi, mid], axis=0)                      # (2, cop, cip, K) uint16
    out = []
    for taps in conv_phases(K, S, pad):
        a = planes[:, :, :, taps]                             # (2, cop, cip, ntap)
        a = a.reshape(2, cop // 32, 32, cip // 16, 2, 8, len(taps))   # [plane][ct][row][step][half][e][tap]
        a = a.transpose(1, 6, 3, 0, 4, 2, 5)                  # [ct][tap][step][plane][half][row][e]
        out.append(np.ascontiguousarray(a).reshape(-1))       # lane = half*32 + row
    return np.concatenate(out).view(np.float32)


def _pack_entries(count_fn, bytes_fn, entry_fn, cs, folded: Mapping[str, np.ndarray], what: str) -> np.ndarray:
    """Pack the tensors a layout enumerates (the vocoder's, or one block's) into a flat f32 arena."""
    n = count_fn(C.byref(cs))
    total = bytes_fn(C.byref(cs))
    if n <= 0 or total == 0:
        raise _lib.SparkMIError(f"{what}: config outside the kernel contract")
    arena = np.zeros(total // 4, dtype=np.float32)
    name = C.create_string_buffer(8192)
    for i in range(n):
        off, nb = C.c_size_t(), C.c_size_t()
        info = (C.c_int32 * 6)()
        _lib.check(entry_fn(C.byref(cs), i, name, 8192, C.byref(off), C.byref(nb), info), what)
        key = name.value.decode()
        if key.startswith("cat:"):
            t = np.concatenate([np.asarray(folded[k], np.float32) for k in key[4:].split("|")], axis=0)
        else:
            t = np.asarray(folded[key], np.float32)
        kind, cout, cin, K, S, pad = list(info)
        data = (t.reshape(-1) if kind == PACK_RAW else
                pack_conv_b(t, kind, S, pad) if kind in (PACK_CONV_B, PACK_CONVT_B) else pack_conv(t, kind, S, pad))
        if data.size * 4 != nb.value:
            raise ValueError(f"{key}: packed {data.size * 4} bytes, library expects {nb.value}")
        arena[off.value // 4: off.value // 4 + data.size] = data
    return arena


def pack_voc_arena(cfg: BiCodecConfig, folded: Mapping[str, np.ndarray], cs: _lib.VocCfg) -> np.ndarray:
    lib = _lib.lib()
    return _pack_entries(lib.smi_voc_arena_count, lib.smi_voc_arena_bytes, lib.smi_voc_arena_entry, cs, folded, "smi_voc_arena_entry")


BLOCK_RESUNIT, BLOCK_DECBLOCK, BLOCK_CONVNEXT = 0, 1, 2


@torch.no_grad()
def run_block(kind: int, params: Mapping[str, np.ndarray], x: Optional[torch.Tensor], xs: Optional[torch.Tensor] = None,
              cond: Optional[torch.Tensor] = None, lens: Optional[Sequence[int]] = None, *, dil: int = 1, K: int = 0, S: int = 1,
              exact_fp32: bool = False) -> torch.Tensor:
    """ONE block of the vocoder on the HIP kernels (``smi_voc_block_run``), built by the functions ``smi_voc_forward`` builds its
    launches with -- the op-level seam for the reference's own layer classes: ResidualUnit (``blocks/layers.py:51-67``),
    DecoderBlock (``encoder_decoder/wave_generator.py:29-53``), ConvNeXtBlock (``blocks/vocos.py:26-110``).
    ``params``: the layer's state_dict (after remove_weight_norm) with keys prefixed "L."; ``x`` (B, C, L) on the GPU; ``xs`` =
    snake(x, block.0.alpha), which the vocoder leaves to the PRODUCER's epilogue (needed by RESUNIT and DECBLOCK)."""
    lib = _lib.lib()
    _lib.require_gfx950()
    t = x if x is not None else xs
    B, Cin, L = t.shape
    cs = _lib.VocBlockCfg(kind=kind, C=Cin, exact_fp32=int(exact_fp32), dil=dil, K=K, S=S)
    if kind == BLOCK_DECBLOCK:
        cs.Cout = int(np.asarray(params["L.block.1.weight"]).shape[1])
    if kind == BLOCK_CONVNEXT:
        cs.I = int(np.asarray(params["L.pwconv1.weight"]).shape[0])
        cs.cond_dim = 0 if cond is None else int(cond.shape[-1])
    flat = {k: np.asarray(v, np.float32) for k, v in params.items()}
    arena = torch.from_numpy(_pack_entries(lib.smi_voc_block_arena_count, lib.smi_voc_block_arena_bytes, lib.smi_voc_block_arena_entry,
                                           cs, flat, "smi_voc_block_arena_entry")).to(t.device)
    Cout = cs.Cout if kind == BLOCK_DECBLOCK else Cin
    y = torch.empty(B, Cout, L * (S if kind == BLOCK_DECBLOCK else 1), dtype=torch.float32, device=t.device)
    ptr = lambda a: C.c_void_p(a.contiguous().data_ptr()) if a is not None else None
    keep = [a.contiguous().float() if a is not None else None for a in (x, xs, cond)]
    hl = (C.c_int32 * B)(*[int(v) for v in lens]) if lens is not None else None
    _lib.check(lib.smi_voc_block_run(C.byref(cs), C.c_void_p(arena.data_ptr()), arena.numel() * 4, ptr(keep[0]), ptr(keep[1]), ptr(keep[2]),
                                     hl, B, L, C.c_void_p(y.data_ptr()), C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)),
               "smi_voc_block_run")
    return y


class BiCodecVocoder:
    """The vocoder half of BiCodec on one MI355X."""

    def __init__(self, cfg: BiCodecConfig, state: Mapping[str, np.ndarray],
                 device: Union[str, torch.device] = "cuda:0", max_batch: int = 1, max_frames: int = 512,
                 state_is_folded: bool = False, arena: Optional[torch.Tensor] = None, exact_fp32: Optional[bool] = None,
                 diag: bool = False):
        """``exact_fp32``: run every contraction on the exact-fp32 matrix pipe (verification mode); default (None ->
        SPARKMI_VOC_EXACT) is the bf16-split pipe, 5e-5 max-abs from it on the waveform.  An ``arena`` must have been
        packed for the same mode."""
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SparkMIError("BiCodecVocoder runs on an MI355X only (device must be cuda:N); there is no CPU path")
        self._lib = _lib.pick(diag)   # diag: libsparkmi_diag.so (SPARKMI_VOC_DEBUG stage dumps and the other switches live there)
        torch.cuda.set_device(self.device)
        _lib.require_gfx950()
        self.max_batch, self.max_frames = max_batch, max_frames
        self._cs = voc_cfg_struct(cfg, max_batch, max_frames, exact_fp32)
        self.exact_fp32 = bool(self._cs.exact_fp32)
        if arena is None:
            folded = state if state_is_folded else fold_weight_norm(dict(state))
            arena = torch.from_numpy(pack_voc_arena(cfg, folded, self._cs)).to(self.device)
        self.arena = arena   # float32 device tensor; must outlive the handle
        self._h = C.c_void_p()
        self._lib.check(self._lib.smi_voc_create(C.byref(self._cs), C.c_void_p(self.arena.data_ptr()),
                                            self.arena.numel() * 4, C.byref(self._h)), "smi_voc_create")
        self.hop = cfg.hop

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.smi_voc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @torch.no_grad()
    def detokenize(self, semantic_tokens: torch.Tensor, global_tokens: torch.Tensor,
                   lengths: Optional[Sequence[int]] = None) -> torch.Tensor:
        """(B, T) semantic ids + (B, 1, Ntok) global ids -> (B, 1, hop*T) float32 on the device.
        ``lengths`` (frames per row) makes the batch ragged: each row then equals an un-padded
        run of that row alone, and its samples beyond hop*length are zero."""
        sem = semantic_tokens.to(self.device, torch.int64).contiguous()
        if sem.ndim == 1:
            sem = sem[None]
        B, T = sem.shape
        glob = global_tokens.to(self.device, torch.int32).reshape(B, -1).contiguous()
        if glob.shape[1] != self.cfg.spk_token_num:
            raise ValueError(f"expected {self.cfg.spk_token_num} global tokens per row, got {glob.shape[1]}")
        lens = np.full(B, T, np.int32) if lengths is None else np.asarray(lengths, np.int32)
        wav = torch.empty((B, 1, self.hop * T), dtype=torch.float32, device=self.device)
        self._lib.check(self._lib.smi_voc_forward(
            self._h, C.c_void_p(sem.data_ptr()), lens.ctypes.data_as(C.POINTER(C.c_int32)),
            C.c_void_p(glob.data_ptr()), B, T, C.c_void_p(wav.data_ptr()), self._stream()), "smi_voc_forward")
        self._keep = (sem, glob)   # inputs must stay alive until the stream has consumed them
        return wav

    # ------------------------------------------------------------------ test / bench entries
    def debug_stage(self, stage: int, channels: int, length: int, batch: int) -> torch.Tensor:
        """Stage activation of the last forward as (B, C, L).  -1 = d-vector (returns (B, out_dim));
        stages >= 0 need SPARKMI_VOC_DEBUG=1 at construction: 0 z_q, 1 prenet+d, 2 conv_in, 3+i block i."""
        n = C.c_size_t()
        if stage == -1:
            out = torch.empty(batch * self.cfg.spk_out_dim, dtype=torch.float32, device=self.device)
            self._lib.check(self._lib.smi_voc_debug_stage(self._h, -1, C.c_void_p(out.data_ptr()), out.numel(),
                                                     C.byref(n), self._stream()), "smi_voc_debug_stage")
            return out.view(batch, -1)
        big = self._dbg_floats() * batch
        out = torch.empty(big, dtype=torch.float32, device=self.device)
        self._lib.check(self._lib.smi_voc_debug_stage(self._h, stage, C.c_void_p(out.data_ptr()), big, C.byref(n),
                                                 self._stream()), "smi_voc_debug_stage")
        per = n.value // batch
        return out.view(batch, per)[:, : channels * length].reshape(batch, channels, length)

    def _dbg_floats(self) -> int:
        c, T = self.cfg, self.max_frames
        mx = max(c.vq_input_dim * T, c.pre_intermediate_dim * T, c.dec_channels * T)
        L = T
        for i, r in enumerate(c.dec_rates):
            L *= r
            mx = max(mx, (c.dec_channels >> (i + 1)) * L)
        return (mx + 63) // 64 * 64

    def launches(self) -> int:
        return int(self._lib.smi_voc_num_launches(self._h))

    def time_launch(self, index: int, iters: int = 10):
        ms, fl = C.c_float(0), C.c_double(0)
        name = C.create_string_buffer(256)
        self._lib.check(self._lib.smi_voc_time_launch(self._h, index, iters, C.byref(ms), C.byref(fl), name, 256,
                                                 self._stream()), "smi_voc_time_launch")
        return name.value.decode(), float(ms.value), float(fl.value)


class BiCodecTokenizer:
    """Drop-in for ``sparktts.models.audio_tokenizer.BiCodecTokenizer``: ``tokenize`` (voice-clone prompt
    encode: wav2vec2 + BiCodec encoder / VQ + speaker encoder, ``sparkmi/encoder.py``) and ``detokenize``
    (the vocoder).  The prompt encoder (a 1 GB arena at full size) is built on the first ``tokenize``."""

    def __init__(self, model_dir: Optional[Path] = None, device: Union[str, torch.device] = None,
                 cfg: Optional[BiCodecConfig] = None, state: Optional[Mapping[str, np.ndarray]] = None,
                 max_batch: int = 1, max_frames: int = 3000, max_prompt_seconds: float = 30.0, **kwargs):
        self.device = torch.device(device if device is not None else "cuda:0")
        self.model_dir = model_dir
        self._state = None
        if cfg is None:
            bdir = Path(model_dir) / "BiCodec"
            cfg = BiCodecConfig.from_yaml(bdir / "config.yaml")
            state = load_bicodec_state(bdir)
            self._state = state
            self.config = TopConfig.from_yaml(Path(model_dir) / "config.yaml")
        else:
            self.config = TopConfig()
        self.model = BiCodecVocoder(cfg, state, self.device, max_batch=max_batch, max_frames=max_frames)
        self._enc = None
        self._max_prompt_seconds = max_prompt_seconds

    # ---------------------------------------------------------------- prompt encode (audio_tokenizer.py:57-130)
    def _encoder(self):
        if self._enc is None:
            from .config_tok import TokCfg, Wav2Vec2Cfg
            from .encoder import BiCodecEncoder
            from .weights import fold_pos_conv_weight_norm, fold_weight_norm, load_wav2vec2_state
            if self.model_dir is None:
                raise _lib.SparkMIError("tokenize needs a model directory (wav2vec2-large-xlsr-53/ and BiCodec/)")
            wdir = Path(self.model_dir) / "wav2vec2-large-xlsr-53"
            if not (wdir / "config.json").exists():
                raise FileNotFoundError(f"{wdir}/config.json: the wav2vec2 feature extractor of the prompt encoder is missing")
            wcfg = Wav2Vec2Cfg.from_json(wdir / "config.json")
            tcfg = TokCfg.from_yaml(Path(self.model_dir) / "BiCodec" / "config.yaml")
            self._enc = BiCodecEncoder(wcfg, tcfg, fold_pos_conv_weight_norm(load_wav2vec2_state(wdir)),
                                       fold_weight_norm(self._state), self.device, max_seconds=self._max_prompt_seconds,
                                       ref_seconds=float(self.config.ref_segment_duration))
        return self._enc

    def get_ref_clip(self, wav: np.ndarray) -> np.ndarray:
        """Reference clip for the speaker embedding (audio_tokenizer.py:57-72)."""
        from .encoder import get_ref_clip
        return get_ref_clip(wav, self.config.sample_rate, self.config.ref_segment_duration, self.config.latent_hop_length)

    def process_audio(self, wav_path) -> Tuple[np.ndarray, torch.Tensor]:
        """Load the prompt and cut its reference clip (audio_tokenizer.py:74-83)."""
        from .encoder import load_audio
        wav = load_audio(wav_path, sampling_rate=self.config.sample_rate, volume_normalize=self.config.volume_normalize)
        ref = torch.from_numpy(self.get_ref_clip(wav)).unsqueeze(0).float()
        return wav, ref

    def extract_wav2vec2_features(self, wavs) -> torch.Tensor:
        """(1, T, hidden) mean of the tapped wav2vec2 hidden states (audio_tokenizer.py:85-100)."""
        wav = np.asarray(wavs, dtype=np.float32).reshape(-1)
        enc = self._encoder()
        enc.tokenize_arrays(wav, self.get_ref_clip(wav))
        return enc.debug_stage("feat").t().unsqueeze(0)

    def tokenize(self, audio_path: str) -> Tuple[torch.Tensor, torch.Tensor]:
        """Prompt audio file -> (global ids (1, 1, Ntok), semantic ids (1, T)) (audio_tokenizer.py:119-130)."""
        wav, ref = self.process_audio(audio_path)
        return self._encoder().tokenize_arrays(wav, ref.numpy())

    def tokenize_batch(self, batch) -> Tuple[torch.Tensor, torch.Tensor]:
        """audio_tokenizer.py:102-117 for a list of equally long prompts: {"wav": [...], "ref_wav": (B, L)}."""
        g, s = zip(*self._encoder().tokenize_many([np.asarray(w) for w in batch["wav"]], [np.asarray(r) for r in batch["ref_wav"]]))
        return torch.cat(g, 0), torch.cat(s, 0)

    def tokenize_many(self, audio_paths: Sequence[str]):
        """Several prompt files at once: [(global ids (1, 1, Ntok), semantic ids (1, T_i))], each equal to ``tokenize(path)``;
        the encodes run side by side on parallel HIP streams (``BiCodecEncoder.tokenize_many``)."""
        wavs, refs = zip(*[self.process_audio(p) for p in audio_paths])
        return self._encoder().tokenize_many(list(wavs), [r.numpy() for r in refs])

    def detokenize(self, global_tokens: torch.Tensor, semantic_tokens: torch.Tensor) -> np.ndarray:
        """(B, Ntok) global ids, (B, T) semantic ids -> waveform: (hop*T,) for B == 1 else (B, hop*T)."""
        global_tokens = global_tokens.unsqueeze(1)
        wav_rec = self.model.detokenize(semantic_tokens, global_tokens)
        return wav_rec.detach().squeeze().cpu().numpy()
