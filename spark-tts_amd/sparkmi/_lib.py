"""ctypes binding of ``libsparkmi.so`` (``include/sparkmi.h``) and of ``libsparkmi_diag.so`` (the same sources built with
``-DSMI_DIAG``: ``include/sparkmi_debug.h``'s entry points, the SPARKMI_* A/B switches, the experimental one-row engine).

``lib()`` is the product library: what ``SparkTTS`` / ``SparkLLM`` / ``BiCodecVocoder`` / ``BiCodecEncoder`` run on by default; it
reads no environment variable.  ``diag()`` is the diagnostics build: ``SparkLLM(..., diag=True)`` (and the vocoder / encoder
likewise) put a handle on it -- tools/, bench.py's per-kernel probes and the tests that look inside a step do.  A handle belongs
to the library that created it.  Both are built in-tree by ``spark-tts_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is no fallback: a missing library raises at first use.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["SPARKMI_LIB"]) if os.environ.get("SPARKMI_LIB") else _HERE / "libsparkmi.so"

SMI_MAX_ROWS = 64
SMI_MAX_EOS = 4
ABI_VERSION = 4


class SparkMIError(RuntimeError):
    pass


class LLMCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "vocab_size", "hidden_size", "num_layers", "num_heads", "num_kv_heads", "head_dim",
        "intermediate_size", "max_slots", "max_positions", "kv_dtype", "use_graph")] + [("rms_eps", C.c_float),
                                                                                        ("kv_page_tokens", C.c_int32), ("kv_pages", C.c_int32), ("wd_plain", C.c_int32),
                                                                                        ("weights_exact", C.c_int32)]


class VocCfg(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in (
        "vq_input_dim", "codebook_size", "codebook_dim",
        "spk_out_dim", "spk_latent_dim", "spk_token_num", "fsq_dims")]
        + [("fsq_levels", C.c_int32 * 8)]
        + [(n, C.c_int32) for n in (
            "pre_input_channels", "pre_dim", "pre_inter", "pre_layers", "pre_out_channels",
            "pre_cond_dim", "pre_num_down", "pre_tanh_final", "dec_in", "dec_channels", "dec_nblocks")]
        + [("dec_rates", C.c_int32 * 8), ("dec_ksizes", C.c_int32 * 8)]
        + [("max_batch", C.c_int32), ("max_frames", C.c_int32), ("exact_fp32", C.c_int32)])


class VocBlockCfg(C.Structure):
    """smi_voc_block_cfg: one reference block (ResidualUnit 0 / DecoderBlock 1 / ConvNeXtBlock 2) for the op-level entry points"""
    _fields_ = [(n, C.c_int32) for n in ("kind", "C", "Cout", "K", "S", "dil", "I", "cond_dim", "exact_fp32")]


class EncCfg(C.Structure):
    _fields_ = ([("w2v_conv_dim", C.c_int32), ("w2v_nconv", C.c_int32), ("w2v_kernel", C.c_int32 * 8), ("w2v_stride", C.c_int32 * 8)]
                + [(n, C.c_int32) for n in ("w2v_hidden", "w2v_layers", "w2v_heads", "w2v_inter", "w2v_pos_k", "w2v_pos_groups")]
                + [("w2v_taps", C.c_int32 * 3), ("w2v_eps", C.c_float)]
                + [(n, C.c_int32) for n in ("enc_in", "enc_dim", "enc_inter", "enc_layers", "enc_out", "enc_num_down",
                                            "codebook_size", "codebook_dim", "n_fft", "win_length", "hop_length", "num_mels",
                                            "ecapa_channels", "ecapa_out", "spk_latent", "spk_tokens", "fsq_dims")]
                + [("fsq_levels", C.c_int32 * 8)]
                + [(n, C.c_int32) for n in ("perc_depth", "perc_heads", "perc_ff_inner", "max_samples", "max_ref_samples", "exact_fp32")])


# section ids of enum smi_llm_section
(LLM_LN1, LLM_WQKV, LLM_BQKV, LLM_WO, LLM_LN2, LLM_WGU, LLM_WD,
 LLM_FINAL_NORM, LLM_LM_HEAD, LLM_ROPE, LLM_TAG) = range(11)


class LLMArenaTag(C.Structure):
    """smi_llm_arena_tag: how an arena was packed (section LLM_TAG); smi_llm_create checks it against the config."""
    _fields_ = [("magic", C.c_char * 8)] + [(n, C.c_int32) for n in (
        "abi_version", "wd_plain", "vocab_size", "hidden_size", "num_layers", "num_heads", "num_kv_heads", "intermediate_size",
        "max_positions", "weights_exact")] + [("reserved", C.c_int32 * 52)]

# every symbol include/sparkmi.h declares: (name, restype, argtypes)
_VP, _I, _SZ = C.c_void_p, C.c_int, C.c_size_t
_P = C.POINTER
SYMBOLS = {
    "smi_version": (_I, []),
    "smi_last_error": (C.c_char_p, []),
    "smi_device_check": (_I, [C.c_char_p, _I]),
    "smi_llm_arena_bytes": (_SZ, [_P(LLMCfg)]),
    "smi_llm_arena_section": (_I, [_P(LLMCfg), _I, _I, _P(_SZ), _P(_SZ)]),
    "smi_llm_create": (_I, [_P(LLMCfg), _VP, _SZ, _P(_VP)]),
    "smi_llm_destroy": (_I, [_VP]),
    "smi_llm_prefill": (_I, [_VP, _P(C.c_int64), _P(C.c_int32), _I, _I, _P(C.c_int64), _I, _VP]),
    "smi_llm_set_sampling": (_I, [_VP, _I, C.c_float, _I, C.c_float, C.c_uint64]),
    "smi_llm_decode": (_I, [_VP, _I, _VP]),
    "smi_llm_all_done": (_I, [_VP, _P(_I), _VP]),
    "smi_llm_get_tokens": (_I, [_VP, _P(C.c_int64), _P(C.c_int32), _I, _VP]),
    "smi_llm_session_begin": (_I, [_VP, _P(C.c_int64), _I, _VP]),
    "smi_llm_admit": (_I, [_VP, _P(C.c_int64), _P(C.c_int32), _I, _I, _P(C.c_int32), _VP]),
    "smi_llm_retire": (_I, [_VP, _I, _VP]),
    "smi_llm_slot_tokens": (_I, [_VP, _I, _P(C.c_int64), _I, _P(C.c_int32), _P(C.c_int32), _VP]),
    "smi_llm_retire_many": (_I, [_VP, _P(C.c_int32), _I, _VP]),
    "smi_llm_slots_tokens": (_I, [_VP, _P(C.c_int32), _I, _P(C.c_int64), _I, _P(C.c_int32), _P(C.c_int32), _VP]),
    "smi_llm_status": (_I, [_VP, _P(C.c_int32), _P(C.c_int32), _VP]),
    "smi_llm_forward_logits": (_I, [_VP, _P(C.c_int64), _I, _VP, _VP]),
    "smi_llm_steps": (_I, [_VP]),
    "smi_llm_kv_pages": (_I, [_VP, _P(C.c_int32), _P(C.c_int32)]),
    "smi_voc_arena_count": (_I, [_P(VocCfg)]),
    "smi_voc_arena_entry": (_I, [_P(VocCfg), _I, C.c_char_p, _I, _P(_SZ), _P(_SZ), _P(C.c_int32)]),
    "smi_voc_arena_bytes": (_SZ, [_P(VocCfg)]),
    "smi_voc_create": (_I, [_P(VocCfg), _VP, _SZ, _P(_VP)]),
    "smi_voc_destroy": (_I, [_VP]),
    "smi_voc_forward": (_I, [_VP, _VP, _P(C.c_int32), _VP, _I, _I, _VP, _VP]),
    "smi_voc_debug_stage": (_I, [_VP, _I, _VP, _SZ, _P(_SZ), _VP]),
    "smi_voc_num_launches": (_I, [_VP]),
    "smi_voc_time_launch": (_I, [_VP, _I, _I, _P(C.c_float), _P(C.c_double), C.c_char_p, _I, _VP]),
    "smi_voc_block_arena_count": (_I, [_P(VocBlockCfg)]),
    "smi_voc_block_arena_bytes": (_SZ, [_P(VocBlockCfg)]),
    "smi_voc_block_arena_entry": (_I, [_P(VocBlockCfg), _I, C.c_char_p, _I, _P(_SZ), _P(_SZ), _P(C.c_int32)]),
    "smi_voc_block_run": (_I, [_P(VocBlockCfg), _VP, _SZ, _VP, _VP, _VP, _P(C.c_int32), _I, _I, _VP, _VP]),
    "smi_enc_arena_count": (_I, [_P(EncCfg)]),
    "smi_enc_arena_entry": (_I, [_P(EncCfg), _I, C.c_char_p, _I, _P(_SZ), _P(_SZ), _P(C.c_int32)]),
    "smi_enc_arena_bytes": (_SZ, [_P(EncCfg)]),
    "smi_enc_create": (_I, [_P(EncCfg), _VP, _SZ, _P(_VP)]),
    "smi_enc_destroy": (_I, [_VP]),
    "smi_enc_forward": (_I, [_VP, _VP, _I, _VP, _I, _VP, _VP, _P(_I), _VP]),
    "smi_enc_debug_stage": (_I, [_VP, C.c_char_p, _VP, _SZ, _P(C.c_int32), _VP]),
    "smi_enc_num_launches": (_I, [_VP]),
    "smi_enc_time_launch": (_I, [_VP, _I, _I, _P(C.c_float), _P(C.c_double), C.c_char_p, _I, _VP]),
}

# include/sparkmi_debug.h: exported by libsparkmi_diag.so only
DEBUG_SYMBOLS = {
    "smi_llm_time_kernel": (_I, [_VP, _I, _I, _I, _P(C.c_float), _VP]),
    "smi_llm_debug_stamps": (_I, [_VP, _I, _I, _P(C.c_double)]),
    "smi_llm_engine": (_I, [_VP, _P(C.c_int32), _P(C.c_int32), C.c_char_p, _I]),
    "smi_llm_set_engine": (_I, [_VP, _I]),
    "smi_llm_engine_plan": (_I, [_P(LLMCfg), _I, _P(C.c_int32)]),
    "smi_llm_engine_stamps": (_I, [_VP, _P(C.c_double), _I]),
    "smi_llm_debug_hidden": (_I, [_VP, _P(C.c_float), _I]),
    "smi_llm_debug_read": (_I, [_VP, _I, _VP, _SZ, _P(_SZ)]),
    "smi_llm_debug_raw_stamps": (_I, [_VP, _P(C.c_uint64), _I]),
    "smi_llm_debug_sample": (_I, [_VP, _P(C.c_float), _I, C.c_uint64, _I, _P(C.c_int32)]),
    "smi_llm_debug_set_kv": (_I, [_VP, _I, _I, _I, _I, _P(C.c_float), _P(C.c_float)]),
    "smi_llm_debug_get_kv": (_I, [_VP, _I, _I, _I, _I, _P(C.c_float), _P(C.c_float)]),
    "smi_llm_debug_layer": (_I, [_VP, _I, _I, _P(C.c_int32), _P(C.c_float), _I]),
}

DIAG_PATH = LIB_PATH.with_name("libsparkmi_diag.so")


class _Lib:
    """A loaded library: attribute access goes to the CDLL (``l.smi_llm_decode(...)``); ``check`` raises with THIS library's
    thread-local error text."""

    def __init__(self, cdll: C.CDLL, path: Path, is_diag: bool):
        self._cdll, self.path, self.is_diag = cdll, path, is_diag

    def __getattr__(self, name):
        return getattr(self._cdll, name)

    def check(self, rc: int, what: str = "") -> None:
        if rc != 0:
            msg = self._cdll.smi_last_error().decode(errors="replace")
            raise SparkMIError(f"{what or self.path.name} failed (code {rc}): {msg}")


def _load(path: Path, symbols, is_diag: bool) -> _Lib:
    if not path.exists():
        raise SparkMIError(
            f"{path} is missing: build it with `make -C spark-tts_amd/csrc` "
            "(or __graft_entry__.build()). sparkmi has no CPU fallback.")
    # libsparkmi needs libamdhip64; PyTorch-ROCm ships its own copy and must own the process's
    # HIP runtime (streams and device memory are shared), so torch is loaded first and the
    # library then binds to the runtime that is already resident.
    import torch  # noqa: F401
    l = C.CDLL(str(path))
    for name, (res, args) in symbols.items():
        if os.environ.get("SPARKMI_LIB") and not hasattr(l, name):
            continue            # A/B runs against an older build (diagnostics only)
        fn = getattr(l, name)   # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    if l.smi_version() != ABI_VERSION:
        raise SparkMIError(f"{path.name} ABI version {l.smi_version()} != {ABI_VERSION}")
    return _Lib(l, path, is_diag)


_lib = None
_diag = None


def lib() -> _Lib:
    """Load (once) and return the product library; raises SparkMIError when it is not built."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, SYMBOLS, False)
    return _lib


def diag() -> _Lib:
    """Load (once) and return the diagnostics build (every product symbol + include/sparkmi_debug.h).  With SPARKMI_LIB set
    (an A/B variant build: `make variant` compiles with -DSMI_DIAG) that library is the diagnostics library."""
    global _diag
    if _diag is None:
        path = LIB_PATH if os.environ.get("SPARKMI_LIB") else DIAG_PATH
        _diag = _load(path, {**SYMBOLS, **DEBUG_SYMBOLS}, True)
    return _diag


def pick(use_diag: bool) -> _Lib:
    return diag() if use_diag else lib()


def check(rc: int, what: str = "", l: "_Lib | None" = None) -> None:
    (l or lib()).check(rc, what)


def require_gfx950() -> str:
    buf = C.create_string_buffer(128)
    check(lib().smi_device_check(buf, 128), "smi_device_check")
    return buf.value.decode()
