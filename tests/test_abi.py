"""libsparkmi.so loads and exports every symbol include/sparkmi.h declares (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from sparkmi import _lib, arena, config as C, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "sparkmi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(smi_[a-z0-9_]+)\s*\(", txt)))


def test_library_is_built():
    assert _lib.LIB_PATH.exists(), "run `make -C spark-tts_amd/csrc` (or __graft_entry__.build())"


def test_every_declared_symbol_is_exported_and_bound():
    l = _lib.lib()
    names = _header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(l, n), f"{n} declared in sparkmi.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes signature in sparkmi/_lib.py"
    assert l.smi_version() == 1


def test_llm_arena_layout_is_consistent():
    l = _lib.lib()
    cfg = C.tiny_llm()
    cs = arena.llm_cfg_struct(cfg, 4, 128, "bf16", True)
    total = l.smi_llm_arena_bytes(ctypes.byref(cs))
    assert total > 0
    spans = []
    for layer in range(cfg.num_hidden_layers):
        for s in range(_lib.LLM_LN1, _lib.LLM_WD + 1):
            off, nb = ctypes.c_size_t(), ctypes.c_size_t()
            assert l.smi_llm_arena_section(ctypes.byref(cs), s, layer, ctypes.byref(off), ctypes.byref(nb)) == 0
            spans.append((off.value, nb.value))
    for s in (_lib.LLM_FINAL_NORM, _lib.LLM_LM_HEAD, _lib.LLM_ROPE):
        off, nb = ctypes.c_size_t(), ctypes.c_size_t()
        assert l.smi_llm_arena_section(ctypes.byref(cs), s, 0, ctypes.byref(off), ctypes.byref(nb)) == 0
        spans.append((off.value, nb.value))
    spans.sort()
    for (o1, n1), (o2, _) in zip(spans, spans[1:]):
        assert o1 + n1 <= o2 and o1 % 256 == 0
    assert spans[-1][0] + spans[-1][1] <= total
    # bad arguments are reported, not crashed on
    assert l.smi_llm_arena_section(ctypes.byref(cs), 99, 0, None, None) == -1
    assert b"section" in l.smi_last_error()
    bad = arena.llm_cfg_struct(C.LLMConfig(hidden_size=100, num_attention_heads=2), 1, 64, "bf16", True)
    assert l.smi_llm_arena_bytes(ctypes.byref(bad)) == 0


def test_pack_tiles_layout():
    rng = np.random.default_rng(0)
    w = W.round_bf16(rng.standard_normal((20, 64)).astype(np.float32))
    t = arena.pack_tiles(w).reshape(2, 2, 4, 16, 8)   # [n_tile][k_tile][k8][n][j]
    for (n, k) in [(0, 0), (3, 9), (15, 31), (16, 32), (19, 63), (7, 40)]:
        bits = t[n // 16, k // 32, (k % 32) // 8, n % 16, k % 8]
        assert W.bf16_bits_to_f32(np.array([bits]))[0] == w[n, k]
    assert not t[1, :, :, 4:, :].any()   # rows 20..31 are zero padding


def test_rope_pair_perm():
    p = arena.rope_pair_perm(2, 64)
    assert p[:6].tolist() == [0, 32, 1, 33, 2, 34] and p[64:68].tolist() == [64, 96, 65, 97]
    assert sorted(p.tolist()) == list(range(128))


def test_arena_packs_full_tiny_model():
    cfg = C.tiny_llm()
    cs = arena.llm_cfg_struct(cfg, 2, 96, "bf16", False)
    a = arena.pack_llm_arena(cfg, W.SyntheticLLM(cfg), cs)
    assert a.dtype == np.uint8 and a.size == _lib.lib().smi_llm_arena_bytes(ctypes.byref(cs))


def test_no_cpu_fallback():
    import torch
    from sparkmi.llm import SparkLLM
    cfg = C.tiny_llm()
    with pytest.raises(_lib.SparkMIError):
        SparkLLM(cfg, W.SyntheticLLM(cfg), device="cpu")
    if not torch.cuda.is_available():
        with pytest.raises(_lib.SparkMIError):
            _lib.require_gfx950()
