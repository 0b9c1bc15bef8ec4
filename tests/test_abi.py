"""libsparkmi.so loads and exports every symbol include/sparkmi.h declares -- and nothing else; libsparkmi_diag.so exports those
plus include/sparkmi_debug.h's (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from sparkmi import _lib, arena, config as C, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(name="sparkmi.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
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
    assert l.smi_version() == _lib.ABI_VERSION == 4


def test_diagnostics_live_in_their_own_library():
    """include/sparkmi_debug.h (timing probes, stamps, scratch dumps, the one-row engine) is exported by libsparkmi_diag.so only."""
    dbg = _header_symbols("sparkmi_debug.h")
    assert "smi_llm_time_kernel" in dbg and "smi_llm_debug_read" in dbg and not set(dbg) & set(_header_symbols())
    d = _lib.diag()
    for n in dbg + _header_symbols():
        assert hasattr(d, n), f"{n} missing from libsparkmi_diag.so"
    for n in dbg:
        assert n in _lib.DEBUG_SYMBOLS, f"{n} has no ctypes signature in sparkmi/_lib.py"
        assert not hasattr(_lib.lib()._cdll, n), f"{n} leaked into the product library"
    assert d.smi_version() == _lib.ABI_VERSION and d.is_diag and not _lib.lib().is_diag


def test_product_library_reads_no_environment():
    """The product build compiles smi_env() to nullptr: no getenv in its import table (the diagnostics build has it)."""
    import subprocess
    und = lambda p: subprocess.run(["nm", "-D", "--undefined-only", str(p)], capture_output=True, text=True, check=True).stdout  # noqa: E731
    assert "getenv" not in und(_lib.LIB_PATH)
    assert "getenv" in und(_lib.DIAG_PATH)


def test_product_library_exports_only_the_declared_abi():
    """Diagnostics (micro-benchmarks) live in libsparkmi_diag.so; the product library's dynamic symbol table holds the
    header's entry points and nothing else of ours."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    ours = sorted(set(re.findall(r" T (smi_[a-z0-9_]+)$", out, flags=re.M)))
    assert ours == _header_symbols(), set(ours) ^ set(_header_symbols())


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
    for s in (_lib.LLM_FINAL_NORM, _lib.LLM_LM_HEAD, _lib.LLM_ROPE, _lib.LLM_TAG):
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


def test_pack_tiles_row_part_order_is_a_permutation_of_the_tile():
    """W_down's tile order (include/sparkmi.h): piece (k8, n = 4q + r) of the plain order sits at q*16 + k8*4 + r."""
    rng = np.random.default_rng(1)
    w = W.round_bf16(rng.standard_normal((32, 96)).astype(np.float32))
    plain = arena.pack_tiles(w).reshape(2, 3, 64, 8)            # [n_tile][k_tile][piece][8]
    parts = arena.pack_tiles(w, row_parts=True).reshape(2, 3, 64, 8)
    for lane in range(64):
        k8, n = lane >> 4, lane & 15
        np.testing.assert_array_equal(parts[:, :, (n >> 2) * 16 + k8 * 4 + (n & 3)], plain[:, :, lane])
        assert (n >> 2) * 16 + k8 * 4 + (n & 3) == ((lane & 12) << 2) + ((lane >> 4) << 2) + (lane & 3)   # the kernels' smi_wlane


def test_rope_pair_perm():
    p = arena.rope_pair_perm(2, 64)
    assert p[:6].tolist() == [0, 32, 1, 33, 2, 34] and p[64:68].tolist() == [64, 96, 65, 97]
    assert sorted(p.tolist()) == list(range(128))


def test_arena_packs_full_tiny_model():
    cfg = C.tiny_llm()
    cs = arena.llm_cfg_struct(cfg, 2, 96, "bf16", False)
    a = arena.pack_llm_arena(cfg, W.SyntheticLLM(cfg), cs)
    assert a.dtype == np.uint8 and a.size == _lib.lib().smi_llm_arena_bytes(ctypes.byref(cs))
    # the arena says how it was packed (smi_llm_arena_tag): smi_llm_create checks it against the config
    off, nb = ctypes.c_size_t(), ctypes.c_size_t()
    assert _lib.lib().smi_llm_arena_section(ctypes.byref(cs), _lib.LLM_TAG, 0, ctypes.byref(off), ctypes.byref(nb)) == 0
    tag = _lib.LLMArenaTag.from_buffer_copy(a[off.value: off.value + nb.value].tobytes())
    assert nb.value == 256 and tag.magic == b"SMIARENA" and tag.abi_version == _lib.ABI_VERSION
    assert (tag.wd_plain, tag.hidden_size, tag.num_layers, tag.max_positions) == (cs.wd_plain, cfg.hidden_size, cfg.num_hidden_layers, 96)


def test_no_cpu_fallback():
    import torch
    from sparkmi.llm import SparkLLM
    cfg = C.tiny_llm()
    with pytest.raises(_lib.SparkMIError):
        SparkLLM(cfg, W.SyntheticLLM(cfg), device="cpu")
    if not torch.cuda.is_available():
        with pytest.raises(_lib.SparkMIError):
            _lib.require_gfx950()


def test_voc_arena_layout_and_packing():
    import ctypes as Cc
    from sparkmi import bicodec
    l = _lib.lib()
    for cfg in (C.tiny_bicodec(), C.spark_0p5b_bicodec()):
        cs = bicodec.voc_cfg_struct(cfg, 2, 64)
        n = l.smi_voc_arena_count(Cc.byref(cs))
        assert n > 50
        names = set()
        name = Cc.create_string_buffer(8192)
        end = 0
        for i in range(n):
            off, nb = Cc.c_size_t(), Cc.c_size_t()
            info = (Cc.c_int32 * 6)()
            assert l.smi_voc_arena_entry(Cc.byref(cs), i, name, 8192, Cc.byref(off), Cc.byref(nb), info) == 0
            assert off.value >= end and off.value % 256 == 0
            end = off.value + nb.value
            names.add(name.value.decode())
        assert end <= l.smi_voc_arena_bytes(Cc.byref(cs))
        assert "decoder.model.0.weight" in names and "quantizer.codebook.weight" in names
    cfg = C.tiny_bicodec()
    sd = W.fold_weight_norm(W.bicodec_detok_state(cfg))
    cs = bicodec.voc_cfg_struct(cfg, 1, 32)
    a = bicodec.pack_voc_arena(cfg, sd, cs)
    assert a.dtype == np.float32 and a.size * 4 == l.smi_voc_arena_bytes(Cc.byref(cs))
    # every parameter detokenize touches was consumed by the arena
    used = set()
    name = Cc.create_string_buffer(8192)
    for i in range(l.smi_voc_arena_count(Cc.byref(cs))):
        l.smi_voc_arena_entry(Cc.byref(cs), i, name, 8192, None, None, None)
        k = name.value.decode()
        used.update(k[4:].split("|") if k.startswith("cat:") else [k])
    assert used == set(sd.keys())


def test_pack_conv_layouts():
    from sparkmi.bicodec import pack_conv, conv_phases, PACK_CONV, PACK_CONVT
    rng = np.random.default_rng(1)
    w = rng.standard_normal((40, 12, 7)).astype(np.float32)       # Conv1d (Cout, Cin, K)
    p = pack_conv(w, PACK_CONV, 1, 0).reshape(2, 7, 2, 64, 4)     # [ct][tap][g][lane][j]
    for (co, ci, k) in [(0, 0, 0), (33, 11, 6), (39, 5, 3), (31, 8, 2)]:
        lane = (ci % 8 % 2) * 32 + co % 32
        assert p[co // 32, k, ci // 8, lane, (ci % 8) // 2] == w[co, ci, k]
    assert not p[1, :, :, 8:32, :].any() and not p[:, :, 1, :, 2:].any()   # zero padding rows / channels
    # ConvTranspose1d k=11, s=5, pad=3: phases have 3 or 2 taps
    assert [len(t) for t in conv_phases(11, 5, 3)] == [2, 2, 3, 2, 2]
    wt = rng.standard_normal((8, 32, 11)).astype(np.float32)      # (Cin, Cout, K)
    pt = pack_conv(wt, PACK_CONVT, 5, 3)
    assert pt.size == 11 * 1 * 1 * 256
    taps0 = conv_phases(11, 5, 3)[0]
    first = pt[: len(taps0) * 256].reshape(len(taps0), 64, 4)
    assert first[1, 32 + 7, 2] == wt[5, 7, taps0[1]]              # lane 39: co 7, half 1; j=2 -> ci 5
