"""Weight-arena packing: checkpoint tensors -> the HBM layouts the HIP kernels stream.

LLM (``smi_llm_*``): every matrix is bf16 in 16-row x 32-column tiles stored
``[n_tile][k_tile][k8:4][n:16][8]`` -- one tile is 1 KiB, exactly one wave64 x 16-byte load and
one ``v_mfma_f32_16x16x32_bf16`` A operand.  Row orders are chosen so epilogues stay in
registers: q/k head rows as RoPE pairs (0,32,1,33,...), gate/up interleaved.  Section offsets
come from the library itself (``smi_llm_arena_section``), so there is one source of truth.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping

import os

import numpy as np

from . import _lib
from .config import LLMConfig
from .weights import bf16_bits_to_f32, f32_to_bf16_bits


class Bf16RoundingReport:
    """What packing a checkpoint into the bf16 arena changed.  north_star fixes the LLM arithmetic at bf16 weights
    with fp32 accumulation; a checkpoint saved in bf16 packs exactly (``max_rel == 0``), one saved in fp32 is ROUNDED
    here, and its logits then differ from the reference's fp32 CPU run by that rounding (``tests/test_llm_gpu.py::
    test_fp32_checkpoint_is_rounded_and_reported`` measures the effect on a synthetic fp32 model)."""

    def __init__(self):
        self.max_rel = 0.0          # max over matrices of max|w - bf16(w)| / max|w|
        self.inexact = 0            # matrices with at least one rounded element
        self.worst = ""

    def add(self, name: str, w: np.ndarray, bits: np.ndarray) -> None:
        scale = float(np.abs(w).max())
        if scale == 0.0:
            return
        err = float(np.abs(bf16_bits_to_f32(bits).reshape(w.shape) - w).max()) / scale
        if err > 0.0:
            self.inexact += 1
            if err > self.max_rel:
                self.max_rel, self.worst = err, name


def wd_row_parts() -> bool:
    """W_down tile order (include/sparkmi.h): row-part-major unless SPARKMI_WD_PLAIN=1 (a packer-side A/B switch: it only sets
    ``smi_llm_cfg.wd_plain``, which travels with the arena's config; the library never reads the variable)."""
    e = os.environ.get("SPARKMI_WD_PLAIN", "")
    return not (e and e != "0")


def pack_tiles(w: np.ndarray, report: "Bf16RoundingReport" = None, name: str = "", row_parts: bool = False) -> np.ndarray:
    """[N, K] fp32 -> uint16 bf16 bits in MFMA A-operand tile order; N padded to 16 with zeros.
    row_parts (W_down): inside a tile the 16-byte pieces are ordered [row part q:4][k8:4][row r:4] instead of [k8:4][n:16]
    (n = 4q + r), so the 4 (or 8) rows that one block of the row-split down_proj kernels loads are 256 (512) contiguous
    bytes -- whole 128-byte lines that no other block touches -- instead of four 64-byte half lines."""
    n, k = w.shape
    if k % 32:
        raise ValueError(f"K={k} must be a multiple of 32")
    npad = (n + 15) // 16 * 16
    bits = f32_to_bf16_bits(w).reshape(n, k)
    if report is not None:
        report.add(name, w, bits)
    if npad != n:
        bits = np.concatenate([bits, np.zeros((npad - n, k), np.uint16)], axis=0)
    if row_parts:
        t = bits.reshape(npad // 16, 4, 4, k // 32, 4, 8).transpose(0, 3, 1, 4, 2, 5)   # [nt][q][r][kt][k8][8] -> [nt][kt][q][k8][r][8]
    else:
        t = bits.reshape(npad // 16, 16, k // 32, 4, 8).transpose(0, 2, 3, 1, 4)
    return np.ascontiguousarray(t).reshape(-1)


def rope_pair_perm(n_heads: int, head_dim: int = 64) -> np.ndarray:
    """Row permutation placing RoPE partners (d, d + head_dim/2) in adjacent rows of each head."""
    half = head_dim // 2
    p = np.arange(head_dim)
    within = (p >> 1) + half * (p & 1)
    return (np.arange(n_heads)[:, None] * head_dim + within[None, :]).reshape(-1)


def o_proj_col_perm(n_heads: int, head_dim: int = 64) -> np.ndarray:
    """Column (input-feature) order of W_o in the arena: 32-column k tiles head-interleaved -- tile (half * n_heads + head)
    holds dims 32 * half .. 32 * half + 31 of `head` -- so that the o_proj kernel's k-tile -> wave map (tile mod n_heads)
    hands each wave one head: the per-head partial sums the fused one-row attention kernel produces itself."""
    if head_dim != 64:
        raise ValueError("head_dim must be 64")
    src = np.empty(n_heads * head_dim, dtype=np.int64)
    for half in range(2):
        for h in range(n_heads):
            t = half * n_heads + h
            src[t * 32: (t + 1) * 32] = h * head_dim + half * 32 + np.arange(32)
    return src


def rope_table(cfg: LLMConfig, max_positions: int) -> np.ndarray:
    """(cos, sin) [max_positions][head_dim/2] float32, computed with the same torch ops as
    transformers' Qwen2RotaryEmbedding (modeling_qwen2.py:81-102) so the bits match the oracle."""
    import torch
    d = cfg.head_dim
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    pos = torch.arange(max_positions, dtype=torch.float32)
    freqs = pos[:, None] * inv_freq[None, :]
    return torch.stack((freqs.cos(), freqs.sin()), dim=-1).numpy().astype(np.float32)


def llm_cfg_struct(cfg: LLMConfig, max_slots: int, max_positions: int, kv_dtype: str,
                   use_graph: bool, kv_page_tokens: int = 0, kv_pages: int = 0, weights_exact: bool = False) -> _lib.LLMCfg:
    """``weights_exact``: the verification mode of include/sparkmi.h -- fp32 matrices in the arena, exact fp32 GEMMs (for
    checkpoints saved in fp32, which the bf16 arena would round)."""
    return _lib.LLMCfg(
        weights_exact=int(bool(weights_exact)),
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_layers=cfg.num_hidden_layers,
        num_heads=cfg.num_attention_heads, num_kv_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
        intermediate_size=cfg.intermediate_size, max_slots=max_slots, max_positions=max_positions,
        kv_dtype={"bf16": 0, "f32": 1}[kv_dtype], use_graph=int(use_graph), rms_eps=cfg.rms_norm_eps,
        kv_page_tokens=int(kv_page_tokens), kv_pages=int(kv_pages), wd_plain=0 if wd_row_parts() else 1)


def pack_llm_arena(cfg: LLMConfig, weights: Mapping[str, np.ndarray], cs: _lib.LLMCfg,
                   strict_bf16: bool = False, report: "Bf16RoundingReport" = None) -> np.ndarray:
    """Host uint8 image of the whole LLM arena (caller uploads it to the GPU).  A checkpoint whose matrices are not
    bf16-representable (an fp32 save) is rounded: that is reported with a warning, or refused with ``strict_bf16``."""
    import warnings
    lib = _lib.lib()
    rep = report if report is not None else Bf16RoundingReport()
    total = lib.smi_llm_arena_bytes(C.byref(cs))
    if total == 0:
        raise _lib.SparkMIError("smi_llm_arena_bytes: config outside the kernel contract")
    arena = np.zeros(total, dtype=np.uint8)

    def put(section: int, layer: int, data: np.ndarray) -> None:
        off, nbytes = C.c_size_t(), C.c_size_t()
        _lib.check(lib.smi_llm_arena_section(C.byref(cs), section, layer, C.byref(off), C.byref(nbytes)),
                   "smi_llm_arena_section")
        raw = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        if raw.size != nbytes.value:
            raise ValueError(f"section {section} layer {layer}: {raw.size} bytes packed, {nbytes.value} expected")
        arena[off.value: off.value + raw.size] = raw

    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731

    def pack(w, report, name, row_parts=False):
        """bf16 MFMA tiles (default) or, in the exact-weights mode, the matrix as it is: fp32 [N padded to 16][K] row-major"""
        if not cs.weights_exact:
            return pack_tiles(w, report, name, row_parts=row_parts)
        n, k = w.shape
        if k % 32:
            raise ValueError(f"K={k} must be a multiple of 32")
        npad = (n + 15) // 16 * 16
        return np.ascontiguousarray(w if npad == n else np.concatenate([w, np.zeros((npad - n, k), np.float32)], axis=0))

    pq = rope_pair_perm(cfg.num_attention_heads, cfg.head_dim)
    pk = rope_pair_perm(cfg.num_key_value_heads, cfg.head_dim)
    po = o_proj_col_perm(cfg.num_attention_heads, cfg.head_dim)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        wq, wk, wv = (f32(weights[p + f"self_attn.{n}_proj.weight"]) for n in "qkv")
        bq, bk, bv = (f32(weights[p + f"self_attn.{n}_proj.bias"]) for n in "qkv")
        put(_lib.LLM_LN1, i, f32(weights[p + "input_layernorm.weight"]))
        put(_lib.LLM_WQKV, i, pack(np.concatenate([wq[pq], wk[pk], wv], axis=0), rep, p + "self_attn.qkv"))
        put(_lib.LLM_BQKV, i, np.concatenate([bq[pq], bk[pk], bv]))
        put(_lib.LLM_WO, i, pack(f32(weights[p + "self_attn.o_proj.weight"])[:, po], rep, p + "self_attn.o_proj"))
        put(_lib.LLM_LN2, i, f32(weights[p + "post_attention_layernorm.weight"]))
        g, u = f32(weights[p + "mlp.gate_proj.weight"]), f32(weights[p + "mlp.up_proj.weight"])
        gu = np.empty((2 * g.shape[0], g.shape[1]), np.float32)
        gu[0::2], gu[1::2] = g, u
        put(_lib.LLM_WGU, i, pack(gu, rep, p + "mlp.gate_up"))
        put(_lib.LLM_WD, i, pack(f32(weights[p + "mlp.down_proj.weight"]), rep, p + "mlp.down_proj", row_parts=not cs.wd_plain))
    put(_lib.LLM_FINAL_NORM, 0, f32(weights["model.norm.weight"]))
    head = "model.embed_tokens.weight" if cfg.tie_word_embeddings else "lm_head.weight"
    if not cfg.tie_word_embeddings:
        raise NotImplementedError("untied lm_head: the kernels gather embeddings from the lm_head tiles")
    put(_lib.LLM_LM_HEAD, 0, pack(f32(weights[head]), rep, head))
    put(_lib.LLM_ROPE, 0, rope_table(cfg, cs.max_positions))
    # how this arena was packed travels WITH it: smi_llm_create compares the tag with the config it is handed (an arena packed
    # under one SPARKMI_WD_PLAIN setting and re-used under another is an error, not wrong logits)
    tag = _lib.LLMArenaTag(magic=b"SMIARENA", abi_version=_lib.ABI_VERSION, wd_plain=cs.wd_plain, vocab_size=cs.vocab_size,
                           hidden_size=cs.hidden_size, num_layers=cs.num_layers, num_heads=cs.num_heads, num_kv_heads=cs.num_kv_heads,
                           intermediate_size=cs.intermediate_size, max_positions=cs.max_positions, weights_exact=cs.weights_exact)
    put(_lib.LLM_TAG, 0, np.frombuffer(bytes(tag), dtype=np.uint8))
    if rep.inexact:
        msg = (f"{rep.inexact} LLM matrices are not bf16-representable (an fp32 checkpoint?): rounded to bf16 for the "
               f"weight arena, max |w - bf16(w)| / max|w| = {rep.max_rel:.2e} at {rep.worst}; outputs will differ from an "
               f"fp32 CPU run of the same checkpoint by that rounding")
        if strict_bf16:
            raise ValueError(msg)
        warnings.warn(msg, stacklevel=2)
    return arena
