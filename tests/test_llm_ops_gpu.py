"""Op-level GPU tests of the LLM half on transformers' OWN classes (tests/golden/llm_ops.npz, generator
tests/golden/gen_golden_llm_ops.py): Qwen2RMSNorm + q/k/v_proj + apply_rotary_pos_emb, eager_attention_forward, o_proj + residual,
the inside of Qwen2MLP, Qwen2MLP + residual -- one decoder layer taken apart.  The HIP side runs that layer's kernels ONE STAGE AT
A TIME through ``smi_llm_debug_layer`` (libsparkmi_diag.so), i.e. through ``launch_one``: the launch builders, per-row-count kernel
choices and prologue / epilogue fusions of a real decode step (as ``smi_voc_block_run`` does for the vocoder's blocks), so a
regression in a fused kernel shows up at its own stage instead of as a logit difference 24 layers later.

Bars: f32 KV cache 1e-5 relative to each stage's scale (exact products, fp32 sums: summation-order noise); bf16 KV cache: stage 0
keys / values to bf16 rounding, later stages 2e-2 (the cached keys of the fixture are not bf16-representable)."""
import os

import numpy as np
import pytest

from sparkmi import config as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "llm_ops.npz"))


def _weights(g, cfg):
    """The fixture's layer as a one-layer checkpoint under HF names (embeddings / final norm are not touched by these tests)."""
    rng = np.random.default_rng(0)
    w = {"model.embed_tokens.weight": (0.02 * rng.standard_normal((cfg.vocab_size, cfg.hidden_size))).astype(np.float32),
         "model.norm.weight": np.ones(cfg.hidden_size, np.float32),
         "model.layers.0.input_layernorm.weight": g["ln1"], "model.layers.0.post_attention_layernorm.weight": g["ln2"]}
    from sparkmi.weights import round_bf16
    w["model.embed_tokens.weight"] = round_bf16(w["model.embed_tokens.weight"])
    for k in g.files:
        if k.startswith("attn/"):
            w["model.layers.0.self_attn." + k[5:]] = g[k]
        elif k.startswith("mlp/"):
            w["model.layers.0.mlp." + k[4:]] = g[k]
    return w


def _engine(g, kv, slots, monkeypatch=None, no_fuse=False):
    from sparkmi.llm import SparkLLM
    cfg = C.LLMConfig(vocab_size=64, hidden_size=256, num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=2,
                      intermediate_size=608, rope_theta=1000000.0, rms_norm_eps=1e-6)
    if no_fuse:
        monkeypatch.setenv("SPARKMI_NO_FUSE_O", "1")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")        # (the q/k/v biases are fp32; the matrices are bf16-exact)
        llm = SparkLLM(cfg, _weights(g, cfg), "cuda:0", max_slots=slots, max_positions=512, kv_dtype=kv, diag=True)
    return cfg, llm


def _load_caches(g, llm, slot_of):
    """slot_of: engine slot -> fixture context (0 / 1 / 2)"""
    for slot, c in slot_of.items():
        llm.debug_set_kv(0, slot, g[f"kcache{c}"].transpose(1, 0, 2), g[f"vcache{c}"].transpose(1, 0, 2))


def _close(got, want, rel, what):
    scale = float(np.abs(want).max())
    err = float(np.abs(got - want).max())
    assert err <= rel * scale, f"{what}: max |diff| {err:.3e} against scale {scale:.3f} (bar {rel:g})"


STAGES = [(0, None), (1, "attn_out"), (2, "h_mid"), (3, "act"), (4, "h_out")]


def _check_rows(g, llm, rows, pick, rel, tag, stages=STAGES):
    """rows: engine (slot, pos) pairs; pick: for each, the fixture row it stands for"""
    x = g["x"][pick]
    for stage, key in stages:
        out = llm.debug_layer(0, rows, x, stage)
        if stage == 0:
            _close(out["q"], g["q_rot"][pick], rel, f"{tag} stage 0 q (RMSNorm + q_proj + bias + RoPE)")
            _close(out["k"], g["k_rot"][pick], max(rel, 4e-3 if llm._cs.kv_dtype == 0 else 0), f"{tag} stage 0 k (appended cache row)")
            _close(out["v"], g["v"][pick], max(rel, 4e-3 if llm._cs.kv_dtype == 0 else 0), f"{tag} stage 0 v")
        else:
            name = {1: "attn", 2: "h", 3: "act", 4: "h"}[stage]
            _close(out[name], g[key][pick], rel, f"{tag} stage {stage} ({key})")


@pytest.mark.parametrize("kv,rel", [("f32", 1e-5), ("bf16", 2e-2)])
def test_mixed_rows_of_one_layer_stage_by_stage(fx, kv, rel):
    """The fixture's five rows in one call: three KV slots with contexts of 9 / 130 / 300 tokens, two consecutive rows of one
    slot among them (a prefill chunk's shape) -> the general (row descriptor) attention kernel, the 2..8-row GEMV kernels."""
    cfg, llm = _engine(fx, kv, 3)
    _load_caches(fx, llm, {0: 0, 1: 1, 2: 2})
    _check_rows(fx, llm, fx["rows"], np.arange(5), rel, f"5 rows, {kv} KV")


@pytest.mark.parametrize("fused", [True, False])
def test_single_row_kernels(fx, monkeypatch, fused):
    """One live row in slot 0 (fixture row 0: position 9 behind 9 cached keys): the lean one-row GEMVs, and the fused attention +
    o_proj kernel with gate_up's head-sum prologue (stage 1 is inside that kernel: checked on the un-fused build of the path)."""
    cfg, llm = _engine(fx, "f32", 1, monkeypatch, no_fuse=not fused)
    _load_caches(fx, llm, {0: 0})
    stages = [s for s in STAGES if not (fused and s[0] == 1)]
    _check_rows(fx, llm, fx["rows"][:1], np.array([0]), 1e-5, f"one row, fused={fused}", stages)


@pytest.mark.parametrize("nrows", [2, 5, 8, 16, 20, 32, 64])
def test_batched_decode_rows(fx, nrows, monkeypatch):
    """`nrows` live sequences, row m in slot m (the slot == row decode kernels; 7..64 rows: the chain-split down_proj; 17+: the
    two-m-tile gate_up, 16-row block rows of QKV / o_proj): each slot holds one of the fixture's three contexts, each row is the
    fixture row of that context -- and must come out as it does alone.  At 2..8 rows the diagnostics build's opt-in form --
    attention with the fused o_proj and a last-arriver head sum (SPARKMI_FUSE2_ROWS=8; measured slower and off by default) -- is
    held to the same vectors and must give the residual rows of the separate o_proj kernel bit for bit."""
    if nrows <= 8:
        cfg, unfused = _engine(fx, "f32", nrows, monkeypatch=None)
        monkeypatch.setenv("SPARKMI_FUSE2_ROWS", "8")   # the last-arriver form is an opt-in of the diagnostics build (measured slower: DESIGN 3.9)
        _, plain = _engine(fx, "f32", nrows)
        monkeypatch.delenv("SPARKMI_FUSE2_ROWS")
        ctx_row = {0: 0, 1: 1, 2: 3}
        which = [m % 3 for m in range(nrows)]
        pick = np.array([ctx_row[c] for c in which])
        rows = np.stack([np.arange(nrows), fx["rows"][pick, 1]], axis=1)
        for e in (plain, unfused):
            _load_caches(fx, e, {m: which[m] for m in range(nrows)})
        _check_rows(fx, unfused, rows, pick, 1e-5, f"{nrows} decode rows, separate o_proj")
        _check_rows(fx, plain, rows, pick, 1e-5, f"{nrows} decode rows, fused o_proj", [s for s in STAGES if s[0] != 1])
        for stage in (2, 4):
            a = plain.debug_layer(0, rows, fx["x"][pick], stage)["h"]
            b = unfused.debug_layer(0, rows, fx["x"][pick], stage)["h"]
            assert np.array_equal(a, b), f"{nrows} rows, stage {stage}: fused and separate o_proj differ in bits"
        return
    cfg, llm = _engine(fx, "f32", nrows)
    ctx_row = {0: 0, 1: 1, 2: 3}                       # fixture context -> the fixture row that decodes right behind it
    which = [m % 3 for m in range(nrows)]
    _load_caches(fx, llm, {m: which[m] for m in range(nrows)})
    pick = np.array([ctx_row[c] for c in which])
    rows = np.stack([np.arange(nrows), fx["rows"][pick, 1]], axis=1)
    _check_rows(fx, llm, rows, pick, 1e-5, f"{nrows} decode rows")
