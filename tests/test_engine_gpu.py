"""The one-row decode engine (csrc/smi_eng.h: all layers of a decode step in ONE persistent launch, hand-offs inside the
launch) against the four-launches-per-layer path it replaces and against the oracle.

The engine restates the launch path's arithmetic product for product and chain for chain, so the bar is equality of
bits: the residual row after every step and every token.  The launch path itself is pinned to the oracle / transformers
golden vectors by tests/test_llm_gpu.py and tests/test_fullsize_gpu.py; the direct engine-vs-oracle comparisons are here too."""
import os

import numpy as np
import pytest

from oracle.llm_ref import Qwen2Ref
from sparkmi import config as C, weights as W

pytestmark = pytest.mark.gpu


def _llm(cfg, syn, **kw):
    from sparkmi.llm import SparkLLM
    return SparkLLM(cfg, syn, device="cuda:0", diag=True, **kw)   # the engine and the residual-row dump: include/sparkmi_debug.h


@pytest.fixture(scope="module")
def tiny():
    cfg = C.tiny_llm()
    return cfg, W.SyntheticLLM(cfg)


def _run(llm, prompt, steps, engine):
    """Prefill + `steps` single decode steps; the residual row after every step and the tokens."""
    llm.set_engine(engine)
    llm.prefill([prompt])
    rows = [llm.debug_hidden()]
    for _ in range(steps):
        llm.decode(1)
        rows.append(llm.debug_hidden())
    return np.stack(rows), llm.tokens(steps + 1)[0]


@pytest.mark.parametrize("graph", [False, True])
def test_engine_equals_launch_path_bit_for_bit_tiny(tiny, graph):
    cfg, syn = tiny
    llm = _llm(cfg, syn, max_positions=320, use_graph=graph)
    assert not llm.engine_info()["enabled"]          # opt-in: the launch path is the default
    llm.set_engine(True)
    info = llm.engine_info()
    assert info["built"] and info["enabled"], f"engine not built: {info['why']}"
    prompt = np.random.Generator(np.random.PCG64(11)).integers(0, cfg.vocab_size, size=37).tolist()
    h_eng, t_eng = _run(llm, prompt, 270, True)      # crosses the 256-key chunk boundary of the attention
    h_ref, t_ref = _run(llm, prompt, 270, False)
    assert t_eng == t_ref
    assert (h_eng.view(np.uint32) == h_ref.view(np.uint32)).all(), "residual rows differ in bits"


def test_engine_tokens_match_oracle_tiny(tiny):
    cfg, syn = tiny
    llm = _llm(cfg, syn, max_positions=128)
    llm.set_engine(True)
    assert llm.engine_info()["enabled"]
    prompt = np.random.Generator(np.random.PCG64(12)).integers(0, cfg.vocab_size, size=21)
    got = llm.generate_ids([prompt.tolist()], 40)[0]
    ref = Qwen2Ref(cfg, syn, kv_dtype="bf16").generate_greedy(prompt, 40)
    assert got == ref


def test_engine_is_deterministic_and_survives_handle_reuse(tiny):
    cfg, syn = tiny
    llm = _llm(cfg, syn, max_positions=128)
    llm.set_engine(True)
    rng = np.random.Generator(np.random.PCG64(13))
    p1 = rng.integers(0, cfg.vocab_size, size=30).tolist()
    p2 = rng.integers(0, cfg.vocab_size, size=9).tolist()
    a = llm.generate_ids([p1], 50)[0]
    b = llm.generate_ids([p2], 3)[0]       # a short generation in between: the next one starts at the same step numbers
    c = llm.generate_ids([p1], 50)[0]
    assert a == c and len(b) == 3


def test_engine_off_for_batches_and_f32_kv(tiny):
    cfg, syn = tiny
    llm = _llm(cfg, syn, max_positions=128, kv_dtype="f32")
    llm.set_engine(True)                           # does not apply to an f32 cache: stays off, says why
    info = llm.engine_info()
    assert not info["built"] and not info["enabled"] and "f32" in info["why"]
    llm2 = _llm(cfg, syn, max_positions=128, max_slots=4)
    llm2.set_engine(True)
    rng = np.random.Generator(np.random.PCG64(14))
    prompts = [rng.integers(0, cfg.vocab_size, size=12 + i).tolist() for i in range(3)]
    got = llm2.generate_ids(prompts, 20)           # three rows: the launch path
    solo = [llm2.generate_ids([p], 20)[0] for p in prompts]   # one row: the engine
    assert got == solo, "a sequence's tokens must not depend on the path its batch size selects"


@pytest.mark.timeout(900)
def test_engine_full_size_equals_launch_path(golden_dir, full_llm):
    """BASELINE configs[1] as benched: 0.5B shape, bf16 KV, hipGraph.  The residual row after every one of 160 steps (they
    cross the 256-key boundary of the attention: context 128 -> 288) and the tokens equal the launch path's bit for bit;
    test_fullsize_gpu.py::test_config2_free_running_bf16_kv_graph_to_waveform pins the same run to the oracle."""
    cfg, syn, arena = full_llm
    g = np.load(os.path.join(golden_dir, "llm_full.npz"))
    from conftest import FULL_MAX_POS
    llm = _llm(cfg, None, max_positions=FULL_MAX_POS, arena=arena)
    llm.set_engine(True)
    info = llm.engine_info()
    assert info["enabled"], info["why"]
    prompt = g["prompt"].tolist()
    h_eng, t_eng = _run(llm, prompt, 160, True)
    h_ref, t_ref = _run(llm, prompt, 160, False)
    assert t_eng == t_ref
    assert (h_eng.view(np.uint32) == h_ref.view(np.uint32)).all()
    llm.set_engine(True)
    a = llm.generate_ids([prompt], 150)[0]      # whole utterance on the replayed graph, twice
    b = llm.generate_ids([prompt], 150)[0]
    assert a == b == t_eng[:150]
