"""The LLM oracle against (a) the committed vectors that transformers' Qwen2ForCausalLM produced
in the build container (tests/golden/gen_golden.py) and (b) transformers itself, live, on a small
config (transformers is a third-party dependency present in the image, not reference code)."""
import os

import numpy as np
import pytest
import torch

from oracle.llm_ref import Qwen2Ref
from sparkmi import config as C, weights as W


@pytest.fixture(scope="module")
def tiny():
    cfg = C.tiny_llm()
    return cfg, W.SyntheticLLM(cfg)


def test_logits_match_golden(tiny, golden_dir):
    cfg, syn = tiny
    g = np.load(os.path.join(golden_dir, "llm_tiny.npz"))
    m = Qwen2Ref(cfg, syn)
    lg = m.forward(g["prompt"]).numpy()
    assert lg.shape == g["logits"].shape
    np.testing.assert_allclose(lg, g["logits"], rtol=0, atol=2e-5)


def test_greedy_matches_golden(tiny, golden_dir):
    cfg, syn = tiny
    g = np.load(os.path.join(golden_dir, "llm_tiny.npz"))
    toks = Qwen2Ref(cfg, syn).generate_greedy(g["prompt"], len(g["greedy"]))
    assert toks == g["greedy"].tolist()
    assert len(set(toks)) > 10  # the synthetic net must not collapse onto one token


def test_cache_path_equals_full_forward(tiny):
    cfg, syn = tiny
    ids = np.random.Generator(np.random.PCG64(5)).integers(0, cfg.vocab_size, size=19)
    full = Qwen2Ref(cfg, syn).forward(ids)
    m = Qwen2Ref(cfg, syn)
    parts = [m.forward(ids[:7]), m.forward(ids[7:8]), m.forward(ids[8:])]
    np.testing.assert_allclose(torch.cat(parts).numpy(), full.numpy(), rtol=0, atol=2e-5)


def test_eos_stops_generation(tiny):
    cfg, syn = tiny
    m = Qwen2Ref(cfg, syn)
    prompt = [5, 6, 7]
    free = m.generate_greedy(prompt, 12)
    stopped = m.generate_greedy(prompt, 12, eos_ids=[free[3]])
    assert stopped == free[: free.index(free[3]) + 1]


def test_bf16_kv_emulation_is_close(tiny):
    cfg, syn = tiny
    ids = np.arange(3, 20)
    a = Qwen2Ref(cfg, syn).forward(ids)
    b = Qwen2Ref(cfg, syn, kv_dtype="bf16").forward(ids)
    d = (a - b).abs().max().item()
    assert 0 < d < 0.1


def test_live_transformers_parity():
    """Same check as the golden one, but against transformers run here and now."""
    tr = pytest.importorskip("transformers")
    cfg = C.tiny_llm(vocab_size=517, layers=2)
    syn = W.SyntheticLLM(cfg, seed=3)
    hc = tr.Qwen2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                        intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads,
                        rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, tie_word_embeddings=True,
                        use_sliding_window=False, attn_implementation="eager")
    m = tr.Qwen2ForCausalLM(hc).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n != "lm_head.weight":
                p.copy_(torch.from_numpy(syn[n]))
    m.tie_weights()
    ids = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).integers(0, 517, size=(1, 15)))
    with torch.no_grad():
        ref = m(ids).logits[0]
        gen = m.generate(ids, attention_mask=torch.ones_like(ids), max_new_tokens=12, do_sample=False,
                         eos_token_id=None, pad_token_id=0)[0, 15:]
    mine = Qwen2Ref(cfg, syn)
    np.testing.assert_allclose(mine.forward(ids[0].numpy()).numpy(), ref.numpy(), rtol=0, atol=2e-5)
    assert Qwen2Ref(cfg, syn).generate_greedy(ids[0].numpy(), 12) == gen.tolist()


def test_full_size_golden_is_present(golden_dir):
    g = np.load(os.path.join(golden_dir, "llm_full.npz"))
    assert g["prompt"].shape == (128,) and g["greedy"].shape == (150,)
    assert g["last_top_ids"].shape == (64,)
