"""Per-op pins of the oracle: each function of oracle/bicodec_ref.py and oracle/llm_ref.py against a vector the
reference's own layer class (vocoder side) or transformers' Qwen2 class (LLM side) produced in the build container
(tests/golden/gen_golden.py: gen_ops_layers -> ops_layers.npz).  They localise a mismatch that the end-to-end vectors
would only report as 'the waveform differs'."""
import os
import types

import numpy as np
import pytest
import torch

from oracle.bicodec_ref import BiCodecDetokRef
from oracle.llm_ref import Qwen2Ref


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "ops_layers.npz"))


def _sub(g, prefix):
    """The layer's parameters under the oracle's naming: '<prefix>/<reference state_dict key>' -> 'L.<key>'."""
    return {"L." + k[len(prefix) + 1:]: g[k] for k in g.files if k.startswith(prefix + "/")}


def _voc(sd):
    ref = BiCodecDetokRef.__new__(BiCodecDetokRef)
    ref.sd, ref.cfg = sd, None
    return ref


@pytest.mark.parametrize("dil", [1, 3, 9])
def test_residual_unit(g, dil):
    """sparktts/modules/blocks/layers.py:51-67"""
    ref = _voc(_sub(g, f"resunit{dil}"))
    y = ref._res_unit(torch.from_numpy(g[f"resunit{dil}.x"]), "L.block", dil)
    np.testing.assert_allclose(y.numpy(), g[f"resunit{dil}.y"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("k,s", [(16, 8), (11, 5), (8, 4), (4, 2)])
def test_decoder_block(g, k, s):
    """sparktts/modules/encoder_decoder/wave_generator.py:29-53 -- every (kernel, stride) of the 0.5B decoder, including
    k = 11 / s = 5 whose output phases have two or three taps"""
    ref = _voc(_sub(g, f"decblock{k}_{s}"))
    x = torch.from_numpy(g[f"decblock{k}_{s}.x"])
    y = ref._decoder_block(x, "L.block", k, s)
    assert y.shape[-1] == x.shape[-1] * s
    np.testing.assert_allclose(y.numpy(), g[f"decblock{k}_{s}.y"], rtol=0, atol=5e-6)


@pytest.mark.parametrize("tag", ["ln", "adaln"])
def test_convnext_block(g, tag):
    """sparktts/modules/blocks/vocos.py:26-110 (LayerNorm and AdaLayerNorm variants)"""
    ref = _voc(_sub(g, f"convnext_{tag}"))
    cond = torch.from_numpy(g[f"convnext_{tag}.cond"]) if tag == "adaln" else None
    y = ref._convnext(torch.from_numpy(g[f"convnext_{tag}.x"]), "L", cond)
    np.testing.assert_allclose(y.numpy(), g[f"convnext_{tag}.y"], rtol=0, atol=2e-6)


def _llm():
    ref = Qwen2Ref.__new__(Qwen2Ref)
    ref.cfg = types.SimpleNamespace(rms_norm_eps=1e-6, num_attention_heads=4, num_key_value_heads=2, head_dim=32, rope_theta=1000000.0)
    ref.inv_freq = 1.0 / (ref.cfg.rope_theta ** (torch.arange(0, 32, 2, dtype=torch.float32) / 32))
    return ref


def test_rmsnorm(g):
    """transformers modeling_qwen2.py Qwen2RMSNorm"""
    y = _llm().rmsnorm(torch.from_numpy(g["rmsnorm.x"]), torch.from_numpy(g["rmsnorm.w"]))
    np.testing.assert_array_equal(y.numpy(), g["rmsnorm.y"])


def test_rope(g):
    """Qwen2RotaryEmbedding + apply_rotary_pos_emb (theta 1e6, absolute positions, half rotation)"""
    ref = _llm()
    pos = torch.from_numpy(g["rope.pos"])
    q = ref.rope(torch.from_numpy(g["rope.q"]).transpose(0, 1), pos)      # oracle layout (S, H, D)
    k = ref.rope(torch.from_numpy(g["rope.k"]).transpose(0, 1), pos)
    np.testing.assert_allclose(q.transpose(0, 1).numpy(), g["rope.q_rot"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(k.transpose(0, 1).numpy(), g["rope.k_rot"], rtol=0, atol=1e-6)


def test_gqa_attention_step(g):
    """eager_attention_forward with repeat_kv: one query position against ten cached keys, 2 query heads per KV head"""
    ref = _llm()
    q = torch.from_numpy(g["attn.q"]).transpose(0, 1)                     # (S=1, Hq, D)
    k = torch.from_numpy(g["attn.k"]).transpose(0, 1)                     # (T, Hkv, D)
    v = torch.from_numpy(g["attn.v"]).transpose(0, 1)
    y = ref.attention(q, k, v, torch.tensor([9]))
    want = g["attn.y"].reshape(1, -1)                                     # (S, Hq * D)
    np.testing.assert_allclose(y.numpy(), want, rtol=0, atol=1e-6)


def test_swiglu_mlp(g):
    """Qwen2MLP: down(silu(gate x) * up x)"""
    import torch.nn.functional as F
    w = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("mlp/")}
    x = torch.from_numpy(g["mlp.x"])
    y = F.linear(F.silu(F.linear(x, w["gate_proj.weight"])) * F.linear(x, w["up_proj.weight"]), w["down_proj.weight"])
    np.testing.assert_allclose(y.numpy(), g["mlp.y"], rtol=0, atol=1e-6)


def test_vocoder_batch_of_four_matches_reference(golden_dir):
    """The reference's own modules on a B = 4 batch (tiny config, equal lengths): the oracle's batch handling."""
    from sparkmi import config as C, weights as W
    g = np.load(os.path.join(golden_dir, "extra.npz"))
    cfg = C.tiny_bicodec()
    ref = BiCodecDetokRef(cfg, W.fold_weight_norm(W.bicodec_detok_state(cfg)))
    wav = ref.detokenize(torch.from_numpy(g["voc4_semantic"]), torch.from_numpy(g["voc4_global"]))
    np.testing.assert_allclose(wav.numpy(), g["voc4_wav"], rtol=0, atol=2e-6)


def test_llm_hidden_states_per_layer_match_transformers(golden_dir):
    """Per-layer hidden-state checksums of the tiny LLM from transformers (output_hidden_states): localises a mismatch to a layer."""
    from sparkmi import config as C, weights as W
    g = np.load(os.path.join(golden_dir, "extra.npz"))
    cfg = C.tiny_llm()
    ref = Qwen2Ref(cfg, W.SyntheticLLM(cfg))
    logits, hid = ref.forward(g["llm_prompt"], return_hidden=True)
    emb = ref.embed[torch.as_tensor(g["llm_prompt"])]
    states = [emb] + hid
    assert len(states) == len(g["llm_hidden_sum"])
    for i, h in enumerate(states):
        if i == len(states) - 1:        # transformers reports the LAST state after the final RMSNorm
            h = ref.rmsnorm(h, ref.final_norm)
        assert abs(float(h.double().abs().sum()) - g["llm_hidden_abs"][i]) < 1e-5 * g["llm_hidden_abs"][i], f"layer {i}"
        np.testing.assert_allclose(h[-1, :8].numpy(), g["llm_hidden_first"][i], rtol=0, atol=2e-5, err_msg=f"layer {i}")


def test_decoder_layer_taken_apart_at_head_dim_64(golden_dir):
    """tests/golden/llm_ops.npz (transformers' Qwen2RMSNorm, q/k/v/o_proj, apply_rotary_pos_emb, eager_attention_forward, Qwen2MLP at
    the head_dim the HIP kernels are built for): the oracle's pieces, stage by stage -- the same vectors tests/test_llm_ops_gpu.py
    holds the kernels to."""
    import torch.nn.functional as F
    g = np.load(os.path.join(golden_dir, "llm_ops.npz"))
    ref = Qwen2Ref.__new__(Qwen2Ref)
    ref.cfg = types.SimpleNamespace(rms_norm_eps=1e-6, num_attention_heads=4, num_key_value_heads=2, head_dim=64, rope_theta=1000000.0)
    ref.inv_freq = 1.0 / (ref.cfg.rope_theta ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    rows, x = g["rows"], t("x")
    M = len(rows)
    pos = torch.from_numpy(rows[:, 1].astype(np.int64))
    xn = ref.rmsnorm(x, t("ln1"))
    q = ref.rope(F.linear(xn, t("attn/q_proj.weight"), t("attn/q_proj.bias")).view(M, 4, 64), pos)
    k = ref.rope(F.linear(xn, t("attn/k_proj.weight"), t("attn/k_proj.bias")).view(M, 2, 64), pos)
    v = F.linear(xn, t("attn/v_proj.weight"), t("attn/v_proj.bias")).view(M, 2, 64)
    np.testing.assert_allclose(q.numpy(), g["q_rot"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(k.numpy(), g["k_rot"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(v.numpy(), g["v"], rtol=0, atol=3e-6)
    ao = torch.zeros(M, 256)
    for i, (s, p) in enumerate(rows):
        own = [j for j, (s2, p2) in enumerate(rows) if s2 == s and p2 <= p]
        K = torch.cat([t(f"kcache{s}").transpose(0, 1)] + [k[j: j + 1] for j in own])     # (T, Hkv, D)
        V = torch.cat([t(f"vcache{s}").transpose(0, 1)] + [v[j: j + 1] for j in own])
        assert K.shape[0] == p + 1
        ao[i] = ref.attention(q[i: i + 1], K, V, torch.tensor([int(p)]))[0]
    np.testing.assert_allclose(ao.numpy(), g["attn_out"], rtol=0, atol=2e-6)
    h_mid = x + F.linear(ao, t("attn/o_proj.weight"))
    np.testing.assert_allclose(h_mid.numpy(), g["h_mid"], rtol=0, atol=5e-6)
    xn2 = ref.rmsnorm(h_mid, t("ln2"))
    act = F.silu(F.linear(xn2, t("mlp/gate_proj.weight"))) * F.linear(xn2, t("mlp/up_proj.weight"))
    np.testing.assert_allclose(act.numpy(), g["act"], rtol=0, atol=1e-5)
    np.testing.assert_allclose((h_mid + F.linear(act, t("mlp/down_proj.weight"))).numpy(), g["h_out"], rtol=0, atol=2e-5)
