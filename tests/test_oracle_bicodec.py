"""The vocoder oracle against vectors produced by the reference's own modules
(tests/golden/gen_golden.py, run in the build container where /root/reference is mounted)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.bicodec_ref import BiCodecDetokRef, fsq_codes, snake
from sparkmi import config as C, weights as W
from sparkmi.pipeline_text import build_clone_prompt, build_control_prompt, parse_global, parse_semantic


@pytest.fixture(scope="module")
def tiny():
    cfg = C.tiny_bicodec()
    return cfg, BiCodecDetokRef(cfg, W.fold_weight_norm(W.bicodec_detok_state(cfg)))


def test_ops_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    y = snake(torch.from_numpy(g["snake_x"]), torch.from_numpy(g["snake_alpha"]))
    np.testing.assert_array_equal(y.numpy(), g["snake_y"])
    # SamplingBlock with ratio 1: (B,T,C) in, (B,C,T) out, value 3x (samper.py:79-100)
    x = torch.from_numpy(g["sampling_x"]).transpose(1, 2)
    np.testing.assert_array_equal((x + x + x).numpy(), g["sampling_y"])
    codes = fsq_codes(torch.arange(4096), [4] * 6)
    np.testing.assert_array_equal(codes.numpy(), g["fsq_codebook"])


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_detokenize_matches_reference(tiny, golden_dir, case):
    cfg, ref = tiny
    g = np.load(os.path.join(golden_dir, "vocoder_tiny.npz"))
    st = {}
    wav = ref.detokenize(torch.from_numpy(g[f"c{case}_semantic"]), torch.from_numpy(g[f"c{case}_global"]), st)
    np.testing.assert_allclose(wav.numpy(), g[f"c{case}_wav"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(st["d_vector"].numpy(), g[f"c{case}_d_vector"], rtol=0, atol=1e-6)
    if f"c{case}_z_q" in g:
        np.testing.assert_allclose(st["z_q"].numpy(), g[f"c{case}_z_q"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(st["prenet_plus_d"].numpy(), g[f"c{case}_prenet_plus_d"], rtol=0, atol=1e-5)
        for i, s in enumerate(st["wavegen"]):
            np.testing.assert_allclose(s.numpy(), g[f"c{case}_wavegen{i}"], rtol=0, atol=1e-5)
    assert wav.shape[-1] == cfg.hop * g[f"c{case}_semantic"].shape[1]


def test_full_size_matches_reference(golden_dir):
    cfg = C.spark_0p5b_bicodec()
    ref = BiCodecDetokRef(cfg, W.fold_weight_norm(W.bicodec_detok_state(cfg)))
    g = np.load(os.path.join(golden_dir, "vocoder_full.npz"))
    wav = ref.detokenize(torch.from_numpy(g["c1_semantic"]), torch.from_numpy(g["c1_global"]))
    np.testing.assert_allclose(wav.numpy(), g["c1_wav"], rtol=0, atol=2e-6)


def test_batch_rows_are_independent(tiny):
    cfg, ref = tiny
    rng = np.random.Generator(np.random.PCG64(3))
    sem = torch.from_numpy(rng.integers(0, cfg.codebook_size, size=(3, 9)))
    glob = torch.from_numpy(rng.integers(0, 4096, size=(3, 1, cfg.spk_token_num)))
    full = ref.detokenize(sem, glob)
    for b in range(3):
        one = ref.detokenize(sem[b:b + 1], glob[b:b + 1])
        np.testing.assert_allclose(full[b:b + 1].numpy(), one.numpy(), rtol=0, atol=1e-5)


def test_weight_norm_fold_shapes():
    cfg = C.tiny_bicodec()
    sd = W.bicodec_detok_state(cfg)
    f = W.fold_weight_norm(sd)
    assert not any(k.endswith(("weight_g", "weight_v")) for k in f)
    # ConvTranspose1d: (C_in, C_out, k), norm per input channel
    v, g = sd["decoder.model.1.block.1.weight_v"], sd["decoder.model.1.block.1.weight_g"]
    w = f["decoder.model.1.block.1.weight"]
    n = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True))
    np.testing.assert_allclose(w, v * (g / n), rtol=1e-6)


def test_prompt_strings(golden_dir):
    d = json.load(open(os.path.join(golden_dir, "prompts.json")))
    for c in d["cases"]:
        if c["kind"] == "control":
            assert build_control_prompt(c["gender"], c["pitch"], c["speed"], c["text"]) == c["expect"]
        else:
            assert build_clone_prompt(c["text"], c["glob"], c["sem"], c["prompt_text"]) == c["expect"]
    with pytest.raises(AssertionError):
        build_control_prompt("robot", "low", "low", "x")
    with pytest.raises(AssertionError):
        build_control_prompt("male", "loud", "low", "x")
    s = "<|bicodec_semantic_12|><|bicodec_global_7|><|bicodec_semantic_8191|>"
    assert parse_semantic(s) == [12, 8191] and parse_global(s) == [7]
