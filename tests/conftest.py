import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "spark-tts_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

FULL_MAX_POS = 704   # one arena layout for every 0.5B-size LLM test (the RoPE section's size is part of the layout)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def full_llm():
    """(cfg, synthetic weights, device arena) of the Spark-TTS-0.5B shape, packed ONCE per session (43 s of host work):
    every full-size LLM test builds its engines on this arena with max_positions == FULL_MAX_POS."""
    import torch
    from sparkmi import config as C, weights as W
    from sparkmi.arena import llm_cfg_struct, pack_llm_arena
    cfg = C.spark_0p5b_llm()
    syn = W.SyntheticLLM(cfg)
    arena = torch.from_numpy(pack_llm_arena(cfg, syn, llm_cfg_struct(cfg, 1, FULL_MAX_POS, "bf16", True))).to("cuda:0")
    return cfg, syn, arena


@pytest.fixture(scope="session")
def full_llm_oracle(full_llm):
    """The fp32 CPU oracle at the 0.5B shape (2 GB of weights, built once); tests set ``.kv_dtype`` as they need."""
    from oracle.llm_ref import Qwen2Ref
    cfg, syn, _ = full_llm
    return Qwen2Ref(cfg, syn)


@pytest.fixture(scope="session")
def full_voc():
    """(cfg, unfolded synthetic state, folded state) of the 0.5B BiCodec detokenizer side."""
    from sparkmi import config as C, weights as W
    cfg = C.spark_0p5b_bicodec()
    sd = W.bicodec_detok_state(cfg)
    return cfg, sd, W.fold_weight_norm(sd)
