#!/usr/bin/env python3
"""Attention kernel time vs batch and context (GPU box): python tools/attn_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import numpy as np, torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
from sparkmi.arena import llm_cfg_struct, pack_llm_arena
cfg = Cf.spark_0p5b_llm()
arena = torch.from_numpy(pack_llm_arena(cfg, W.SyntheticLLM(cfg), llm_cfg_struct(cfg, 1, 3200, "bf16", True))).to("cuda:0")
for B in (1, 32):
    for P in (128, 500, 1500, 3000):
        llm = SparkLLM(cfg, None, "cuda:0", max_slots=B, max_positions=3200, arena=arena, diag=True)
        prompts = [np.random.Generator(np.random.PCG64(1 + b)).integers(0, cfg.vocab_size, size=P).tolist() for b in range(B)]
        llm.prefill(prompts); llm.decode(4); torch.cuda.synchronize()
        t = llm.time_kernel("attn", iters=96, in_sequence=True) * 1e3
        print(f"B={B:2d} ctx~{P + 5:4d}: attention {t:6.2f} us in sequence", flush=True)
