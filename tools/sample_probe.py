#!/usr/bin/env python3
"""lm_head and token-selection kernel times, greedy vs sampling (GPU box): python tools/sample_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import numpy as np, torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
cfg = Cf.spark_0p5b_llm()
for B in (1, 8):
    llm = SparkLLM(cfg, W.SyntheticLLM(cfg), "cuda:0", max_slots=B, max_positions=512, diag=True)
    prompts = [np.random.Generator(np.random.PCG64(1 + b)).integers(0, cfg.vocab_size, size=128).tolist() for b in range(B)]
    for mode in (False, True):
        llm.set_sampling(mode, seed=1)
        llm.prefill(prompts); llm.decode(8); torch.cuda.synchronize()
        r = [round(llm.time_kernel(n, iters=64, in_sequence=s) * 1e3, 2) for n, s in (("lm_head", False), ("finalize", False))]
        step = round(llm.time_kernel("step", iters=100) * 1e3, 1)
        print(f"B={B} do_sample={mode}: (lm_head, token selection) us {r} step {step}", flush=True)
