#!/usr/bin/env python3
"""Greedy vs sampling decode speed at full size (GPU box): python tools/sample_time.py"""
import os, sys, time
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
        ts = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = llm.generate_ids(prompts, 150, do_sample=mode, seed=it)
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"B={B} do_sample={mode}: {min(ts):.1f} ms for 150 tokens ({(min(ts) - 3) / 149 * 1e3:.0f} us/step incl. prefill share)")
