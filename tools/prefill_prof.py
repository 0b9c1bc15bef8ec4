#!/usr/bin/env python3
"""32 x 128-token prefill, five times (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
cfg = Cf.spark_0p5b_llm()
B, P = 32, 128
llm = SparkLLM(cfg, W.SyntheticLLM(cfg), "cuda:0", max_slots=B, max_positions=512)
prompts = [np.random.Generator(np.random.PCG64(1 + b)).integers(0, cfg.vocab_size, size=P).tolist() for b in range(B)]
for it in range(5):
    llm.prefill(prompts)
torch.cuda.synchronize()
print("done")
