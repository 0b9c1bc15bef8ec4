#!/usr/bin/env python3
"""In-flight batching overhead at full size (GPU box): 32 requests x 150 tokens, 8 live, vs four static batches of 8."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import numpy as np, torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
cfg = Cf.spark_0p5b_llm()
llm = SparkLLM(cfg, W.SyntheticLLM(cfg), "cuda:0", max_slots=8, max_positions=512)
prompts = [np.random.Generator(np.random.PCG64(1 + b)).integers(0, cfg.vocab_size, size=128).tolist() for b in range(32)]
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(0, 32, 8):
        llm.generate_ids(prompts[i:i + 8], 150)
    t_static = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = sum(len(t) for _, t in llm.serve(((i, p, 150, None) for i, p in enumerate(prompts)), max_live=8, decode_stride=10))
    t_serve = time.perf_counter() - t0
    print(f"static 4 x batch 8: {t_static * 1e3:.1f} ms; serve (8 live, stride 10): {t_serve * 1e3:.1f} ms for {n} tokens", flush=True)
# ragged budgets: in-flight batching should win
budgets = [int(x) for x in np.random.Generator(np.random.PCG64(5)).integers(30, 300, size=32)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(0, 32, 8):
    llm.generate_ids(prompts[i:i + 8], max(budgets[i:i + 8]))
t_static = time.perf_counter() - t0
torch.cuda.synchronize(); t0 = time.perf_counter()
n = sum(len(t) for _, t in llm.serve(((i, p, budgets[i], None) for i, p in enumerate(prompts)), max_live=8, decode_stride=10))
print(f"ragged budgets 30..300: static (each batch runs to its longest) {t_static * 1e3:.1f} ms; serve {(time.perf_counter() - t0) * 1e3:.1f} ms for {n} tokens", flush=True)
