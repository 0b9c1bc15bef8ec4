#!/usr/bin/env python3
"""Prefill time of 32 x 128-token prompts (the k_pgemm path) for one or more library builds:
   python tools/pgemm_ab.py [lib.so ...]   (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(%r, "spark-tts_amd"))
import torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
cfg = Cf.spark_0p5b_llm()
res = []
llm = None
for B, P in ((32, 128), (64, 128), (8, 400)):
    del llm
    llm = SparkLLM(cfg, W.SyntheticLLM(cfg), "cuda:0", max_slots=B, max_positions=512, diag=True)
    prompts = [np.random.Generator(np.random.PCG64(1 + b)).integers(0, cfg.vocab_size, size=P).tolist() for b in range(B)]
    ts = []
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        llm.prefill(prompts); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    llm.decode(3)
    tok = llm.tokens(4)
    res.append((B, P, round(min(ts), 2), [t[:4] for t in tok[:2]]))
print(os.environ.get("SPARKMI_LIB", "default"), res)
''' % ROOT
for lib in sys.argv[1:] or [""]:
    env = dict(os.environ)
    if lib:
        env["SPARKMI_LIB"] = os.path.join(ROOT, lib)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-1500:], flush=True)
