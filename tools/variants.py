#!/usr/bin/env python3
"""A/B of k_gemm load-placement variants (SPARKMI_VARIANT): python tools/variants.py  (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(%r, "spark-tts_amd"))
import torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
cfg = Cf.spark_0p5b_llm()
B = int(os.environ.get("VARIANTS_B", "1"))
llm = SparkLLM(cfg, W.SyntheticLLM(cfg), "cuda:0", max_slots=B, max_positions=512, diag=True)
prompt = [np.random.Generator(np.random.PCG64(1 + b)).integers(0, cfg.vocab_size, size=128).tolist() for b in range(B)]
llm.prefill(prompt); llm.decode(40); torch.cuda.synchronize()
out = []
for rep in range(2):
    out = [round(llm.time_kernel(n, iters=96) * 1e3, 2) for n in ("qkv", "attn", "o_proj", "gate_up", "down", "lm_head", "finalize")]
llm.prefill(prompt); llm.decode(8)
step = round(llm.time_kernel("step", iters=100) * 1e3, 1)
print("variant", os.environ.get("SPARKMI_VARIANT", "0"), out, "graph step", step)
''' % ROOT
for v in sys.argv[1:] or ["0", "1", "2", "3", "4", "5", "6", "7"]:
    env = dict(os.environ, SPARKMI_VARIANT=v)
    if v.startswith("env:"):
        k, val = v[4:].split("=", 1)
        env = dict(os.environ, SPARKMI_VARIANT=v, **{k: val})
    if v.startswith("tune:"):
        env = dict(os.environ, SPARKMI_TUNE=v[5:], SPARKMI_VARIANT=v)
    if v.startswith("lib:"):
        env = dict(os.environ, SPARKMI_LIB=os.path.join(ROOT, v[4:]), SPARKMI_VARIANT="lib")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-800:], flush=True)
