#!/usr/bin/env python3
"""Round-4 batch decode A/B on one box (GPU box): python tools/r04_batch_ab.py 32 16 8
For each row count: eager per-kernel loops, in-sequence per-kernel times (as inside the graph) and the graph step, for the
default build path and with the chain-split down_proj turned off (SPARKMI_DC_MIN=1000: the round-3 kernels)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(%r, "spark-tts_amd"))
import torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
cfg = Cf.spark_0p5b_llm()
B = int(os.environ["AB_B"])
llm = SparkLLM(cfg, W.SyntheticLLM(cfg), "cuda:0", max_slots=B, max_positions=512, diag=True)
prompt = [np.random.Generator(np.random.PCG64(1 + b)).integers(0, cfg.vocab_size, size=128).tolist() for b in range(B)]
llm.prefill(prompt); llm.decode(40); torch.cuda.synchronize()
names = ("qkv", "attn", "o_proj", "gate_up", "down", "lm_head", "finalize")
eager = [round(llm.time_kernel(n, iters=96) * 1e3, 2) for n in names]
seq = [round(llm.time_kernel(n, iters=96, in_sequence=True) * 1e3, 2) for n in names[:5]]
llm.prefill(prompt); llm.decode(8)
steps = [round(llm.time_kernel("step", iters=100) * 1e3, 1) for _ in range(2)]
print("B", B, os.environ.get("AB_TAG"), "eager", eager, "in-seq", seq, "graph step", steps, flush=True)
''' % ROOT
for B in sys.argv[1:] or ["32"]:
    for rep in range(2):
        variants = (("new", {}), ("old-down", {"SPARKMI_DC_MIN": "1000"})) if int(B) > 8 else (("fused o_proj (FUSE2)", {}), ("separate o_proj", {"SPARKMI_FUSE2_ROWS": "0"}))
        if int(B) == 1:
            variants = (("per-head partials summed by gate_up (default)", {}), ("last-arriver head sum", {"SPARKMI_FUSE1_LAST": "1"}))
        for tag, env in variants:
            subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AB_B=B, AB_TAG=tag, **env), check=False)
