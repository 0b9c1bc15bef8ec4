#!/usr/bin/env python3
"""Prefill only, for rocprofv3 --kernel-trace --stats (GPU box):  PF=32x128 rocprofv3 --kernel-trace --stats -d out -- python3 tools/prefill_prof.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
from sparkmi.arena import llm_cfg_struct, pack_llm_arena
B, P = (int(v) for v in os.environ.get("PF", "32x128").split("x"))
cfg = Cf.spark_0p5b_llm()
arena = torch.from_numpy(pack_llm_arena(cfg, W.SyntheticLLM(cfg), llm_cfg_struct(cfg, 1, 1024, "bf16", True))).to("cuda:0")
llm = SparkLLM(cfg, None, "cuda:0", max_slots=B, max_positions=1024, arena=arena, diag=True)
prompts = [np.random.Generator(np.random.PCG64(1 + b)).integers(0, cfg.vocab_size, size=P).tolist() for b in range(B)]
for it in range(int(os.environ.get("PF_ITERS", "5"))):
    llm.prefill(prompts)
    torch.cuda.synchronize()
print("ok", llm.tokens(1)[0][0])
