#!/usr/bin/env python3
"""Per-kernel decode probes at a given batch: python tools/probe_batch.py 32  (GPU box)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
from sparkmi.arena import llm_cfg_struct, pack_llm_arena
cfg = Cf.spark_0p5b_llm()
arena = torch.from_numpy(pack_llm_arena(cfg, W.SyntheticLLM(cfg), llm_cfg_struct(cfg, 1, 512, "bf16", True))).to("cuda:0")
for B in [int(a) for a in sys.argv[1:]] or [1, 8, 32]:
    llm = SparkLLM(cfg, None, "cuda:0", max_slots=B, max_positions=512, arena=arena, diag=True)
    rng = np.random.Generator(np.random.PCG64(1))
    prompts = [rng.integers(0, cfg.vocab_size, size=128).tolist() for _ in range(B)]
    torch.cuda.synchronize(); import time; t0 = time.perf_counter()
    llm.prefill(prompts); torch.cuda.synchronize(); t_pre = time.perf_counter() - t0
    llm.decode(40); torch.cuda.synchronize()
    out = {n: round(llm.time_kernel(n, iters=48) * 1e3, 2) for n in ("qkv", "attn", "o_proj", "gate_up", "down", "lm_head", "finalize")}
    llm.prefill(prompts); llm.decode(8)
    step = round(llm.time_kernel("step", iters=60) * 1e3, 1)
    print(f"B={B} prefill {t_pre * 1e3:.2f} ms  kernels(us) {out}  graph step {step} us", flush=True)
    del llm
