#!/usr/bin/env python3
"""Decode step with the paged KV cache against the contiguous cache at 0.5B (GPU box):  python tools/paged_time.py
A paged engine (kv_page_tokens > 0, the serving configuration) runs the same kernels as a contiguous one since round 3 -- the
one-row attention with the fused o_proj and the slot == row batch kernel look a token's row up through the slot's page-table row
(k_attn<.., PG = 1>) -- so what is printed here is the cost of that lookup (+17 us at one row, +22 at 32 rows at 0.5B)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import numpy as np, torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
cfg = Cf.spark_0p5b_llm()
syn = W.SyntheticLLM(cfg)
arena = None
for B in (1, 8, 32):
    prompts = [np.random.Generator(np.random.PCG64(1 + b)).integers(0, cfg.vocab_size, size=128).tolist() for b in range(B)]
    row = []
    for name, kw in (("contiguous", {}), ("paged, 64-token pages", dict(kv_page_tokens=64, kv_pages=B * 8 + 8))):
        llm = SparkLLM(cfg, syn if arena is None else None, "cuda:0", max_slots=B, max_positions=512, arena=arena, **kw, diag=True)
        arena = llm.arena
        llm.prefill(prompts); llm.decode(8); torch.cuda.synchronize()
        us = min(llm.time_kernel("step", iters=100) for _ in range(3)) * 1e3
        row.append((name, us))
        del llm
    print(f"{B:3d} rows: " + "   ".join(f"{n} {u:7.1f} us" for n, u in row) + f"   (+{row[1][1] - row[0][1]:.1f} us)", flush=True)
