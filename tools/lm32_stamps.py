#!/usr/bin/env python3
"""Where a streamed group of k_lm32 (lm_head at 17..32 rows) spends its time (GPU box): python tools/lm32_stamps.py [rows]
Block 0, wave 0: per group, microseconds from the group's start to: MFMAs issued + partial sums written | reduce barrier passed |
group done (epilogue + second barrier); and the gap to the next group's start."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = Cf.spark_0p5b_llm()
llm = SparkLLM(cfg, W.SyntheticLLM(cfg), "cuda:0", max_slots=B, max_positions=512, diag=True)
rng = np.random.Generator(np.random.PCG64(1))
llm.prefill([rng.integers(0, cfg.vocab_size, size=128).tolist() for _ in range(B)]); llm.decode(8); torch.cuda.synchronize()
out = (C.c_double * 8)()
for rep in range(3):
    llm._lib.check(llm._lib.smi_llm_debug_stamps(llm._h, 5, 0, out), "stamps")
    raw = (C.c_uint64 * 128)()
    llm._lib.check(llm._lib.smi_llm_debug_raw_stamps(llm._h, raw, 128), "raw")
    t = np.array(raw, dtype=np.float64).reshape(32, 4) * 0.01
    t = t[t[:, 0] > 0]
    rel = t - t[:, :1]
    gap = np.append(t[1:, 0] - t[:-1, 3], 0.0)
    print(f"rep {rep}: {len(t)} groups, total {t[-1, 3] - t[0, 0]:.1f} us")
    for i in (0, 1, 2, 5, 10, len(t) - 2):
        print(f"  group {i:2d}: mfma+write {rel[i, 1]:.2f} | barrier {rel[i, 2]:.2f} | done {rel[i, 3]:.2f} | gap to next {gap[i]:.2f}")
    print(f"  mean: mfma+write {rel[:, 1].mean():.2f} | barrier {rel[:, 2].mean():.2f} | done {rel[:, 3].mean():.2f}")
print("lm_head eager", round(llm.time_kernel("lm_head", iters=24) * 1e3, 2), "us")
