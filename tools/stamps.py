#!/usr/bin/env python3
"""In-kernel phase timing of the decode GEMM kernels: python tools/stamps.py  (GPU box)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
cfg = Cf.spark_0p5b_llm()
llm = SparkLLM(cfg, W.SyntheticLLM(cfg), "cuda:0", max_positions=512, diag=True)
prompt = np.random.Generator(np.random.PCG64(1)).integers(0, cfg.vocab_size, size=128).tolist()
llm.prefill([prompt]); llm.decode(40); torch.cuda.synchronize()
f = llm._lib.smi_llm_debug_stamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
names = ["entry", "w issued", "prologue done", "barrier1", "mfma done", "reduce barrier", "epilogue done"]
for kname, kid in [("qkv", 0), ("o_proj", 2), ("gate_up", 3), ("down", 4)]:
    acc = np.zeros(8)
    for layer in range(4, 12):
        out = (C.c_double * 8)()
        assert f(llm._h, kid, layer, out) == 0, llm._lib.smi_last_error()
        acc += np.array(list(out))
    acc /= 8
    print(f"{kname:8s} clock {acc[7]:6.0f} MHz | " + " | ".join(f"{n} {acc[i]:.2f}" for i, n in enumerate(names)))
    for name in ("qkv", "attn", "o_proj", "gate_up", "down", "lm_head"):
        pass
for name in ("qkv", "attn", "o_proj", "gate_up", "down", "lm_head", "finalize", "step"):
    print(name, round(llm.time_kernel(name, iters=48) * 1e3, 2), "us (eager back-to-back)")
