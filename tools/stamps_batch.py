#!/usr/bin/env python3
"""In-kernel phase stamps of the decode GEMM kernels at a given number of live rows (GPU box):
    python tools/stamps_batch.py 32 [16 64 ...]
Per kernel: mean over the blocks of (stamp - first block's entry), microseconds: entry (= dispatch skew), prologue done
(operands staged / RMSNorm factors), MFMA loop done, reduce barrier passed, epilogue done; then eager back-to-back timings."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
cfg = Cf.spark_0p5b_llm()
names = {0: "entry", 1: "prologue", 4: "mfma", 5: "reduce-barrier", 6: "epilogue"}
for B in [int(a) for a in sys.argv[1:]] or [32]:
    llm = SparkLLM(cfg, W.SyntheticLLM(cfg), "cuda:0", max_slots=B, max_positions=512, diag=True)
    rng = np.random.Generator(np.random.PCG64(1))
    prompts = [rng.integers(0, cfg.vocab_size, size=128).tolist() for _ in range(B)]
    llm.prefill(prompts); llm.decode(20); torch.cuda.synchronize()
    f = llm._lib.smi_llm_debug_stamps
    f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    print(f"--- {B} rows")
    for kname, kid in [("qkv", 0), ("o_proj", 2), ("gate_up", 3), ("down", 4)]:
        acc = np.zeros(8)
        for layer in range(4, 12):
            out = (C.c_double * 8)()
            assert f(llm._h, kid, layer, out) == 0, llm._lib.smi_last_error()
            acc += np.array(list(out))
        acc /= 8
        print(f"{kname:8s} " + " | ".join(f"{n} {acc[i]:.2f}" for i, n in names.items()), flush=True)
    for name in ("qkv", "attn", "o_proj", "gate_up", "down", "lm_head", "step"):
        print(f"  {name:8s} {llm.time_kernel(name, iters=48) * 1e3:7.2f} us", flush=True)
    del llm
