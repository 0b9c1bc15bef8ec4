#!/usr/bin/env python3
"""In-kernel evidence for the helper blocks' prefetch (GPU box):  python tools/prefetch_stamps.py
rocprofv3's serialised dispatches empty the L2 between kernels (profiles/r03_pmc_prefetch.txt: gate_up misses 141 468 lines with
the prefetch on AND off), so counters cannot show whether the lines a producer's helpers requested are there when the consumer
starts.  s_memrealtime stamps inside the consumer can: the stamped kernel runs right behind its producers (debug_stamps kernel
id + 32), once in a process with the prefetch on and once with SPARKMI_NO_PREFETCH=1 (child processes: the switch is read at
create).  'mfma' = all weight tiles have arrived and been multiplied."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(%r, "spark-tts_amd"))
import numpy as np, torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
cfg = Cf.spark_0p5b_llm()
llm = SparkLLM(cfg, W.SyntheticLLM(cfg), "cuda:0", max_positions=512, diag=True)
llm.prefill([np.random.Generator(np.random.PCG64(1)).integers(0, cfg.vocab_size, size=128).tolist()]); llm.decode(20); torch.cuda.synchronize()
f = llm._lib.smi_llm_debug_stamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
names = {0: "entry", 1: "prologue", 4: "mfma", 5: "reduce-barrier", 6: "epilogue"}
for kname, kid in [("qkv", 0), ("gate_up", 3), ("down", 4)]:
    for mode, add in (("alone", 0), ("behind its producers", 32)):
        acc = np.zeros(8); n = 0
        for rep in range(3):
            for layer in range(4, 20):
                out = (C.c_double * 8)()
                assert f(llm._h, kid + add, layer, out) == 0, llm._lib.smi_last_error()
                acc += np.array(list(out)); n += 1
        acc /= n
        print(f"  {kname:8s} {mode:22s} " + " | ".join(f"{nm} {acc[i]:.2f}" for i, nm in names.items()), flush=True)
for k in ("qkv", "attn", "gate_up", "down"):
    print(f"  in-sequence {k:8s} {llm.time_kernel(k, iters=96, in_sequence=True) * 1e3:6.2f} us")
print(f"  decode step {llm.time_kernel('step', iters=200) * 1e3:7.1f} us")
''' % ROOT
for label, env in (("prefetch ON (default)", {}), ("prefetch OFF (SPARKMI_NO_PREFETCH=1)", {"SPARKMI_NO_PREFETCH": "1"})):
    print(f"--- {label}", flush=True)
    e = dict(os.environ); e.pop("SPARKMI_NO_PREFETCH", None); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=500)
    print(r.stdout, end="")
    if r.returncode:
        print(r.stderr[-2000:]); sys.exit(1)
