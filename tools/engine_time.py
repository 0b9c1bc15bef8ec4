#!/usr/bin/env python3
"""One-row decode step at the 0.5B shape with the persistent-layer engine (csrc/smi_eng.h) and with the four launches per
layer it replaces, on one box (GPU): whole step (graph replay), the layers alone, tokens equal.  SPARKMI_ENGINE_STAMPS=1
adds the per-edge stamps of the last engine launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import numpy as np
import torch
from sparkmi import config as C, weights as W
from sparkmi.arena import llm_cfg_struct, pack_llm_arena
from sparkmi.llm import SparkLLM

cfg = C.spark_0p5b_llm()
MAXPOS = 704
t0 = time.time()
arena = torch.from_numpy(pack_llm_arena(cfg, W.SyntheticLLM(cfg), llm_cfg_struct(cfg, 1, MAXPOS, "bf16", True))).to("cuda:0")
print(f"arena packed in {time.time() - t0:.1f}s", flush=True)
prompt = np.random.Generator(np.random.PCG64(1234)).integers(0, cfg.vocab_size, size=128).tolist()
# loader pacing variants (SPARKMI_ENGINE_BURST fills per look at the arrival counter, SPARKMI_ENGINE_SLEEP x 64-cycle pauses):
# python tools/engine_time.py 4,0 1,2 16,0
for knob in sys.argv[1:]:
    b, z, q = (knob.split(",") + ["1"])[:3]
    os.environ["SPARKMI_ENGINE_BURST"], os.environ["SPARKMI_ENGINE_SLEEP"], os.environ["SPARKMI_ENGINE_POLL"] = b, z, q
    v = SparkLLM(cfg, None, "cuda:0", max_positions=MAXPOS, arena=arena, diag=True)
    v.set_engine(True)
    v.prefill([prompt]); v.decode(20)
    lay = v.time_kernel("layers", iters=32) * 1e3
    print(f"burst {b} sleep {z} quiet-poll {q}: layers {lay:7.1f} us ({lay / cfg.num_hidden_layers:5.2f} per layer)", flush=True)
    v.close()
for k in ("SPARKMI_ENGINE_BURST", "SPARKMI_ENGINE_SLEEP", "SPARKMI_ENGINE_POLL"):
    os.environ.pop(k, None)
llm = SparkLLM(cfg, None, "cuda:0", max_positions=MAXPOS, arena=arena, diag=True)
llm.set_engine(True)
print("engine:", llm.engine_info(), flush=True)
res = {}
for on in (True, False):
    llm.set_engine(on)
    toks = llm.generate_ids([prompt], 150)[0]
    llm.prefill([prompt]); llm.decode(20)
    step = llm.time_kernel("step", iters=128) * 1e3
    llm.prefill([prompt]); llm.decode(20)
    lay = llm.time_kernel("layers", iters=32) * 1e3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    llm.generate_ids([prompt], 150)
    wall = (time.perf_counter() - t0) * 1e3
    print(f"engine {'on ' if on else 'off'}: step {step:7.1f} us   layers {lay:7.1f} us ({lay / cfg.num_hidden_layers:5.2f} per layer)   "
          f"150-token generate {wall:6.1f} ms   first tokens {toks[:6]}", flush=True)
    res[on] = toks
assert res[True] == res[False], "engine and launch path disagree"
print("tokens identical over 150 steps")
if os.environ.get("SPARKMI_ENGINE_STAMPS"):
    llm.set_engine(True)
    llm.prefill([prompt]); llm.decode(40)
    s = llm.engine_stamps()
    names = ["A h", "B qkv", "C attn", "D h_mid", "E act", "QKV done", "gate_up done", "down done"]
    d0, dh = s[0], s[1]
    order = [0, 5, 2, 3, 6, 4, 7]
    print("CU 0   (us after the layer's A, in time order):", "  ".join(f"{names[i]} {np.mean(d0[2:, i] - d0[2:, 0]):5.2f}" for i in order))
    print("head CU (us after the layer's A):", "  ".join(f"{names[i]} {np.mean(dh[2:, i] - dh[2:, 0]):5.2f}" for i in (0, 5, 1, 3, 6)), f" attention published {np.mean(dh[2:, 2] - dh[2:, 0]):5.2f}")
    m = lambda i, j: float(np.mean(d0[2:, i] - d0[2:, j]))
    print(f"CU 0 wave 0, QKV window: barrier -> jobs start {m(8, 0):.2f}, jobs {m(9, 8):.2f}, epilogue {m(10, 9):.2f}, -> barrier out {m(5, 10):.2f}")
    print(f"CU 0 wave 0, gate_up window: barrier -> jobs start {m(11, 3):.2f}, jobs {m(12, 11):.2f}, epilogue {m(13, 12):.2f}, -> barrier out {m(6, 13):.2f}")
    print(f"CU 0 wave 0, act hand-off: gate_up barrier out -> sweep done {m(14, 6):.2f}, staging {m(15, 14):.2f}, -> barrier out {m(4, 15):.2f}")
    d7 = s[2]
    m7 = lambda i, j: float(np.mean(d7[2:, i] - d7[2:, j]))
    print(f"CU 0 wave 7 (loader), QKV window: barrier -> jobs start {m7(8, 0):.2f}, jobs {m7(9, 8):.2f}, epilogue {m7(10, 9):.2f}, -> barrier out {m7(5, 10):.2f}")
    print(f"CU 0 wave 7 (loader), gate_up window: barrier -> jobs start {m7(11, 3):.2f}, jobs {m7(12, 11):.2f}, epilogue {m7(13, 12):.2f}, -> barrier out {m7(6, 13):.2f}")
    raw = llm.engine_stamps()   # (microseconds since the first stamp; slot 14 of row 2 holds shader CYCLES scaled the same way)
    cyc = (raw[2][-1, 14] - raw[2][2, 14]) * 100.0     # undo the 0.01 scaling: cycles
    us = raw[2][-1, 0] - raw[2][2, 0]
    print(f"shader clock during the launch: {cyc / us:.0f} MHz")
    print("layer period (CU 0):", float(np.mean(np.diff(d0[2:, 0]))), "us")
