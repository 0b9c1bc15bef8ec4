#!/usr/bin/env python3
"""Per-launch vocoder timing: python tools/voc_profile.py [B] [T]  (GPU box)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import torch
from sparkmi import config as C, weights as W
from sparkmi.bicodec import BiCodecVocoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 150
cfg = C.spark_0p5b_bicodec()
voc = BiCodecVocoder(cfg, W.bicodec_detok_state(cfg), "cuda:0", max_batch=B, max_frames=T + 10,
                     diag=any(k.startswith("SPARKMI_") for k in os.environ))   # SPARKMI_* switches exist in the diagnostics build only
rng = np.random.Generator(np.random.PCG64(3))
sem = torch.from_numpy(rng.integers(0, 8192, size=(B, T)))
glob = torch.from_numpy(rng.integers(0, 4096, size=(B, 1, 32)))
for _ in range(3):
    wav = voc.detokenize(sem, glob)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    wav = voc.detokenize(sem, glob)
e1.record(); torch.cuda.synchronize()
print(f"B={B} T={T}: forward {e0.elapsed_time(e1) / 5:.3f} ms  ({voc.launches()} launches)")
rows = [voc.time_launch(i, iters=5) for i in range(voc.launches())]
tot = sum(r[1] for r in rows); fl = sum(r[2] for r in rows)
print(f"sum of launches {tot:.3f} ms, {fl / 1e9:.1f} GFLOP, {fl / tot / 1e9:.1f} TFLOP/s")
groups = {}
for n, ms, f in rows:
    key = n.split(".convnext")[0] if "convnext" in n else n
    import re
    key = re.sub(r"\.\d+\.", ".N.", n)
    g = groups.setdefault(key, [0, 0.0, 0.0]); g[0] += 1; g[1] += ms; g[2] += f
for k, (c, ms, f) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{ms:8.3f} ms  x{c:3d}  {f / ms / 1e9 if ms else 0:7.1f} TF/s  {k}")
if os.environ.get("VOC_PROFILE_ALL"):
    for n, ms, f in rows:
        print(f"{ms:8.3f} ms {f / ms / 1e9 if ms else 0:7.1f} TF/s  {n}")
