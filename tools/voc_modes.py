#!/usr/bin/env python3
"""Vocoder: exact-fp32 matrix pipe vs the bf16-split pipe (default) on the 0.5B golden vectors -- error and time (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import numpy as np, torch
from sparkmi import config as C, weights as W
from sparkmi.bicodec import BiCodecVocoder
cfg = C.spark_0p5b_bicodec()
sd = W.bicodec_detok_state(cfg)
g = np.load(os.path.join(ROOT, "tests", "golden", "vocoder_full.npz"))
rng = np.random.Generator(np.random.PCG64(7))
sem32 = torch.from_numpy(rng.integers(0, cfg.codebook_size, size=(32, 150)))
glob32 = torch.from_numpy(rng.integers(0, 4096, size=(32, 1, cfg.spk_token_num)))
out = {}
for exact in (True, False):
    voc = BiCodecVocoder(cfg, sd, "cuda:0", max_batch=32, max_frames=160, exact_fp32=exact)
    w = voc.detokenize(torch.from_numpy(g["c0_semantic"]), torch.from_numpy(g["c0_global"])).cpu().numpy()
    err = np.abs(w - g["c0_wav"]).max()
    def t(sem, glob, n=5):
        voc.detokenize(sem, glob); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            voc.detokenize(sem, glob)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    t1 = t(torch.from_numpy(g["c0_semantic"]), torch.from_numpy(g["c0_global"]), 20)
    t32 = t(sem32, glob32, 5)
    w32 = voc.detokenize(sem32, glob32).cpu().numpy()
    out[exact] = (w, w32)
    print(f"exact_fp32={exact}: max|wav - reference vector| {err:.3e};  B=1 {t1:.2f} ms, B=32 {t32:.1f} ms", flush=True)
    if not exact:
        voc.detokenize(sem32, glob32)
        rows = []
        for i in range(voc.launches()):
            nm, ms, fl = voc.time_launch(i, iters=3)
            rows.append((ms, nm, fl))
        tot = sum(r[0] for r in rows)
        print(f"  B=32 launches: {len(rows)}, sum {tot:.1f} ms")
        for ms, nm, fl in sorted(rows, reverse=True)[:int(os.environ.get("VOC_TOP", "12"))]:
            print(f"   {nm:45s} {ms:7.3f} ms  {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:7.1f} TFLOP/s (fp32-equivalent)")
print(f"bf16-split vs exact: B=1 max|d| {np.abs(out[True][0] - out[False][0]).max():.3e}, B=32 max|d| {np.abs(out[True][1] - out[False][1]).max():.3e}")
