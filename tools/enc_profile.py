#!/usr/bin/env python3
"""Per-launch timing of the prompt encoder at full size (GPU box): python tools/enc_profile.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import numpy as np, torch
from sparkmi import config as C, config_tok as T, weights as W
from sparkmi.encoder import BiCodecEncoder, get_ref_clip
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
wcfg, tcfg, vcfg = T.xlsr53(), T.spark_0p5b_tok(), C.spark_0p5b_bicodec()
enc = BiCodecEncoder(wcfg, tcfg, W.fold_pos_conv_weight_norm(W.wav2vec2_state(wcfg)),
                     W.fold_weight_norm(W.bicodec_tok_state(tcfg, vcfg.vq_input_dim)), "cuda:0", max_seconds=max(secs, 6.0))
wav = (0.1 * np.random.default_rng(0).standard_normal(int(16000 * secs))).astype(np.float32)
ref = get_ref_clip(wav, 16000, 6.0, 320).astype(np.float32)
for _ in range(2):
    enc.tokenize_arrays(wav, ref)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); g, s = enc.tokenize_arrays(wav, ref); s.cpu(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"prompt {secs:.1f} s -> {s.shape[1]} frames: tokenize wall {min(ts):.2f} ms")
rows = [enc.time_launch(i, 3) for i in range(enc.launches())]
tot = sum(r[1] for r in rows)
print(f"{len(rows)} launches, sum {tot:.2f} ms, {sum(r[2] for r in rows) / 1e9:.1f} GFLOP")
agg = {}
for n, ms, fl in rows:
    k = n.split(".")[-1] if n.startswith("w2v.encoder.layers") else n
    k = "w2v.layer." + k if n.startswith("w2v.encoder.layers") else (n if not n.startswith("encoder.") else "bicodec." + n.split(".")[-1])
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl
for k, (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{k:46s} x{cnt:3d} {ms:8.3f} ms  {fl / max(ms, 1e-9) / 1e9:8.2f} TF/s")
