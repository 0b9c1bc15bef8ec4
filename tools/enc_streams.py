#!/usr/bin/env python3
"""Do N prompt encodes overlap when each runs on its own HIP stream (own handle = own scratch, shared arena)?  (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import numpy as np, torch
from sparkmi import config as C, config_tok as T, weights as W
from sparkmi.encoder import BiCodecEncoder, get_ref_clip
wcfg, tcfg, vcfg = T.xlsr53(), T.spark_0p5b_tok(), C.spark_0p5b_bicodec()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
e0 = BiCodecEncoder(wcfg, tcfg, W.fold_pos_conv_weight_norm(W.wav2vec2_state(wcfg)),
                    W.fold_weight_norm(W.bicodec_tok_state(tcfg, vcfg.vq_input_dim)), "cuda:0", max_seconds=6.0)
encs = [e0] + [BiCodecEncoder(wcfg, tcfg, None, None, "cuda:0", max_seconds=6.0, arena=e0.arena) for _ in range(N - 1)]
rng = np.random.default_rng(0)
wavs = [(0.1 * rng.standard_normal(96000)).astype(np.float32) for _ in range(N)]
refs = [get_ref_clip(w, 16000, 6.0, 320).astype(np.float32) for w in wavs]
streams = [torch.cuda.Stream() for _ in range(N)]

def run(nstreams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    for i in range(N):
        with torch.cuda.stream(streams[i % nstreams]):
            outs.append(encs[i].tokenize_arrays(wavs[i], refs[i]))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, outs
for ns in (1, 1, 2, 4, 8, 1, 8):
    ms, outs = run(ns)
    print(f"{N} encodes on {ns} stream(s): {ms:.2f} ms", flush=True)
base = run(1)[1]
par = run(8)[1]
print("same tokens:", all(torch.equal(a[1], b[1]) and torch.equal(a[0], b[0]) for a, b in zip(base, par)))
