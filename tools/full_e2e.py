#!/usr/bin/env python3
"""Full-size end-to-end check (GPU box): a synthetic Spark-TTS-0.5B model directory (0.5B LLM, full BiCodec, wav2vec2-large
shape) is written to disk and loaded through the drop-in class; controllable TTS, voice cloning from a wav file, streaming and
the in-flight batcher all run once.  python tools/full_e2e.py"""
import os, sys, tempfile, time, wave
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import numpy as np, torch
from sparkmi import config as C, config_tok as T, synthetic
from sparkmi.pipeline import SparkTTS
from sparkmi.streaming import crossfade

d = tempfile.mkdtemp(prefix="spark_full_")
t0 = time.time()
synthetic.make_model_dir(d, llm_cfg=C.spark_0p5b_llm(), voc_cfg=C.spark_0p5b_bicodec(), w2v_cfg=T.xlsr53(), tok_cfg=T.spark_0p5b_tok())
print(f"model dir written in {time.time() - t0:.1f}s: {sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(d) for f in fs) / 1e9:.2f} GB", flush=True)
t0 = time.time()
tts = SparkTTS(d, torch.device("cuda:0"), max_batch=4, max_positions=2048, max_frames=1200)
tts._eos = None     # random weights: run the token budget instead of stopping at a chance eos
print(f"loaded in {time.time() - t0:.1f}s", flush=True)
# a prompt wav (3 s)
t = np.arange(48000) / 16000.0
x = 0.3 * np.sin(2 * np.pi * (120 + 25 * np.sin(2 * np.pi * 0.9 * t)) * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 2.1 * t))
wp = os.path.join(d, "prompt.wav")
with wave.open(wp, "wb") as w:
    w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes((x * 32767).astype("<i2").tobytes())
torch.cuda.synchronize(); t0 = time.time()
g, s = tts.audio_tokenizer.tokenize(wp)
torch.cuda.synchronize()
print(f"tokenize (first call builds the 1 GB encoder arena): {time.time() - t0:.1f}s -> {s.shape[1]} semantic, {g.numel()} global tokens", flush=True)
for name, kw in (("clone", dict(text="A sentence to be spoken in the prompt's voice.", prompt_speech_path=wp, prompt_text="hello there")),):
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        try:
            wav = tts.inference(**kw, do_sample=False, max_new_tokens=200)
            print(f"{name}: {len(wav)} samples ({len(wav) / 16000:.2f} s audio) in {(time.time() - t0) * 1e3:.1f} ms, |wav|max {np.abs(wav).max():.3f}", flush=True)
        except ValueError as e:
            print(f"{name}: {e}", flush=True)
chunks = []
torch.cuda.synchronize(); t0 = time.time(); first = None
try:
    for c in tts.inference_stream(text="Streaming synthesis.", prompt_speech_path=wp, prompt_text="hi", do_sample=True, seed=7, max_new_tokens=300):
        first = first or (time.time() - t0)
        chunks.append(c)
    print(f"stream: {len(chunks)} chunks, first after {first * 1e3:.1f} ms, total {(time.time() - t0) * 1e3:.1f} ms, joined {len(crossfade(chunks, 1600))} samples", flush=True)
except ValueError as e:
    print("stream:", e, flush=True)
reqs = [dict(text=f"request {i} " * (1 + i % 3), prompt_tokens=(g, s[:, : 40 + 10 * i]), prompt_text=None) for i in range(6)]
torch.cuda.synchronize(); t0 = time.time(); n = 0
try:
    for i, wav in tts.serve(reqs, do_sample=False, max_new_tokens=120):
        n += 1
    print(f"serve: {n} of {len(reqs)} requests finished in {(time.time() - t0) * 1e3:.1f} ms (4 live at a time)", flush=True)
except ValueError as e:
    print("serve:", e, flush=True)
print("full e2e done")
