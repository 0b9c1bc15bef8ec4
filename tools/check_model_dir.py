#!/usr/bin/env python3
"""HIP path vs the CPU oracle on ANY model directory in the published layout (LLM/{config.json, *.safetensors, tokenizer
files}, BiCodec/{config.yaml, model.safetensors}, config.yaml) -- the check a person with the real Spark-TTS-0.5B checkpoint
runs first (GPU box):

    python tools/check_model_dir.py <model_dir> [--tokens 32] [--text "..."]

Stands beside the reference's own loading and call sites: BiCodec.load_from_checkpoint (sparktts/models/bicodec.py:69-111),
SparkTTS.__init__ (cli/SparkTTS.py:46-51), generate (:197-204), detokenize (:231-234).  Reports, and exits non-zero on a FAIL:
  1. what rounding the bf16 weight arena applied to the checkpoint (exact for a bf16 checkpoint);
  2. teacher-forced logits of a fixed prompt: max |diff| to the oracle on the ROUNDED weights (bar 2e-3) and on the
     checkpoint's own weights (what bf16 rounding costs, for information);
  3. N greedy tokens, f32 KV cache (must equal the oracle's) and bf16 KV cache (equal, or first difference on a near-tie);
  4. one detokenize of those tokens: waveform max |diff| to the oracle vocoder (north_star bar 1e-3).
Without a directory argument a synthetic tiny model directory is written and checked (the self-test the GPU suite runs)."""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))

import numpy as np
import torch


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("model_dir", nargs="?")
    ap.add_argument("--tokens", type=int, default=32)
    ap.add_argument("--text", default="The quick brown fox jumps over the lazy dog near the quiet river")
    ap.add_argument("--max-positions", type=int, default=1024)
    a = ap.parse_args(argv)
    from pathlib import Path
    from oracle.bicodec_ref import BiCodecDetokRef
    from oracle.llm_ref import Qwen2Ref
    from sparkmi import synthetic, weights as W
    from sparkmi.arena import llm_cfg_struct, pack_llm_arena
    from sparkmi.bicodec import BiCodecVocoder
    from sparkmi.config import BiCodecConfig, LLMConfig
    from sparkmi.llm import SparkLLM
    from sparkmi.pipeline_text import build_clone_prompt

    d = a.model_dir
    if d is None:
        d = tempfile.mkdtemp(prefix="spark_check_")
        synthetic.make_model_dir(d)
        print(f"no directory given: wrote a synthetic tiny model to {d}")
    d = Path(d)
    fails = []

    def verdict(name, ok, detail):
        print(f"[{'ok' if ok else 'FAIL'}] {name}: {detail}", flush=True)
        if not ok:
            fails.append(name)

    # ---- 1. load, pack, rounding report
    t0 = time.time()
    lcfg = LLMConfig.from_json(d / "LLM" / "config.json")
    state = W.load_llm_state(d / "LLM")
    cs = llm_cfg_struct(lcfg, 1, a.max_positions, "f32", True)
    from sparkmi.arena import Bf16RoundingReport
    import warnings
    report = Bf16RoundingReport()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # the rounding is reported below, not warned about
        host = pack_llm_arena(lcfg, state, cs, report=report)
    arena = torch.from_numpy(host).to("cuda:0")
    print(f"LLM: {lcfg.num_hidden_layers} layers, hidden {lcfg.hidden_size}, vocab {lcfg.vocab_size}; packed in {time.time() - t0:.1f}s; "
          f"bf16 rounding of the checkpoint: {vars(report)}", flush=True)

    # ---- 2. teacher-forced logits of a fixed prompt
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(str(d / "LLM"))
    rng = np.random.Generator(np.random.PCG64(1234))
    glob = rng.integers(0, 4096, size=32).tolist()
    prompt = build_clone_prompt(a.text, glob, [], None)
    ids = tok([prompt], return_tensors="pt").input_ids[0].tolist()
    rounded = {k: (torch.as_tensor(np.asarray(v)).to(torch.bfloat16).to(torch.float32).numpy() if np.asarray(v).ndim == 2 else np.asarray(v))
               for k, v in state.items()}
    ref = Qwen2Ref(lcfg, rounded, kv_dtype="f32")
    llm = SparkLLM(lcfg, None, "cuda:0", max_positions=a.max_positions, kv_dtype="f32", arena=arena)
    got = llm.forward_logits(ids).cpu().numpy()
    want = ref.forward(np.asarray(ids)).numpy()
    err = float(np.abs(got - want).max())
    verdict("logits vs oracle (bf16-rounded weights)", err < 2e-3 and bool((got.argmax(-1) == want.argmax(-1)).all()), f"max |diff| {err:.2e} over {len(ids)} positions (bar 2e-3), arg-max equal at every position")
    raw = Qwen2Ref(lcfg, state, kv_dtype="f32").forward(np.asarray(ids)).numpy()
    print(f"      for information: vs the checkpoint's unrounded weights max |diff| {float(np.abs(got - raw).max()):.2e}, greedy agreement {float((got.argmax(-1) == raw.argmax(-1)).mean()):.3f}")

    # ---- 3. greedy tokens
    n = a.tokens
    toks = llm.generate_ids([ids], n)[0]
    wtoks = ref.generate_greedy(ids, n)
    verdict(f"{n} greedy tokens, f32 KV", toks == wtoks, "identical to the oracle" if toks == wtoks else f"first difference at token {next(i for i in range(n) if toks[i] != wtoks[i])}")
    llm16 = SparkLLM(lcfg, None, "cuda:0", max_positions=a.max_positions, kv_dtype="bf16", arena=arena)
    t16 = llm16.generate_ids([ids], n)[0]
    ref.kv_dtype = "bf16"
    w16 = ref.generate_greedy(ids, n)
    if t16 == w16:
        verdict(f"{n} greedy tokens, bf16 KV (the default)", True, "identical to the oracle's bf16-KV emulation")
    else:
        i = next(k for k in range(n) if t16[k] != w16[k])
        lg = torch.from_numpy(got[-1]) if i == 0 else llm.forward_logits(list(ids) + list(w16[:i]))[-1].float().cpu()
        top = torch.topk(lg, 2)
        tie = {int(top.indices[0]), int(top.indices[1])} == {t16[i], w16[i]} and float(top.values[0] - top.values[1]) < 2e-2
        verdict(f"{n} greedy tokens, bf16 KV (the default)", tie, f"first difference at token {i}: top-2 logit gap {float(top.values[0] - top.values[1]):.2e} ({'a near-tie: legitimate' if tie else 'NOT a tie'})")

    # ---- 4. detokenize
    vcfg = BiCodecConfig.from_yaml(d / "BiCodec" / "config.yaml")
    sd = W.load_bicodec_state(d / "BiCodec")
    sem = torch.tensor([[t % vcfg.codebook_size for t in toks]])
    g = torch.tensor([[glob[: vcfg.spk_token_num]]])
    voc = BiCodecVocoder(vcfg, sd, "cuda:0", max_batch=1, max_frames=max(n, 8))
    wav = voc.detokenize(sem, g).cpu().numpy().reshape(-1)
    owav = BiCodecDetokRef(vcfg, W.fold_weight_norm(sd)).detokenize(sem, g).numpy().reshape(-1)
    e = float(np.abs(wav - owav).max())
    verdict("detokenize vs oracle vocoder", wav.shape == owav.shape and e < 1e-3, f"{wav.size} samples, max |diff| {e:.2e} (north_star bar 1e-3)")
    print("RESULT:", "all checks passed" if not fails else f"FAILED: {fails}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
