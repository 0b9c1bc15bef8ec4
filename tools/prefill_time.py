#!/usr/bin/env python3
"""Prefill wall time by path (GPU box): python tools/prefill_time.py
   modes: grouped decode GEMMs (default below SPARKMI_PGEMM_MIN_ROWS), k_pgemm (MIN_ROWS=0), 32-row chunks."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(%r, "spark-tts_amd"))
import torch
from sparkmi import config as Cf, weights as W
from sparkmi.llm import SparkLLM
from sparkmi.arena import llm_cfg_struct, pack_llm_arena
cfg = Cf.spark_0p5b_llm()
arena = torch.from_numpy(pack_llm_arena(cfg, W.SyntheticLLM(cfg), llm_cfg_struct(cfg, 1, 1024, "bf16", True))).to("cuda:0")
res = []
sizes = [tuple(int(v) for v in x.split("x")) for x in os.environ.get("PF_SIZES", "1x128,1x64,1x400,4x128,8x128,16x128,32x128").split(",")]
for B, P in sizes:
    llm = SparkLLM(cfg, None, "cuda:0", max_slots=B, max_positions=1024, arena=arena, diag=True)
    prompts = [np.random.Generator(np.random.PCG64(1 + b)).integers(0, cfg.vocab_size, size=P).tolist() for b in range(B)]
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        llm.prefill(prompts); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    tok = llm.tokens(1)
    res.append((B, P, round(min(ts), 2), tok[0][0]))
print(os.environ.get("MODE"), res)
''' % ROOT
modes = (("grouped", {"SPARKMI_PGEMM_MIN_ROWS": "100000"}), ("pgemm", {"SPARKMI_PGEMM_MIN_ROWS": "0"}), ("chunks", {"SPARKMI_PREFILL_CHUNKS": "1"}))
if len(sys.argv) > 1 and sys.argv[1] == "pg2":   # k_pgemm with the 2 x 2 wave grid vs the 1 x 4 one (SPARKMI_TUNE2 bit 16384), forced at every size
    os.environ.setdefault("PF_SIZES", "1x128,1x460,4x128,8x460,32x128")
    modes = (("pgemm 2x2", {"SPARKMI_PGEMM_MIN_ROWS": "0"}), ("pgemm 1x4", {"SPARKMI_PGEMM_MIN_ROWS": "0", "SPARKMI_TUNE2": "16384"}), ("default", {}))
if len(sys.argv) > 1 and sys.argv[1] == "mix":   # which GEMMs go through the prefill GEMM at mid-size prompts
    os.environ.setdefault("PF_SIZES", "1x128,1x460,4x128,8x128")
    modes = (("default", {}), ("gate_up", {"SPARKMI_PGEMM_MIN_GU": "0"}), ("gate_up+qkv", {"SPARKMI_PGEMM_MIN_GU": "0", "SPARKMI_PGEMM_MIN_QKV": "0"}),
             ("gate_up+down", {"SPARKMI_PGEMM_MIN_GU": "0", "SPARKMI_PGEMM_MIN_D": "0"}), ("all", {"SPARKMI_PGEMM_MIN_ROWS": "0"}))
if len(sys.argv) > 1 and sys.argv[1] == "mix2":
    os.environ.setdefault("PF_SIZES", "2x128,12x128,16x128,24x128")
    gd = {"SPARKMI_PGEMM_MIN_GU": "0", "SPARKMI_PGEMM_MIN_D": "0"}
    modes = (("gate_up", {"SPARKMI_PGEMM_MIN_GU": "0"}), ("gate_up+down", gd), ("gate_up+down+o", dict(gd, SPARKMI_PGEMM_MIN_O="0")),
             ("gate_up+down+qkv", dict(gd, SPARKMI_PGEMM_MIN_QKV="0")), ("all", {"SPARKMI_PGEMM_MIN_ROWS": "0"}))
if len(sys.argv) > 1 and sys.argv[1] == "pf2":   # prefill attention: 16-row tiles on the matrix pipes vs one wave per (row, head)
    os.environ.setdefault("PF_SIZES", "1x128,1x460,8x460,32x128,1x1000")
    modes = (("attn tiles (k_attn_pf2)", {}), ("attn per row (k_attn_pf)", {"SPARKMI_ATTN_PF2": "0"}))
if len(sys.argv) > 1 and sys.argv[1] == "gu1":   # gate_up's one-batch shape beyond 32 rows (row-grouped prefill)
    modes = tuple((f"grouped gu1<={r}", {"SPARKMI_PGEMM_MIN_ROWS": "100000", "SPARKMI_GU1_ROWS": str(r)}) for r in (32, 128, 100000))
if len(sys.argv) > 1 and sys.argv[1] == "r04":   # round 4: kernels by each sequence's own length; few-row shapes + split-K up to 512 rows
    os.environ.setdefault("PF_SIZES", "1x128,1x66,1x256,1x460,4x128,8x460,32x128")
    modes = (("default (own length: prefill GEMM family, few-row shapes <= 512 rows)", {}), ("many-row shapes at any size", {"SPARKMI_PG_SPLIT_ROWS": "0"}),
             ("row-grouped decode GEMMs (round 3's choice below 288..1280 rows)", {"SPARKMI_PGEMM_MIN_ROWS": "100000"}), ("round-3 thresholds by total rows", {"SPARKMI_PGEMM_MIN_GU": "288", "SPARKMI_PGEMM_MIN_D": "896", "SPARKMI_PGEMM_MIN_QKV": "1280", "SPARKMI_PGEMM_MIN_O": "1280"}))
for mode, env in modes:
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MODE=mode, **env), capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-1500:], flush=True)
