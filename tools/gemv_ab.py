#!/usr/bin/env python3
"""One-row GEMV: VALU against the matrix pipe on the product's weight layouts (csrc/diag/diag_gemv.hip; GPU box):
    python tools/gemv_ab.py
Checks each variant's y against numpy on the same inputs, then times eager back-to-back launches that walk 48 weight copies
(more than the Infinity Cache holds), next to the product kernels' stand-alone probe of the same kind."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import numpy as np
import torch
from sparkmi import _lib, config as Cf, weights as W
from sparkmi.weights import bf16_bits_to_f32, f32_to_bf16_bits
_lib.lib()
l = C.CDLL(str(_lib.LIB_PATH.with_name("libsparkmi_diag.so")))
l.smi_last_error.restype = C.c_char_p
f = l.smi_diag_gemv
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_void_p]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = np.random.Generator(np.random.PCG64(5))
COPIES = 48

for shape, (name, N, K, parts) in enumerate([("down_proj 896 x 4864", 896, 4864, 4), ("qkv 1152 x 896", 1152, 896, 1)]):
    NT, KT = N // 16, K // 32
    wf = bf16_bits_to_f32(f32_to_bf16_bits((rng.standard_normal((N, K)) * 0.05).astype(np.float32))).reshape(N, K)
    x = rng.standard_normal(K).astype(np.float32)
    # tiles [NT][KT][64 pieces][8 bf16]: piece = k8 * 16 + n (plain) or (n >> 2) * 16 + k8 * 4 + (n & 3) (row-part-major, W_down)
    t = f32_to_bf16_bits(wf).reshape(NT, 16, KT, 4, 8).transpose(0, 2, 3, 1, 4)          # [nt][kt][k8][n][e]
    if parts == 4:
        t = t.reshape(NT, KT, 4, 4, 4, 8).transpose(0, 1, 3, 2, 4, 5)                     # [nt][kt][q][k8][r][e]
    one = np.ascontiguousarray(t).reshape(-1)
    Wd = torch.from_numpy(np.tile(one, COPIES).view(np.int16)).cuda()
    # triples [KT][3][4][8 bf16]
    hi = f32_to_bf16_bits(x); r1 = x - bf16_bits_to_f32(hi); mi = f32_to_bf16_bits(r1); lo = f32_to_bf16_bits(r1 - bf16_bits_to_f32(mi))
    xs = np.stack([hi.reshape(KT, 4, 8), mi.reshape(KT, 4, 8), lo.reshape(KT, 4, 8)], axis=1)
    XS = torch.from_numpy(np.ascontiguousarray(xs).view(np.int16)).cuda()
    X = torch.from_numpy(x).cuda()
    want = wf.astype(np.float64) @ x.astype(np.float64)
    print(f"--- {name}: {N * K * 2 / 1e6:.2f} MB of weights per launch, {COPIES} copies walked")
    for var, vname in enumerate(["MFMA, one per tile, hi/mid/lo columns (product arithmetic)", "VALU fp32 FMA on expanded bf16 weights (exact x)",
                                 "VALU v_dot2_f32_bf16, x rounded to bf16 (inexact bound)",
                                 "MFMA, four chains per wave: full 1 KiB load instructions"]):
        Y = torch.zeros(N, dtype=torch.float32, device="cuda")
        us = C.c_float(0)
        rc = f(var, shape, C.c_void_p(Wd.data_ptr()), COPIES, C.c_void_p(XS.data_ptr()), C.c_void_p(X.data_ptr()), C.c_void_p(Y.data_ptr()),
               COPIES * 20, C.byref(us), st)
        assert rc == 0, l.smi_last_error()
        err = float(np.abs(Y.cpu().numpy() - want).max())
        print(f"  {vname:62s} {us.value:7.2f} us/launch  {N * K * 2 / us.value / 1e6:6.2f} TB/s   max |y - f64| {err:.2e}", flush=True)

from sparkmi.llm import SparkLLM
cfg = Cf.spark_0p5b_llm()
llm = SparkLLM(cfg, W.SyntheticLLM(cfg), "cuda:0", max_positions=512, diag=True)
llm.prefill([rng.integers(0, cfg.vocab_size, size=128).tolist()]); llm.decode(8); torch.cuda.synchronize()
for k in ("down", "qkv"):
    print(f"product {k:5s} kernel, same kind of probe (eager back to back, walking the 24 layers): {llm.time_kernel(k, iters=480) * 1e3:.2f} us/launch")
