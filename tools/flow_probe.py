#!/usr/bin/env python3
"""Dataflow-chain microbenchmark (GPU box): what does a phase hand-off cost INSIDE one launch when the blocks of later
phases are already resident with their weights in registers?  (csrc/diag/diag_flow.hip)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import torch
from sparkmi import _lib
_lib.lib()
l = C.CDLL(str(_lib.LIB_PATH.with_name("libsparkmi_diag.so")))
l.smi_last_error.restype = C.c_char_p
f = l.smi_diag_flow
f.restype = C.c_int
f.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
              C.c_uint, C.POINTER(C.c_float), C.POINTER(C.c_uint), C.c_void_p]
big = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(name, sizes, loads, nlayers=24, block=1024, reuse=0, launches=20, limit=200000):
    n = len(sizes)
    us = C.c_float(0)
    err = (C.c_uint * 4)()
    rc = f((C.c_int * n)(*sizes), (C.c_int * n)(*loads), n, nlayers, block, reuse, launches, C.c_void_p(big.data_ptr()), big.numel(),
           limit, C.byref(us), err, st)
    assert rc == 0, l.smi_last_error()
    phases = n * nlayers
    mb = nlayers * sum(s * ld for s, ld in zip(sizes, loads)) * block * 16 / 1e6
    print(f"{name:44s} block {block:4d} reuse {reuse}: {us.value:8.1f} us/launch  {us.value / phases:6.2f} us/phase  "
          f"{mb:7.1f} MB -> {mb / us.value / 1e3:5.2f} TB/s   timeout {err[0]} stale {err[1]} max_polls {err[2]}", flush=True)
    return err[0]


# the decode layer's shape (1024-thread blocks): QKV 72, attention 7, o_proj 112, gate_up 304, down 224
pat = [72, 7, 112, 304, 224]
if run("tiny: 2 layers, no weights", pat, [0] * 5, nlayers=2, launches=3):
    sys.exit("dataflow chain timed out on the smallest case: blocks are not dispatched in index order?")
run("decode-layer pattern, no weights", pat, [0] * 5)
run("decode-layer pattern, weight-sized loads", pat, [2, 0, 1, 4, 3])
run("decode-layer pattern, weight-sized loads", pat, [2, 0, 1, 4, 3], reuse=1)
run("uniform 256-block phases, no weights", [256], [0], nlayers=120)
run("uniform 256-block phases, 2 loads", [256], [2], nlayers=120)
run("uniform 64-block phases, no weights", [64], [0], nlayers=120)
run("uniform 8-block phases, no weights", [8], [0], nlayers=120)
run("512-thread blocks, decode pattern x2 blocks", [144, 14, 224, 608, 448], [2, 0, 1, 4, 3], block=512)
