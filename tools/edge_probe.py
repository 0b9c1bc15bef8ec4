#!/usr/bin/env python3
"""Hand-off skeleton of the one-row decode engine (GPU box): what do the five all-to-all edges of a decode layer cost
inside ONE persistent launch, with granule hand-offs?  (csrc/diag/diag_edge.hip)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import numpy as np
import torch
from sparkmi import _lib
_lib.lib()
l = C.CDLL(str(_lib.LIB_PATH.with_name("libsparkmi_diag.so")))
l.smi_last_error.restype = C.c_char_p
f = l.smi_diag_edge
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_uint),
              C.c_void_p, C.c_void_p]
big = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(name, layers=24, launches=20, dma_kib=0):
    us = C.c_float(0)
    err = (C.c_uint * 4)()
    stamps = np.zeros((2, layers, 8), dtype=np.uint64)
    rc = f(layers, launches, dma_kib, C.c_void_p(big.data_ptr()), big.numel(), 50.0, C.byref(us), err,
           stamps.ctypes.data_as(C.c_void_p), st)
    assert rc == 0, l.smi_last_error()
    s = stamps.astype(np.float64) * 0.01   # us
    # CU 0 (a down CU): A, C, D, E stamps; head CU: A, B, D
    d0 = s[0]
    seg = {"C-A": d0[:, 2] - d0[:, 0], "D-C": d0[:, 3] - d0[:, 2], "E-D": d0[:, 4] - d0[:, 3], "A'-E": np.append(d0[1:, 0] - d0[:-1, 4], np.nan)}
    h = s[1]
    segh = {"B-A": h[:, 1] - h[:, 0]}
    txt = "  ".join(f"{k} {np.nanmean(v[2:]):.2f}" for k, v in {**seg, **segh}.items())
    print(f"{name:40s} layers {layers:3d} dma {dma_kib:2d} KiB/wave/layer: {us.value:8.1f} us/launch  {us.value / layers:6.2f} us/layer   "
          f"timeout {err[0]} where {err[1]} wrong {err[3]}   [{txt}]", flush=True)
    return err[0]


if run("tiny: 2 layers", layers=2, launches=2):
    sys.exit("hand-off skeleton timed out on the smallest case")
run("edges only", 24, 20, 0)
run("edges only, 96 layers", 96, 10, 0)
run("edges + 4 KiB/wave/layer stream", 24, 20, 4)
run("edges + 15 KiB/wave/layer stream (decode layer)", 24, 20, 15)
run("edges + 15 KiB, 96 layers", 96, 10, 15)
