#!/usr/bin/env python3
"""Launch-floor microbenchmarks on the GPU box: python tools/ubench.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import torch
from sparkmi import _lib
l = _lib.lib()
f = l.smi_ubench_chain
f.restype = C.c_int
f.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
buf = torch.zeros(1 << 28, dtype=torch.uint8, device="cuda")      # 256 MiB
scr = torch.zeros(64, dtype=torch.float32, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(name, kind, grid, block, lds=0, loads=0, n=120, iters=50):
    us = C.c_float(0)
    rc = f(kind, grid, block, lds, loads, n, iters, C.c_void_p(buf.data_ptr()), C.c_void_p(scr.data_ptr()), C.byref(us), st)
    assert rc == 0, l.smi_last_error()
    mb = grid * block * loads * 16 / 1e6
    print(f"{name:46s} grid {grid:5d} x {block:4d}  {us.value:7.2f} us/kernel" + (f"  {mb:7.2f} MB  {mb / us.value / 1e3 if loads else 0:6.2f} TB/s" if loads else ""))
run("empty, no args", 0, 1, 64)
run("empty, no args", 0, 256, 256)
run("empty, no args", 0, 2048, 256)
run("200B arg + dependent write", 1, 1, 64)
run("200B arg + dependent write", 1, 72, 256)
run("200B arg + dependent write", 1, 304, 256)
run("200B arg + dependent write", 1, 56, 1024)
run("+ 30 KB dyn LDS + barrier", 2, 72, 256, lds=30720)
run("+ 60 KB dyn LDS + barrier", 2, 304, 256, lds=61440)
for grid, block, loads in [(72, 256, 7), (56, 256, 7), (304, 256, 14), (608, 256, 7), (56, 1024, 10), (224, 256, 10), (256, 1024, 4), (2594, 256, 28), (1024, 256, 71), (512, 512, 71)]:
    run("stream loads/thread=%d" % loads, 3, grid, block, loads=loads)
