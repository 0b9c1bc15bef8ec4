#!/usr/bin/env python3
"""Launch-floor microbenchmarks on the GPU box: python tools/ubench.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import torch
from sparkmi import _lib
_lib.lib()   # torch + the HIP runtime first
l = C.CDLL(str(_lib.LIB_PATH.with_name('libsparkmi_diag.so')))   # diagnostics library (csrc/diag/)
l.smi_last_error.restype = C.c_char_p
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
f2 = l.smi_ubench_chain2
f2.restype = C.c_int
f2.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
big = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")      # 1 GiB > Infinity Cache
def run2(name, grid, block, loads, rotate, n=120, iters=20):
    us = C.c_float(0)
    nbytes = grid * block * loads * 16
    rot = (nbytes + 4095) // 4096 * 4096 if rotate else 0
    rc = f2(3, grid, block, 0, loads, n, iters, C.c_void_p(big.data_ptr()), big.numel(), rot, C.c_void_p(scr.data_ptr()), C.byref(us), st)
    assert rc == 0, l.smi_last_error()
    print(f"{name:8s} grid {grid:5d} x {block:4d} loads/thread {loads:3d}  {nbytes / 1e6:7.2f} MB  {us.value:7.2f} us/kernel  {nbytes / us.value / 1e6:6.2f} TB/s")
fx = l.smi_ubench_xcc
fx.restype = C.c_int
fx.argtypes = [C.c_int] * 6 + [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
NK = 100
ctr = torch.zeros((2 * NK + 2) * 8, dtype=torch.int32, device="cuda")
def runx(name, work, helpers, block, loads):
    us = C.c_float(0)
    rc = fx(work, helpers, block, loads, NK, 20, C.c_void_p(big.data_ptr()), big.numel(), C.c_void_p(ctr.data_ptr()), C.c_void_p(scr.data_ptr()), C.byref(us), st)
    assert rc == 0, l.smi_last_error()
    nbytes = work * block * loads * 16
    print(f"{name:34s} work {work:4d} helpers {helpers:4d} x {block:4d} loads {loads:3d} {nbytes / 1e6:7.2f} MB {us.value:7.2f} us/kernel {nbytes / us.value / 1e6:6.2f} TB/s")
run("empty", 0, 256, 256)
run2("warm", 608, 256, 7, False); run2("cold", 608, 256, 7, True)
runx("xcc-affine, no prefetch (cold)", 608, 0, 256, 7)
runx("xcc-affine + helpers prefetch next", 608, 64, 256, 7)
runx("xcc-affine + helpers prefetch next", 608, 128, 256, 7)
runx("xcc-affine + helpers prefetch next", 608, 256, 256, 7)
runx("xcc-affine, no prefetch (cold)", 72, 0, 256, 7)
runx("xcc-affine + helpers prefetch next", 72, 64, 256, 7)
runx("xcc-affine + helpers prefetch next", 72, 184, 256, 7)
runx("xcc-affine, no prefetch (cold)", 304, 0, 256, 14)
runx("xcc-affine + helpers prefetch next", 304, 208, 256, 14)
runx("xcc-affine, no prefetch (cold)", 56, 0, 1024, 10)
runx("xcc-affine + helpers prefetch next", 56, 200, 1024, 10)
