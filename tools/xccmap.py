#!/usr/bin/env python3
"""How does the dispatcher map blocks to XCDs across consecutive kernels of a graph?  (GPU box)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import torch
from sparkmi import _lib
_lib.lib()   # torch + the HIP runtime first
l = C.CDLL(str(_lib.LIB_PATH.with_name('libsparkmi_diag.so')))   # diagnostics library (csrc/diag/)
l.smi_last_error.restype = C.c_char_p
f = l.smi_ubench_xccmap
f.restype = C.c_int
f.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
grids = [72, 14, 56, 608, 56]
blocks = [1024, 512, 512, 512, 1024]
n, reps, stride = len(grids), 4, 1024
out = torch.full((reps * n * stride,), 255, dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rc = f((C.c_int * n)(*grids), (C.c_int * n)(*blocks), n, reps, C.c_void_p(out.data_ptr()), stride, st)
assert rc == 0, l.smi_last_error()
o = out.cpu().numpy().reshape(reps * n, stride)
for k in range(reps * n):
    g = grids[k % n]
    row = o[k, :g]
    rr = all(int(row[b]) == (int(row[0]) + b) % 8 for b in range(g))
    print(f"kernel {k:2d} grid {g:4d}: first XCDs {row[:16].tolist()}  strict round-robin from {int(row[0])}: {rr}")
