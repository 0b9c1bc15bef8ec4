"""Host-side AddressSanitizer run of the C-ABI library ("race detection / sanitizers"; GPU ASAN is not available on
this pool, so the sanitizer sees the host code on the CPU box): `make asan` builds libsparkmi_asan.so with the host half
instrumented; a child process preloads clang's ASAN runtime and drives every entry point that needs no device -- arena
layouts, the vocoder / encoder arena tables, the one-row engine's plan builder (vectors of vectors, the most intricate host
code in the library), argument validation and the create / destroy error paths (no GPU here: create fails after its host-side
set-up and must unwind cleanly)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spark-tts_amd", "csrc")
LIB = os.path.join(ROOT, "spark-tts_amd", "sparkmi", "asan", "libsparkmi_asan.so")

CHILD = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.environ["SMI_ROOT"], "spark-tts_amd"))
from sparkmi import _lib, config as Cf
from sparkmi.arena import llm_cfg_struct
l = _lib.diag()                                  # SPARKMI_LIB points at the ASAN build (compiled -DSMI_DIAG: the engine's plan builder is in it)
assert l.smi_version() == _lib.ABI_VERSION
for cfg in (Cf.tiny_llm(), Cf.spark_0p5b_llm()):
    cs = llm_cfg_struct(cfg, 4, 512, "bf16", True)
    assert l.smi_llm_arena_bytes(C.byref(cs)) > 0
    off, n = C.c_size_t(), C.c_size_t()
    for sec in range(11):
        for layer in ((0,) if sec >= 7 else range(cfg.num_hidden_layers)):
            assert l.smi_llm_arena_section(C.byref(cs), sec, layer, C.byref(off), C.byref(n)) == 0
    assert l.smi_llm_arena_section(C.byref(cs), 99, 0, C.byref(off), C.byref(n)) != 0      # error path + message
    assert b"section" in l.smi_last_error()
    st = (C.c_int32 * 8)()
    for ncu in (16, 64, 256, 304, 1024, 5000):
        l.smi_llm_engine_plan(C.byref(cs), ncu, st)                                         # any verdict, no heap error
    bad = llm_cfg_struct(cfg, 4, 512, "bf16", True); bad.head_dim = 48
    assert l.smi_llm_arena_bytes(C.byref(bad)) == 0
    h = C.c_void_p()
    rc = l.smi_llm_create(C.byref(cs), C.c_void_p(256), 1 << 40, C.byref(h))               # no device here: fails, unwinds
    assert rc != 0 and not h.value
vc = _lib.VocCfg()
from sparkmi.bicodec import voc_cfg_struct
vs = voc_cfg_struct(Cf.tiny_bicodec(), 2, 64)
nent = l.smi_voc_arena_count(C.byref(vs))
name = C.create_string_buffer(8192); info = (C.c_int32 * 6)()
off, n = C.c_size_t(), C.c_size_t()
for i in range(nent):
    assert l.smi_voc_arena_entry(C.byref(vs), i, name, 8192, C.byref(off), C.byref(n), info) == 0
assert l.smi_voc_arena_entry(C.byref(vs), nent + 5, name, 8192, C.byref(off), C.byref(n), info) != 0   # past the table
l.smi_voc_arena_entry(C.byref(vs), 0, name, 2, C.byref(off), C.byref(n), info)                        # tiny name buffer: any verdict, no overrun
for kind, kw in ((0, dict(C=24, dil=3)), (1, dict(C=32, Cout=16, K=11, S=5)), (2, dict(C=48, I=112, cond_dim=20)), (2, dict(C=48, I=112)),
                 (1, dict(C=32, Cout=16, K=9, S=4))):                                                  # the last one is outside the contract
    bc = _lib.VocBlockCfg(kind=kind, **kw)
    nb_ = l.smi_voc_block_arena_count(C.byref(bc))
    assert (nb_ > 0) == (kw.get("K") != 9) and (l.smi_voc_block_arena_bytes(C.byref(bc)) > 0) == (nb_ > 0)
    for i in range(nb_ + 1):
        rc = l.smi_voc_block_arena_entry(C.byref(bc), i, name, 8192, C.byref(off), C.byref(n), info)
        assert (rc == 0) == (i < nb_)
    assert l.smi_voc_block_run(C.byref(bc), C.c_void_p(256), 0, None, None, None, None, 1, 8, None, None) != 0   # validation only
print("asan child ok")
'''


def _runtime():
    c = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return c[-1] if c else None


@pytest.mark.timeout(1500)
def test_host_code_is_clean_under_address_sanitizer():
    rt = _runtime()
    if rt is None:
        pytest.skip("clang's ASAN runtime is not in this image")
    # make decides whether the instrumented library is stale (sources or headers newer than it)
    r = subprocess.run(["make", "-j4", "-C", CSRC, "asan"], capture_output=True, text=True, timeout=1400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=rt, SPARKMI_LIB=LIB, SMI_ROOT=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23:protect_shadow_gap=0")
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "asan child ok" in r.stdout, (r.returncode, r.stdout[-1000:], r.stderr[-3000:])
