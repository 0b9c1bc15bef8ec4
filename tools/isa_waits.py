#!/usr/bin/env python3
"""Memory-op / wait skeleton of one kernel's ISA, in program order (no GPU needed):
     python tools/isa_waits.py spark-tts_amd/csrc/smi_llm.hip 'k_gemm<1, 1, 8, 2, 2, 1, 1, 0, 1, 6, 2>' [first_line [last_line]]
   Prints every global/flat/scratch load and store, s_load, s_waitcnt, LDS write, barrier, branch label and MFMA with its
   line number inside the kernel, plus ScratchSize / VGPRs / code length.  What to look for (profiles/README.md):
   a `s_waitcnt vmcnt(0)` right behind a load inside a loop (one memory round trip per trip), `scratch_` / `flat_load`
   (an array or a `cond ? *p : 0` the compiler could not keep in registers), waits that sit in front of the MFMAs but
   belong to loads only the epilogue needs (vmcnt counts in order: every wait also covers everything issued before it)."""
import re, subprocess, sys, tempfile, os
src, want = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 9
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{root}/include",
                    f"-I{root}/spark-tts_amd/csrc", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out, src],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
names = {}
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        names[i] = m.group(1)
dem = subprocess.run(["c++filt"] + list(names.values()), capture_output=True, text=True).stdout.split("\n")
norm = lambda s: s.replace("(anonymous namespace)::", "").replace(" ", "")
start = next((i for (i, _), dn in zip(names.items(), dem) if norm(want) in norm(dn)), None)
if start is None:
    sys.exit("kernel not found; candidates:\n" + "\n".join(sorted(set(norm(x).split("(")[0] for x in dem if "k_" in x))))
pat = re.compile(r"global_|flat_|scratch_|s_load|s_waitcnt|ds_write|s_barrier|mfma|^\.LBB|s_endpgm")
n = 0
for l in lines[start + 1:]:
    n += 1
    if re.match(r"^_Z\w+:", l):
        break
    t = l.strip()
    if lo <= n <= hi and pat.search(t) and not t.startswith(";"):
        print(f"{n:5d}  {t.split(';')[0].strip()[:100]}")
    m = re.search(r"; (ScratchSize|NumVgprs|NumAgprs|codeLenInByte|Occupancy)\b.*", t)
    if m:
        print("      ", m.group(0))
        if "Occupancy" in m.group(0):
            break
