#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes: python tools/pmc_table.py <FETCH_SIZE dir> <WRITE_SIZE dir>"""
import csv, glob, re, sys
from collections import defaultdict


def load(d):
    out = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            key = (name, r.get("Grid_Size", ""))
            out[key][0] += 1
            out[key][1] += float(r["Counter_Value"])
    return out


fetch, write = load(sys.argv[1]), load(sys.argv[2])
print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), MI355X gfx950; per-launch averages")
print("# FETCH_SIZE is in KiB and, on gfx950, half the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM): read MB = 2 * KiB * 1024 / 1e6")
print(f"{'kernel':58s} {'grid':>8s} {'n':>5s} {'FETCH KiB':>11s} {'read MB (x2)':>13s} {'WRITE KiB':>10s}")
for key, (n, tot) in sorted(fetch.items(), key=lambda kv: -kv[1][1]):
    w = write.get(key, [1, 0.0])
    print(f"{key[0][:58]:58s} {key[1]:>8s} {n:5d} {tot / n:11.1f} {2 * tot / n * 1024 / 1e6:13.2f} {w[1] / max(w[0], 1):10.1f}")
