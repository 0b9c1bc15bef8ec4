#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes:
    python tools/pmc_table.py <FETCH_SIZE dir> <WRITE_SIZE dir> [--json out.json --batch B --tokens T]
The table goes to stdout; --json also writes the decode-step kernels' HBM bytes per launch keyed by role
(qkv / attn / o_proj / gate_up / down / lm_head) together with the hash of the HIP sources the library was
built from -- bench.py reads `roofline.traffic` from that file when the hash matches its own build."""
import argparse, csv, glob, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(d):
    out = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            key = (name, r.get("Grid_Size", ""))
            out[key][0] += 1
            out[key][1] += float(r["Counter_Value"])
    return out


ap = argparse.ArgumentParser()
ap.add_argument("fetch")
ap.add_argument("write")
ap.add_argument("--json")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--tokens", type=int, default=12)
a = ap.parse_args()
fetch, write = load(a.fetch), load(a.write)
print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), MI355X gfx950; per-launch averages")
print("# FETCH_SIZE is in KiB and, on gfx950, half the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM): read MB = 2 * KiB * 1024 / 1e6")
print(f"{'kernel':58s} {'grid':>8s} {'n':>5s} {'FETCH KiB':>11s} {'read MB (x2)':>13s} {'WRITE KiB':>10s}")
rows = []
for key, (n, tot) in sorted(fetch.items(), key=lambda kv: -kv[1][1]):
    w = write.get(key, [1, 0.0])
    rd, wr = 2 * tot / n * 1024, w[1] / max(w[0], 1) * 1024
    rows.append((key[0], key[1], n, rd, wr))
    print(f"{key[0][:58]:58s} {key[1]:>8s} {n:5d} {tot / n:11.1f} {rd / 1e6:13.2f} {wr / 1024:10.1f}")

if a.json:
    from bench import build_hash
    # decode-step kernels run once per layer per decode step: 24 x (tokens - 1) launches or more (the first step adds one
    # per layer at batch <= 64); template argument 7 of k_gemm is the epilogue (0 residual, 1 SwiGLU, 2 QKV)
    steps = a.tokens - 1
    roles = {}
    resid = []
    comb = None
    for name, grid, n, rd, wr in rows:
        if n < 24 * steps and not name.startswith("k_lm") and not name.startswith("k_engine"):
            continue
        ent = {"kernel": name, "grid": grid, "launches": n, "read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
        if name.startswith("k_engine") and n >= steps:
            roles.setdefault("engine", ent)       # all layers of a one-row step in one launch (csrc/smi_eng.h)
        elif name.startswith("k_lm") and n >= steps:
            roles.setdefault("lm_head", ent)
        elif name.startswith("k_attn<"):
            roles.setdefault("attn", ent)
        elif name.startswith("k_down1<") or name.startswith("k_downS<"):
            roles.setdefault("down", ent)         # down_proj at 1 .. 6 rows (four chains per wave)
        elif name.startswith("k_downC<"):
            roles.setdefault("down", ent)         # 7 .. 64 rows: one chain per block ...
        elif name.startswith("k_resid_comb<"):
            comb = ent                            # ... + the in-order combine: a second launch of the same operation
        elif name.startswith("k_gemm<"):
            args = [x.strip() for x in name[name.index("<") + 1: name.rindex(">")].split(",")]
            epi = int(args[6])
            if epi == 2:
                roles.setdefault("qkv", ent)
            elif epi == 1:
                roles.setdefault("gate_up", ent)
            elif epi == 0:
                resid.append(ent)
    if comb is not None and roles.get("down", {}).get("kernel", "").startswith("k_downC<"):
        d = roles["down"]
        roles["down"] = {"kernel": d["kernel"] + " + " + comb["kernel"], "grid": d["grid"], "launches": d["launches"],
                         "read_bytes": d["read_bytes"] + comb["read_bytes"], "write_bytes": d["write_bytes"] + comb["write_bytes"],
                         "hbm_bytes_per_launch": d["hbm_bytes_per_launch"] + comb["hbm_bytes_per_launch"], "launches_per_operation": 2}
    resid.sort(key=lambda e: -e["read_bytes"])
    if "down" in roles:                           # the residual GEMMs left are o_proj's (none at one row: fused into attention)
        if resid:
            roles["o_proj"] = resid[0]
    else:
        if resid:
            roles["down"] = resid[0]
        if len(resid) > 1:
            roles["o_proj"] = resid[1]
    json.dump({"build": build_hash(), "batch": a.batch, "tokens": a.tokens,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only, eager launches; "
                         "read bytes = 2 x FETCH_SIZE KiB x 1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B), write bytes = WRITE_SIZE KiB x 1024; "
                         "a producer's helper-block prefetch for a later kernel is part of ITS launch's traffic",
               "kernels": roles}, open(a.json, "w"), indent=1)
    print(f"# wrote {a.json}: {sorted(roles)}")
