#!/bin/bash
# Timing-only builds of the ring k_pgemm with parts removed (SMI_PG_ABL bits: 1 MFMAs, 2 in-loop LDS-DMA, 4 epilogue, 8 fragment reads):
# (bit 2 without bit 8 multiplies never-written LDS -- timing only -- and one such run did not finish within gpurun's silence limit:
#  combine it with 8, as in 11 and 15)
#   tools/pg_ablate.sh build 1 2 3 ...   (here: builds spark-tts_amd/sparkmi/ab/libsparkmi_abl<bits>.so)
#   tools/pg_ablate.sh run BxP 1 2 3 ... (GPU box: per-kernel averages of each build)
cd "$(dirname "$0")/.." || exit 1
mode=$1; shift
if [ "$mode" = build ]; then
  cd spark-tts_amd/csrc && mkdir -p ab ../sparkmi/ab
  for b in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -Wno-unused-function -ffp-contract=off -DSMI_PG_ABL=$b -c smi_llm.hip -o ab/smi_llm_abl$b.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../sparkmi/ab/libsparkmi_abl$b.so ab/smi_llm_abl$b.o $(ls smi_*.o | grep -v smi_llm.o) || exit 1
  done
else
  pf=$1; shift
  for b in "$@"; do
    lib=$PWD/spark-tts_amd/sparkmi/ab/libsparkmi_abl$b.so
    [ "$b" = 0 ] && lib=$PWD/spark-tts_amd/sparkmi/libsparkmi.so
    SPARKMI_LIB=$lib SPARKMI_PGEMM_MIN_ROWS=0 PF_ITERS=3 timeout -k 10 120 tools/prefill_prof.sh abl$b $pf > /dev/null 2>&1 || echo "abl$b: failed or timed out"
    python3 - gpurun_out/pfprof_abl${b}_$pf.csv abl$b <<'PY'
import csv, sys, os
if os.path.exists(sys.argv[1]):
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_pgemm" in r["Name"] or "k_attn_pf" in r["Name"]]
    print(sys.argv[2], "  ".join(r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] + " %.1f" % (float(r["AverageNs"]) / 1e3) for r in rows), flush=True)
PY
  done
fi
