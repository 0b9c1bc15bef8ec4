# Counters of the vocoder's conv kernels at batch 32 x 150 frames, per (kernel, grid) (GPU box): bash tools/pmc_voc.sh <tag> [ENV=VAL]
# one rocprofv3 pass per counter group (--pmc + --kernel-trace only), eager launches
set -e
cd $GRAFT_REPO_ROOT
tag=${1:-x}
if [ -n "$2" ]; then export "$2"; fi
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmcvoc_${tag}_$i -o r -- python3 tools/voc_profile.py 32 > gpurun_out/pmcvoc_${tag}_$i.out 2> gpurun_out/pmcvoc_${tag}_$i.log || { tail -5 gpurun_out/pmcvoc_${tag}_$i.log; echo "group $i failed"; }
done
python3 - <<PY | tee gpurun_out/pmcvoc_${tag}_table.txt
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob("gpurun_out/pmcvoc_${tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        if not name.startswith(("k_conv", "k_resunit")): continue
        k = (name, r.get("Grid_Size", ""))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
print("# per launch averages; SQ_* wave counters are quad-cycles summed over waves")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    a = {n: v / cnt[k][n] for n, v in c.items()}
    w = a.get("SQ_WAVE_CYCLES", 0) or 1.0
    gui = a.get("GRBM_GUI_ACTIVE", 0) or 1.0
    hit, miss = a.get("TCC_HIT_sum", 0), a.get("TCC_MISS_sum", 0)
    print("%-28s grid %-9s n=%3d | wait %4.1f%% stall %4.1f%% active %4.1f%% (valu %4.1f lds %4.1f vmem %4.1f) | mfma busy %5.1f%% of %8.0f cycles (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs | L2 hit %4.1f%% (%.3g req) | lds conflict %4.1f%% of idx-active, %.3g lds insts" % (
        k[0][:28], k[1], max(cnt[k].values()), 100 * a.get("SQ_WAIT_ANY", 0) / w, 100 * a.get("SQ_WAIT_INST_ANY", 0) / w, 100 * a.get("SQ_ACTIVE_INST_ANY", 0) / w,
        100 * a.get("SQ_ACTIVE_INST_VALU", 0) / w, 100 * a.get("SQ_ACTIVE_INST_LDS", 0) / w, 100 * a.get("SQ_ACTIVE_INST_VMEM", 0) / w,
        100 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8 * 1024), gui / 8, 100 * hit / (hit + miss or 1), hit + miss,
        100 * a.get("SQ_LDS_BANK_CONFLICT", 0) / (a.get("SQ_LDS_IDX_ACTIVE", 0) or 1), a.get("SQ_INSTS_LDS", 0)))
PY
find gpurun_out -path "*pmcvoc_${tag}_*" -name "*.csv" -size +1M -delete
