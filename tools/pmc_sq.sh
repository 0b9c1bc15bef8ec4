# Where the waves of the vocoder's conv kernels spend their cycles (GPU box): bash tools/pmc_sq.sh <tag>
# one pass, SQ counters only (+ --kernel-trace), eager launches, vocoder alone at batch 32 x 150 frames
set -e
cd $GRAFT_REPO_ROOT
tag=${1:-x}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_sq -o r -- python3 tools/voc_profile.py 32 > gpurun_out/pmc_${tag}_sq.out 2> gpurun_out/pmc_${tag}_sq.log || { tail -5 gpurun_out/pmc_${tag}_sq.log; exit 1; }
python3 - <<PY | tee gpurun_out/pmc_${tag}_sq_table.txt
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for f in glob.glob("gpurun_out/pmc_${tag}_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[name] += 1
print("# per kernel: share of wave cycles parked (s_waitcnt / barrier), issue-stalled, issuing; of the issuing cycles: VALU / LDS / VMEM; VALU instructions per launch")
for name, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:14]:
    w = c["SQ_WAVE_CYCLES"] or 1.0
    print("%-34s n=%4d  wait %5.1f%%  stall %5.1f%%  active %5.1f%%  (valu %5.1f%% lds %5.1f%% vmem %5.1f%%)  valu insts/launch %.3g" % (
        name[:34], cnt[name], 100 * c["SQ_WAIT_ANY"] / w, 100 * c["SQ_WAIT_INST_ANY"] / w, 100 * c["SQ_ACTIVE_INST_ANY"] / w,
        100 * c["SQ_ACTIVE_INST_VALU"] / w, 100 * c["SQ_ACTIVE_INST_LDS"] / w, 100 * c["SQ_ACTIVE_INST_VMEM"] / w, c["SQ_INSTS_VALU"] / max(cnt[name], 1)))
PY
find gpurun_out/pmc_${tag}_sq -name "*.csv" -size +2M -delete
