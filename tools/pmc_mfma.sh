# MFMA utilisation per kernel (GPU box): bash tools/pmc_mfma.sh <tag>
# one pass each, --pmc MfmaUtil (+ --kernel-trace only), eager launches: (a) the vocoder alone at batch 32 x 150 frames
# (tools/voc_profile.py 32), (b) a batch-32 bench step with 12 new tokens (prefill GEMM + decode kernels)
set -e
cd $GRAFT_REPO_ROOT
tag=${1:-x}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_mfma_voc -o r -- python3 tools/voc_profile.py 32 > gpurun_out/pmc_${tag}_mfma_voc.out 2> gpurun_out/pmc_${tag}_mfma_voc.log
timeout -k 10 500 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_mfma_llm -o r -- python3 bench.py --batch 32 --steps 1 --warmup 0 --new-tokens 12 --no-cpu-baseline --no-probes --no-graph > gpurun_out/pmc_${tag}_mfma_llm.json 2> gpurun_out/pmc_${tag}_mfma_llm.log
python3 - <<PY > gpurun_out/pmc_${tag}_mfma_table.txt
import csv, glob, re
from collections import defaultdict
print("# rocprofv3 --pmc MfmaUtil (= SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs) * 100): per-launch averages, eager launches")
for part, title in (("voc", "vocoder alone, batch 32 x 150 frames (tools/voc_profile.py 32)"), ("llm", "bench.py --batch 32 --new-tokens 12 --no-graph: prefill GEMM (k_pgemm), decode GEMVs, lm_head")):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob("gpurun_out/pmc_${tag}_mfma_%s/**/*counter_collection.csv" % part, recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            if part == "llm" and name.startswith(("k_conv", "k_dwln", "k_gemv1", "k_codebook", "k_fsq", "k_zero", "__amd")):
                continue
            k = (name, r.get("Grid_Size", ""))
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    print("## " + title)
    print(f"{'kernel':60s} {'grid':>9s} {'n':>5s} {'MfmaUtil %':>10s}")
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1] / kv[1][0]):
        if t / n >= 0.3:
            print(f"{k[0][:60]:60s} {k[1]:>9s} {n:5d} {t / n:10.2f}")
PY
cat gpurun_out/pmc_${tag}_mfma_table.txt
find gpurun_out/pmc_${tag}_mfma_voc gpurun_out/pmc_${tag}_mfma_llm -name "*.csv" -size +2M -delete
