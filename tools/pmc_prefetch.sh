# Does the helper blocks' prefetch land?  L2 hit / miss and fabric read requests per decode kernel with the prefetch on and off
# (GPU box): bash tools/pmc_prefetch.sh <tag>
# One pass per counter group, --kernel-trace only, eager launches (12 tokens = 264+ launches per kernel).  The profiler serialises
# the kernels; what it shows is whether lines a producer's helpers requested are still in the consumer's L2 after the kernel
# boundary (TCC_HIT / TCC_MISS of the consumer) and whether the consumer's fabric reads drop (TCC_EA0_RDREQ).
set -e
cd $GRAFT_REPO_ROOT
tag=${1:-x}
export TMPDIR=/tmp
mkdir -p gpurun_out
for mode in on off; do
  for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    g=$(echo $grp | tr ' ' '_')
    if [ $mode = off ]; then export SPARKMI_NO_PREFETCH=1; else unset SPARKMI_NO_PREFETCH; fi   # (any value turns the prefetch off)
    timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmcpf_${tag}_${mode}_$g -o r -- python3 bench.py --steps 1 --warmup 0 --new-tokens 12 --no-cpu-baseline --no-probes --no-graph --no-config4 > gpurun_out/pmcpf_${tag}_${mode}_$g.json 2> gpurun_out/pmcpf_${tag}_${mode}_$g.log || { tail -5 gpurun_out/pmcpf_${tag}_${mode}_$g.log; exit 1; }
  done
done
python3 - <<PY | tee gpurun_out/pmcpf_${tag}_table.txt
import csv, glob, re
from collections import defaultdict
def load(mode):
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob("gpurun_out/pmcpf_${tag}_%s_*/**/*counter_collection.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name][r["Counter_Name"]] += 1
    return {k: {c: v / cnt[k][c] for c, v in d.items()} for k, d in acc.items()}, {k: max(d.values()) for k, d in cnt.items()}
on, n_on = load("on"); off, n_off = load("off")
print("# per launch, decode kernels at one row (eager, 12 tokens); prefetch on | off (SPARKMI_NO_PREFETCH=1)")
print("%-44s %6s | %10s %10s %6s %10s | %10s %10s %6s %10s" % ("kernel", "n", "L2 hit", "L2 miss", "hit%", "EA rdreq", "L2 hit", "L2 miss", "hit%", "EA rdreq"))
for k in sorted(on, key=lambda k: -n_on[k]):
    if n_on[k] < 200 and not k.startswith("k_lm"): continue
    a, b = on[k], off.get(k, {})
    f = lambda d: (d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0), 100 * d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1), d.get("TCC_EA0_RDREQ_sum", 0))
    print("%-44s %6d | %10.0f %10.0f %6.1f %10.0f | %10.0f %10.0f %6.1f %10.0f" % ((k[:44], n_on[k]) + f(a) + f(b)))
PY
find gpurun_out -path "*pmcpf_${tag}_*" -name "*.csv" -size +2M -delete
