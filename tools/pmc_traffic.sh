# HBM traffic per launch from PMC counters (GPU box): bash tools/pmc_traffic.sh <tag> [batch]
# Separate passes per counter, --kernel-trace only (no other trace domains), eager launches (the profiler's counter
# collection does not survive hipGraph replay), 12 tokens = 264+ launches per decode kernel.
# Writes gpurun_out/pmc_<tag>_table.txt and gpurun_out/pmc_traffic_<tag>.json (copy the latter to profiles/rNN_pmc_traffic_<tag>.json:
# bench.py takes roofline.traffic from it when its build hash matches).
set -e
cd $GRAFT_REPO_ROOT
tag=${1:-x}
batch=${2:-1}
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -o r -- python3 bench.py --steps 1 --warmup 0 --batch $batch --uniform --new-tokens 12 --no-cpu-baseline --no-probes --no-graph > gpurun_out/pmc_${tag}_$c.json 2> gpurun_out/pmc_${tag}_$c.log
done
python3 tools/pmc_table.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE --json gpurun_out/pmc_traffic_${tag}.json --batch $batch --tokens 12 > gpurun_out/pmc_${tag}_table.txt
head -30 gpurun_out/pmc_${tag}_table.txt
find gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE -name "*.csv" -size +2M -delete
