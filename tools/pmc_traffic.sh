# HBM traffic per launch from PMC counters (GPU box): bash tools/pmc_traffic.sh <tag>
# Separate passes per counter, --kernel-trace only (no other trace domains), eager launches (the profiler's counter
# collection does not survive hipGraph replay), 12 tokens = 264+ launches per decode kernel.
set -e
cd $GRAFT_REPO_ROOT
tag=${1:-x}
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -o r -- python3 bench.py --steps 1 --warmup 0 --new-tokens 12 --no-cpu-baseline --no-probes --no-graph > gpurun_out/pmc_${tag}_$c.json 2> gpurun_out/pmc_${tag}_$c.log
done
python3 tools/pmc_table.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE > gpurun_out/pmc_${tag}_table.txt
cat gpurun_out/pmc_${tag}_table.txt | head -30
find gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE -name "*.csv" -size +2M -delete
