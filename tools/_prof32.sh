cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b32 -o r -- python3 bench.py --steps 3 --warmup 1 --batch 32 --no-cpu-baseline --no-probes > gpurun_out/bench_under_rocprof_b32.json 2> gpurun_out/prof_b32.log
find gpurun_out/prof_b32 -name "*kernel_trace*" -delete
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_b32/r_kernel_stats.csv')))
for r in rows[:16]:
    print(r['Name'][:80].ljust(80), r['Calls'], r['AverageNs'], r['Percentage'])
PY
