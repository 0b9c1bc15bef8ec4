cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 200 python tools/lm32_stamps.py 32 > gpurun_out/r04_lm32_stamps.txt 2>&1; cat gpurun_out/r04_lm32_stamps.txt | tail -30
timeout -k 10 500 python tools/r04_batch_ab.py 6 5 > gpurun_out/r04_batch_ab3.txt 2>&1; grep "^B " gpurun_out/r04_batch_ab3.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04a_b32 -o r -- python3 bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/bench_under_rocprof_r04a_b32.json 2> gpurun_out/prof_r04a_b32.log
find gpurun_out/prof_r04a_b32 -name "*kernel_trace*" -delete
python - <<'P'
import csv, glob
f = glob.glob("gpurun_out/prof_r04a_b32/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:16]:
    print(r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
P
