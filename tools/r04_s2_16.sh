cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --no-cpu-baseline --no-probes > gpurun_out/r04s2_bench_b1.json 2> gpurun_out/r04s2_bench_b1.err; echo "b1 rc $?"
timeout -k 10 300 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r04s2_bench_b32.json 2> gpurun_out/r04s2_bench_b32.err; echo "b32 rc $?"
timeout -k 10 300 python bench.py --clone --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r04s2_bench_clone8.json 2> gpurun_out/r04s2_bench_clone8.err; echo "clone rc $?"
python - <<'P'
import json
for n in ("b1","b32","clone8"):
    d=json.load(open(f"gpurun_out/r04s2_bench_{n}.json"))
    print(n, {k:d[k] for k in ("value","ms_per_step")}, d["stage_ms"], d.get("config4",{}).get("value"))
P
