cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PF_SIZES=1x128,1x66,1x256,32x128 timeout -k 10 300 python tools/prefill_time.py r04 2>&1 | head -2
timeout -k 10 500 python bench.py > gpurun_out/r04_bench_b_b1.json 2> gpurun_out/r04_bench_b_b1.err; echo "bench rc $?"; tail -3 gpurun_out/r04_bench_b_b1.err; python - <<'P'
import json
d=json.load(open("gpurun_out/r04_bench_b_b1.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["stage_ms"], d["roofline"], d["roofline_step"]["frac"], d["cpu_baseline"]["third_party"], d.get("streaming"))
print([ (k["kernel"], round(k["avg_us"],2)) for k in d["kernels"]])
P
timeout -k 10 300 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_b_b32.json 2> gpurun_out/r04_bench_b_b32.err; echo "bench32 rc $?"; python - <<'P'
import json
d=json.load(open("gpurun_out/r04_bench_b_b32.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["stage_ms"], d["roofline_step"]["frac"], d.get("config4",{}).get("value"))
print([ (k["kernel"], round(k["avg_us"],2)) for k in d["kernels"]])
P
timeout -k 10 300 python bench.py --clone --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r04_bench_b_clone8.json 2> gpurun_out/r04_bench_b_clone8.err; echo "clone rc $?"; python - <<'P'
import json
d=json.load(open("gpurun_out/r04_bench_b_clone8.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["stage_ms"])
P
