# Full GPU regression (GPU box): tests, the default bench line (with the CPU baseline), batch 32 and the voice-clone config.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_b1.json 2> gpurun_out/bench_b1.err
cut -c1-700 gpurun_out/bench_b1.json
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --batch 32 --no-cpu-baseline > gpurun_out/bench_b32.json 2> gpurun_out/bench_b32.err
cut -c1-300 gpurun_out/bench_b32.json
timeout -k 10 300 python bench.py --clone --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-probes > gpurun_out/bench_clone8.json 2> gpurun_out/bench_clone8.err
cut -c1-300 gpurun_out/bench_clone8.json
timeout -k 10 200 python tools/enc_profile.py 6 > gpurun_out/enc_profile.txt 2>&1
head -3 gpurun_out/enc_profile.txt
