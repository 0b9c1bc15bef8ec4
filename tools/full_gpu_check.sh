set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_b1.json 2> gpurun_out/bench_b1.err
cat gpurun_out/bench_b1.json
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --batch 32 --no-cpu-baseline > gpurun_out/bench_b32.json 2> gpurun_out/bench_b32.err
cat gpurun_out/bench_b32.json
