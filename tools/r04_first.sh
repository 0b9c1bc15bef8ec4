# Round-4 first GPU call (GPU box): the whole -m gpu suite on the refactored tree (product / diagnostics libraries), the
# two-rank rehearsal of the multi-GPU path on one card (gloo, SPARKMI_ONE_GPU=1: all the multi-rank evidence a one-GPU lease can
# give; the line is kept under profiles/), and the default bench line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest_gpu_a.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r04_pytest_gpu_a.log
SPARKMI_ONE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r04_bench_2rank_rehearsal.json 2> gpurun_out/r04_bench_2rank_rehearsal.err; echo "2-rank rc $?"; tail -c 600 gpurun_out/r04_bench_2rank_rehearsal.json; tail -5 gpurun_out/r04_bench_2rank_rehearsal.err
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r04_bench_a_b1.json 2> gpurun_out/r04_bench_a_b1.err; echo "bench rc $?"; cut -c1-600 gpurun_out/r04_bench_a_b1.json
