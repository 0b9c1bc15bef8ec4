set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_enc_gpu.py tests/test_abi.py -x -q -m gpu > gpurun_out/r03a_tests.txt 2>&1
timeout -k 10 400 python bench.py > gpurun_out/r03a_bench_b1.json 2> gpurun_out/r03a_bench_b1.err
bash tools/bench_prof.sh r03a
bash tools/pmc_traffic.sh r03a_b1 1
timeout -k 10 300 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03a_bench_b32.json 2> gpurun_out/r03a_bench_b32.err
tail -c 600 gpurun_out/r03a_bench_b1.json
