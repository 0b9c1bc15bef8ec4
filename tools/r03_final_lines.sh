# The three bench lines of the final build with the PMC profiles under profiles/ (GPU box): bash tools/r03_final_lines.sh
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/r03f_bench_b1.json 2> gpurun_out/r03f_bench_b1.err
timeout -k 10 300 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03f_bench_b32.json 2> gpurun_out/r03f_bench_b32.err
timeout -k 10 300 python bench.py --clone --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r03f_bench_clone8.json 2> gpurun_out/r03f_bench_clone8.err
tail -c 200 gpurun_out/r03f_bench_b1.json
