# Round-4 measurement set on one box (GPU box): bash tools/r04_snapshot.sh <tag>
# PMC passes first (their jsons go under profiles/ ON THE BOX so that the bench lines that follow carry roofline.traffic),
# then the three bench lines, the two-rank rehearsal of the multi-GPU path on one card, then rocprofv3 --kernel-trace --stats
# of the B = 1 and B = 32 commands.
set -e
cd $GRAFT_REPO_ROOT
tag=${1:-r04}
mkdir -p gpurun_out
for b in 1 8 32; do
  bash tools/pmc_traffic.sh ${tag}_b$b $b > gpurun_out/${tag}_pmc_b$b.log 2>&1
  cp gpurun_out/pmc_traffic_${tag}_b$b.json profiles/r04_pmc_traffic_${tag}_b$b.json
  cp gpurun_out/pmc_${tag}_b${b}_table.txt gpurun_out/r04_pmc_hbm_traffic_${tag}_b$b.txt
done
timeout -k 10 500 python bench.py > gpurun_out/r04_bench_${tag}_b1_cpu_baseline.json 2> gpurun_out/r04_bench_${tag}_b1.err
timeout -k 10 300 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_${tag}_b32_ragged.json 2> gpurun_out/r04_bench_${tag}_b32.err
timeout -k 10 300 python bench.py --clone --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r04_bench_${tag}_clone8.json 2> gpurun_out/r04_bench_${tag}_clone8.err
SPARKMI_ONE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r04_bench_${tag}_2rank_rehearsal_one_gpu_gloo.json 2> gpurun_out/r04_bench_${tag}_2rank_rehearsal.err
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_b1 -o r -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config4 > gpurun_out/r04_bench_under_rocprof_${tag}_b1.json 2> gpurun_out/prof_${tag}_b1.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_b32 -o r -- python3 bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r04_bench_under_rocprof_${tag}_b32.json 2> gpurun_out/prof_${tag}_b32.log
find gpurun_out/prof_${tag}_b1 gpurun_out/prof_${tag}_b32 -name "*kernel_trace*" -delete
cp $(find gpurun_out/prof_${tag}_b1 -name "*kernel_stats.csv" | head -1) gpurun_out/r04_rocprofv3_kernel_stats_${tag}_b1.csv
cp $(find gpurun_out/prof_${tag}_b32 -name "*kernel_stats.csv" | head -1) gpurun_out/r04_rocprofv3_kernel_stats_${tag}_b32.csv
tail -c 400 gpurun_out/r04_bench_${tag}_b1_cpu_baseline.json
