# Round-3 measurement set on one box (GPU box): bash tools/r03_snapshot.sh <tag>
# PMC passes first (their jsons go under profiles/ ON THE BOX so that the bench lines that follow carry roofline.traffic),
# then the three bench lines, then rocprofv3 --kernel-trace --stats of the B = 1 and B = 32 commands.
set -e
cd $GRAFT_REPO_ROOT
tag=${1:-r03}
mkdir -p gpurun_out
for b in 1 8 32; do
  bash tools/pmc_traffic.sh ${tag}_b$b $b > gpurun_out/${tag}_pmc_b$b.log 2>&1
  cp gpurun_out/pmc_traffic_${tag}_b$b.json profiles/r03_pmc_traffic_${tag}_b$b.json
done
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench_b1.json 2> gpurun_out/${tag}_bench_b1.err
timeout -k 10 300 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_b32.json 2> gpurun_out/${tag}_bench_b32.err
timeout -k 10 300 python bench.py --clone --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/${tag}_bench_clone8.json 2> gpurun_out/${tag}_bench_clone8.err
bash tools/bench_prof.sh ${tag}
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_b32 -o r -- python3 bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/bench_under_rocprof_${tag}_b32.json 2> gpurun_out/prof_${tag}_b32.log
find gpurun_out/prof_${tag}_b32 -name "*kernel_trace*" -delete
tail -c 300 gpurun_out/${tag}_bench_b1.json
