# bench + rocprofv3 kernel stats of the same command (GPU box):  bash tools/bench_prof.sh <tag>
set -e
cd $GRAFT_REPO_ROOT
tag=${1:-x}
mkdir -p gpurun_out
true
true
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -o r -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config4 > gpurun_out/bench_under_rocprof_${tag}.json 2> gpurun_out/prof_${tag}.log
find gpurun_out/prof_${tag} -name "*kernel_stats*" | head
find gpurun_out/prof_${tag} -name "*kernel_trace*" -delete
