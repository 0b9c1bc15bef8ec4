# final GPU test run of the round's second half (GPU box): bash tools/r04_final_tests2.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04_pytest_gpu_final_b.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04_pytest_gpu_final_b.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
