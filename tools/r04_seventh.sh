cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -m gpu -q -x --durations=12 > gpurun_out/r04_pytest_gpu_f.log 2>&1; echo "pytest rc $?"; tail -22 gpurun_out/r04_pytest_gpu_f.log
