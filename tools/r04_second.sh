cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest_gpu_b.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r04_pytest_gpu_b.log
timeout -k 10 600 python tools/r04_batch_ab.py 32 8 > gpurun_out/r04_batch_ab.txt 2>&1; cat gpurun_out/r04_batch_ab.txt
