cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r04_smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest_gpu_final.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04_pytest_gpu_final.log
