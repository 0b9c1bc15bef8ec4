cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04s2_t12.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04s2_t12.txt
for b in 32 8 1; do timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | sed -n 2,8p; done
