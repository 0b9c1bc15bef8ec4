cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python tools/r04_batch_ab.py 8 2 1 > gpurun_out/r04_batch_ab5.txt 2>&1; grep "^B " gpurun_out/r04_batch_ab5.txt
