cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_llm_ops_gpu.py tests/test_llm_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "ops or batched_decode or few_row or ragged or fused_o_proj or session or retire or serve" > gpurun_out/r04_pytest_gpu_g.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r04_pytest_gpu_g.log
timeout -k 10 600 python tools/r04_batch_ab.py 8 4 2 > gpurun_out/r04_batch_ab4.txt 2>&1; grep "^B " gpurun_out/r04_batch_ab4.txt
