cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_llm_ops_gpu.py tests/test_llm_gpu.py -m gpu -q -x -k "ops or exact_weights or arena_and_config or sampl or fp32_checkpoint or ragged_batch or few_row" > gpurun_out/r04_pytest_gpu_c.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r04_pytest_gpu_c.log
timeout -k 10 700 python tools/r04_batch_ab.py 32 16 8 4 > gpurun_out/r04_batch_ab2.txt 2>&1; grep "^B " gpurun_out/r04_batch_ab2.txt
