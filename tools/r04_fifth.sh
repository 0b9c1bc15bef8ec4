cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python tools/lm32_stamps.py 32 > gpurun_out/r04_lm32_stamps.txt 2>&1; tail -22 gpurun_out/r04_lm32_stamps.txt
timeout -k 10 900 python -m pytest tests/test_llm_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "prefill or config3 or config5 or does_not_depend or ragged or paged or session" > gpurun_out/r04_pytest_gpu_d.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r04_pytest_gpu_d.log
timeout -k 10 600 python tools/prefill_time.py r04 > gpurun_out/r04_prefill_time.txt 2>&1; cat gpurun_out/r04_prefill_time.txt
timeout -k 10 400 python tools/r04_batch_ab.py 6 5 > gpurun_out/r04_batch_ab3.txt 2>&1; grep "^B " gpurun_out/r04_batch_ab3.txt
