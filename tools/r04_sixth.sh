cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_llm_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "prefill or config3 or config5 or does_not_depend or ragged or paged or session or golden" > gpurun_out/r04_pytest_gpu_e.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r04_pytest_gpu_e.log
PF_SIZES=1x128,1x66,1x256,1x460,4x128,32x128 timeout -k 10 500 python tools/prefill_time.py r04 > gpurun_out/r04_prefill_time2.txt 2>&1; cat gpurun_out/r04_prefill_time2.txt
bash tools/prefill_prof.sh r04a 1x128 2>&1 | tail -16
timeout -k 10 200 python tools/lm32_stamps.py 32 2>&1 | tail -6
AB_B=32 AB_TAG=new timeout -k 10 300 python tools/r04_batch_ab.py 32 2>&1 | grep "^B " | head -2
