cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_llm_gpu.py tests/test_engine_gpu.py tests/test_llm_ops_gpu.py -m gpu -q -x -k "golden or greedy or single_row or engine or few_row or fused" > gpurun_out/r04_pytest_gpu_h.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r04_pytest_gpu_h.log
timeout -k 10 600 python tools/variants.py 0 lib:spark-tts_amd/sparkmi/ab/libsparkmi_old.so 0 lib:spark-tts_amd/sparkmi/ab/libsparkmi_old.so 0 lib:spark-tts_amd/sparkmi/ab/libsparkmi_old.so > gpurun_out/r04_fusedo_float2_ab.txt 2>&1; cat gpurun_out/r04_fusedo_float2_ab.txt
