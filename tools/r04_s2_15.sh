cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_voc_gpu.py tests/test_ops_gpu.py tests/test_streaming.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r04s2_t15.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04s2_t15.txt
for i in 1 2 3; do timeout -k 10 200 python -m pytest tests/test_voc_gpu.py -x -q -m gpu -k "multi_phase or fused_residual or ragged or full_size" 2>&1 | tail -1; done
for b in 32 8 1; do timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | sed -n 2,7p; done
