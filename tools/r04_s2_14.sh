cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_voc_gpu.py tests/test_ops_gpu.py tests/test_streaming.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r04s2_t14.txt 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r04s2_t14.txt
for b in 32 8 1; do
  echo "== B=$b two launches (SPARKMI_RESFUSE=0)"; SPARKMI_RESFUSE=0 VOC_PROFILE_ALL=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | grep -E "forward|sum of|model.[34].block.[234].block" 
  echo "== B=$b fused"; SPARKMI_X=1 VOC_PROFILE_ALL=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | grep -E "forward|sum of|model.[34].block.[234].block"
done
