cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 1 2; do
  echo "== B=$b no WPF"; SPARKMI_CB_NOWPF=1 VOC_PROFILE_ALL=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | sed -n 2,12p
  echo "== B=$b WPF"; SPARKMI_X=1 VOC_PROFILE_ALL=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | sed -n 2,12p
done
timeout -k 10 600 python -m pytest tests/test_voc_gpu.py tests/test_ops_gpu.py tests/test_enc_gpu.py tests/test_streaming.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r04s2_t9.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04s2_t9.txt
