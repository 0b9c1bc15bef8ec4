cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_voc_gpu.py tests/test_ops_gpu.py tests/test_enc_gpu.py tests/test_streaming.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r04s2_t19.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04s2_t19.txt
for b in 1 2; do
  echo "== B=$b no WPF"; SPARKMI_CB_NOWPF=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | grep -E "forward|sum of|pwconv"
  echo "== B=$b WPF 2 ahead"; SPARKMI_X=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | grep -E "forward|sum of|pwconv"
done
echo "== enc no WPF"; SPARKMI_CB_NOWPF=1 timeout -k 10 100 python tools/enc_profile.py 6 2>&1 | sed -n 2,12p
echo "== enc WPF 2 ahead"; SPARKMI_X=1 timeout -k 10 100 python tools/enc_profile.py 6 2>&1 | sed -n 2,12p
