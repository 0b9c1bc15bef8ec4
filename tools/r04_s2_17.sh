cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_voc_gpu.py tests/test_ops_gpu.py tests/test_enc_gpu.py tests/test_streaming.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r04s2_t17.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04s2_t17.txt
for b in 1 2; do
  echo "== B=$b one step ahead (SPARKMI_CB_NOWALL=1)"; SPARKMI_CB_NOWALL=1 VOC_PROFILE_ALL=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | grep -E "forward|sum of|model.[12].block.[234].block.conv7|convT|conv_in|embed" | grep -v " x "
  echo "== B=$b all taps at once"; SPARKMI_X=1 VOC_PROFILE_ALL=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | grep -E "forward|sum of|model.[12].block.[234].block.conv7|convT|conv_in|embed" | grep -v " x "
done
echo "== B=1 T=50"; SPARKMI_CB_NOWALL=1 timeout -k 10 100 python tools/voc_profile.py 1 50 2>&1 | sed -n 2,3p; SPARKMI_X=1 timeout -k 10 100 python tools/voc_profile.py 1 50 2>&1 | sed -n 2,3p
echo "== enc"; SPARKMI_CB_NOWALL=1 timeout -k 10 100 python tools/enc_profile.py 6 2>&1 | sed -n 2,3p; SPARKMI_X=1 timeout -k 10 100 python tools/enc_profile.py 6 2>&1 | sed -n 2,3p
