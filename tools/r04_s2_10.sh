cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_voc_gpu.py tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/r04s2_t10.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04s2_t10.txt
for b in 32 8 4; do
  echo "== B=$b k_convb (SPARKMI_CBT=0)"; SPARKMI_CBT=0 VOC_PROFILE_ALL=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | grep -E "forward|sum of|convT" 
  echo "== B=$b k_convbT"; SPARKMI_X=1 VOC_PROFILE_ALL=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | grep -E "forward|sum of|convT"
done
