cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 32 8 1; do
  echo "== B=$b chunks 32 (SPARKMI_CB_CHG=0)"; SPARKMI_CB_CHG=0 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | sed -n 2,8p
  echo "== B=$b new"; SPARKMI_X=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | sed -n 2,8p
done
echo "== enc old"; SPARKMI_CB_CHG=0 timeout -k 10 100 python tools/enc_profile.py 6 2>&1 | sed -n 2,3p
echo "== enc new"; SPARKMI_X=1 timeout -k 10 100 python tools/enc_profile.py 6 2>&1 | sed -n 2,3p
timeout -k 10 600 python -m pytest tests/test_voc_gpu.py tests/test_ops_gpu.py tests/test_enc_gpu.py tests/test_streaming.py -x -q -m gpu > gpurun_out/r04s2_t8.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04s2_t8.txt
