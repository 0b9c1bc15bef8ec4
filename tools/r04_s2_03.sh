cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_voc_gpu.py -x -q -m gpu -k "register_tiled or full_size" > gpurun_out/r04s2_t3.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04s2_t3.txt
for b in 32 8 1; do VOC_PROFILE_ALL=1 timeout -k 10 200 python tools/voc_profile.py $b 150 > gpurun_out/r04s2_voc2_b$b.txt 2>&1; sed -n 2,8p gpurun_out/r04s2_voc2_b$b.txt; done
SPARKMI_CB2_MIN=0 timeout -k 10 200 python tools/voc_profile.py 8 150 2>&1 | sed -n 2,8p
timeout -k 10 300 python tools/variants.py env:SPARKMI_FAKE_OHEADS=4 0 > gpurun_out/r04s2_fake_oheads.txt 2>&1; cat gpurun_out/r04s2_fake_oheads.txt
