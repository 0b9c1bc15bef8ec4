cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
VOC_PROFILE_ALL=1 timeout -k 10 200 python tools/voc_profile.py 1 150 > gpurun_out/r04s2_voc_b1.txt 2>&1
VOC_PROFILE_ALL=1 timeout -k 10 200 python tools/voc_profile.py 32 150 > gpurun_out/r04s2_voc_b32.txt 2>&1
PF_SIZES=1x128 timeout -k 10 200 python tools/prefill_time.py r04 > gpurun_out/r04s2_prefill.txt 2>&1
head -3 gpurun_out/r04s2_voc_b1.txt gpurun_out/r04s2_voc_b32.txt gpurun_out/r04s2_prefill.txt
