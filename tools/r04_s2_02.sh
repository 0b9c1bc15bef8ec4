cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python tools/variants.py 0 tune:0,0,8,0 env:SPARKMI_NO_FUSE_O=1 0 tune:0,0,8,0 > gpurun_out/r04s2_gu_ntb2.txt 2>&1
cat gpurun_out/r04s2_gu_ntb2.txt
