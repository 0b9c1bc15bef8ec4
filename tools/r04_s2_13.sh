cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
VARIANTS_B=32 timeout -k 10 400 python tools/variants.py 0 env:SPARKMI_XF=1 0 env:SPARKMI_XF=1 > gpurun_out/r04s2_xf_b32.txt 2>&1; cat gpurun_out/r04s2_xf_b32.txt
VARIANTS_B=20 timeout -k 10 300 python tools/variants.py 0 env:SPARKMI_XF=1 2>&1 | tail -2
SPARKMI_XF=1 timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py tests/test_llm_gpu.py -x -q -m gpu -k "batch or ragged or config" > gpurun_out/r04s2_t13.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04s2_t13.txt
