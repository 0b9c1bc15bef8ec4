# Builds the COMMITTED (git HEAD) HIP sources as spark-tts_amd/sparkmi/ab/libsparkmi_head.so, for A/B against the working tree:
#   bash tools/build_head_variant.sh && gpurun -- 'python tools/variants.py lib:spark-tts_amd/sparkmi/ab/libsparkmi_head.so env:SPARKMI_X=0'
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/csrc_head && mkdir -p /tmp/csrc_head spark-tts_amd/sparkmi/ab
for f in $(git ls-files spark-tts_amd/csrc | grep -E 'smi_[a-z_]*\.(hip|h)$'); do git show HEAD:$f > /tmp/csrc_head/$(basename $f); done
cd /tmp/csrc_head
for f in smi_core smi_enc smi_llm smi_voc; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/root/repo/include -I. -Wall -Wno-unused-function -ffp-contract=off -c $f.hip -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/spark-tts_amd/sparkmi/ab/libsparkmi_head.so smi_core.o smi_enc.o smi_llm.o smi_voc.o
ls -la /root/repo/spark-tts_amd/sparkmi/ab/
