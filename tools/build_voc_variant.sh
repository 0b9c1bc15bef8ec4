# A/B build of the conv kernels only: bash tools/build_voc_variant.sh NAME "-DFLAG=.. -DFLAG2=.."  ->  spark-tts_amd/sparkmi/ab/libsparkmi_NAME.so
# (smi_voc.hip and smi_enc.hip recompiled with the flags, the LLM / core objects of the diagnostics build reused)
set -e
cd "$(dirname "$0")/../spark-tts_amd/csrc"
name=$1; flags=$2
mkdir -p ab/$name ../sparkmi/ab
CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -Wno-unused-function -ffp-contract=off -DSMI_DIAG"
for f in smi_voc smi_enc; do /opt/rocm/bin/hipcc $CXXFLAGS $flags -c $f.hip -o ab/$name/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../sparkmi/ab/libsparkmi_$name.so ab/$name/smi_voc.o ab/$name/smi_enc.o diag/d_smi_llm.o diag/d_smi_core.o diag/diag_*.o
echo built ../sparkmi/ab/libsparkmi_$name.so
