cd $GRAFT_REPO_ROOT
for v in o0b0 o1b0 o1b1 o1b1occ3; do
  echo "== $v"; SPARKMI_LIB=$PWD/spark-tts_amd/sparkmi/ab/libsparkmi_$v.so timeout -k 10 100 python tools/voc_profile.py 32 150 2>&1 | sed -n 2,9p
done
echo "== old kernel"; SPARKMI_CB2_MIN=0 timeout -k 10 100 python tools/voc_profile.py 32 150 2>&1 | sed -n 2,9p
