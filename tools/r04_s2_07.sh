cd $GRAFT_REPO_ROOT
for v in "SPARKMI_CB2_XCD=0" "SPARKMI_CB2_XCD=1" "SPARKMI_CB2_W3=1"; do
  echo "== $v"; env $v VOC_PROFILE_ALL=1 timeout -k 10 100 python tools/voc_profile.py 32 150 2>&1 | grep -E "forward|sum of|conv7@|conv1.res@" | head -40
done
