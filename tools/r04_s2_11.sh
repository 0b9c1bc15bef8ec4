cd $GRAFT_REPO_ROOT
for b in 32 8; do
  echo "== B=$b k_convb (SPARKMI_CBT=0)"; SPARKMI_CBT=0 VOC_PROFILE_ALL=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | grep -E "forward|sum of|convT" 
  echo "== B=$b k_convbT NC1"; SPARKMI_X=1 VOC_PROFILE_ALL=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | grep -E "forward|sum of|convT"
  echo "== B=$b k_convbT<4,2>"; SPARKMI_CBT_QB2=1 VOC_PROFILE_ALL=1 timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | grep -E "forward|sum of|convT"
done
