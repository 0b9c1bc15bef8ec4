# A/B of two library builds on the vocoder and the prompt encoder (GPU box): bash tools/lib_ab.sh <libA.so>
A=$PWD/$1
for b in 1 8 32; do
  echo "== A B=$b"; SPARKMI_LIB=$A timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | sed -n 2,7p
  echo "== new B=$b"; timeout -k 10 100 python tools/voc_profile.py $b 150 2>&1 | sed -n 2,7p
done
echo "== A enc"; SPARKMI_LIB=$A timeout -k 10 100 python tools/enc_profile.py 6 2>&1 | sed -n 2,3p
echo "== new enc"; timeout -k 10 100 python tools/enc_profile.py 6 2>&1 | sed -n 2,3p
