# A/B timing of two builds of libmispec.so (abtmp/A.so, abtmp/B.so) on the same GPU box:
#   bash scripts/ab.sh <kbench section> <grep pattern>
SEC=${1:-bf16}
PAT=${2:-ablate|module Magnitude}
for rep in 1 2; do
for v in A B; do
cp abtmp/$v.so nnaudio_amd/csrc/libmispec.so
echo "== $v"
timeout 300 python scripts/kbench.py $SEC 2>&1 | grep -E "$PAT"
done
done
