# A/B timing of two builds of libmispec.so (abtmp/A.so, abtmp/B.so) on the same GPU box
for rep in 1 2; do
for v in A B; do
cp abtmp/$v.so nnaudio_amd/csrc/libmispec.so
echo "== $v"
timeout 300 python scripts/kbench.py bf16 2>&1 | grep -E "ablate\[|module Magnitude|module Complex"
done
done
