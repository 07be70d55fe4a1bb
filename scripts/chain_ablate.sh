#!/bin/bash
# build libmispec.so variants with the chain kernel's development ablations (-DCH_ABL=bits) and time them (GPU box)
#   bash scripts/chain_ablate.sh "0 1 2 4 8 12 16"
set -u
cd nnaudio_amd/csrc
cp libmispec.so /tmp/libmispec_keep.so
for bits in $1; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -Wno-int-to-pointer-cast -DCH_ABL=$bits -I ../../include -c cqt_chain.hip -o /tmp/cqt_chain_abl.o 2>/dev/null \
   && hipcc --offload-arch=gfx950 -shared -fPIC _obj/mispec.o _obj/octave_stream.o /tmp/cqt_chain_abl.o -o libmispec.so \
   && (cd ../..; echo -n "CH_ABL=$bits: "; timeout 120 python scripts/chain_check.py --one | tail -1; echo -n "   single unit (B=2, 3 s): "; timeout 120 python scripts/chain_check.py --small | tail -1)
done
cp /tmp/libmispec_keep.so libmispec.so
