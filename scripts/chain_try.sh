#!/bin/bash
# build a -DCH_ABL=<bits> variant of the chain kernel into libmispec.so, run one small forward, restore the library
cd nnaudio_amd/csrc
cp libmispec.so /tmp/libmispec_keep.so
for bits in $1; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -Wno-int-to-pointer-cast -Wno-unused-value -DCH_ABL=$bits -I ../../include -c cqt_chain.hip -o /tmp/cqt_chain_abl.o \
 && hipcc --offload-arch=gfx950 -shared -fPIC _obj/mispec.o _obj/octave_stream.o /tmp/cqt_chain_abl.o -o libmispec.so \
 && (cd ../..; echo "CH_ABL=$bits"; timeout 60 python scripts/chain_check.py --small 2>&1 | tail -2)
done
cp /tmp/libmispec_keep.so libmispec.so
