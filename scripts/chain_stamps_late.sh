#!/bin/bash
# phase clock of a LATE workgroup (warm instruction cache) of the chain kernel in the bench-shape launch
#   bash scripts/chain_stamps_late.sh <workgroup> <lines>
cd nnaudio_amd/csrc
cp libmispec.so /tmp/libmispec_keep.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -Wno-int-to-pointer-cast -Wno-unused-value -DCH_ABL=${CHABL:-64} -DCH_STAMP_WG=${1:-600} ${3:-} -I ../../include -c cqt_chain.hip -o /tmp/cqt_chain_abl.o \
 && hipcc --offload-arch=gfx950 -shared -fPIC _obj/mispec.o _obj/octave_stream.o /tmp/cqt_chain_abl.o -o libmispec.so \
 && (cd ../..; MISPEC_CHAIN_STAMPS=1 timeout 120 python scripts/chain_check.py --one 2>&1 | grep -A80 "chain stamps" | tail -${2:-32})
cp /tmp/libmispec_keep.so libmispec.so
