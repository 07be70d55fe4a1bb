#!/bin/bash
# time the bench-shape CQT1992v2 forward with variant builds of the chain kernel:  bash scripts/chain_variant.sh "-DCH_PRIO=1" "-DCH_PRIO=2" ...
cd nnaudio_amd/csrc
cp libmispec.so /tmp/libmispec_keep.so
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -Wno-int-to-pointer-cast -Wno-unused-value $flags -I ../../include -c cqt_chain.hip -o /tmp/cqt_chain_var.o \
   && hipcc --offload-arch=gfx950 -shared -fPIC _obj/mispec.o _obj/octave_stream.o /tmp/cqt_chain_var.o -o libmispec.so \
   && (cd ../..; for r in 1 2 3; do echo -n "[$flags] "; timeout 120 python bench.py --workload cqt --steps 100 --warmup 20 --extras 0 --cpu-baseline 0 --traffic off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done)
done
cp /tmp/libmispec_keep.so libmispec.so
