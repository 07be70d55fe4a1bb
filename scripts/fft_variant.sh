#!/bin/bash
# time the headline STFT step with variant builds of mispec.hip:  bash scripts/fft_variant.sh "-DMISPEC_FFT_PRIO=1" ...
cd nnaudio_amd/csrc
cp libmispec.so /tmp/libmispec_keep.so
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -Wno-int-to-pointer-cast $flags -I ../../include -c mispec.hip -o /tmp/mispec_var.o \
   && hipcc --offload-arch=gfx950 -shared -fPIC /tmp/mispec_var.o _obj/octave_stream.o _obj/cqt_chain.o -o libmispec.so \
   && (cd ../..; for r in 1 2 3; do echo -n "[$flags] "; timeout 120 python bench.py --steps 300 --warmup 50 --extras 0 --cpu-baseline 0 --traffic off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done)
done
cp /tmp/libmispec_keep.so libmispec.so
