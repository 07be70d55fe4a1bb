#!/bin/bash
# FETCH/WRITE traffic of the dominant STFT kernel under the two tile orders
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ablate
mkdir -p $OUT
for dbg in 256 512; do
  for P in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
    tag=$(echo $P | tr ' ' '_')
    rocprofv3 --pmc $P --output-format csv -d $OUT/d${dbg}_$tag -o pmc -- python scripts/prof_ablate.py $dbg > $OUT/d${dbg}_$tag.log 2>&1
    echo "=== debug=$dbg $P"; python scripts/summarize_prof.py $OUT/d${dbg}_$tag | grep -A6 "2, 2, 2, 2" | grep per-dispatch
  done
done
