#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ablate
mkdir -p $OUT
P1="SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
for dbg in 256 257 259; do
  i=1
  for P in "$P1" "$P2"; do
    rocprofv3 --pmc $P --output-format csv -d $OUT/d${dbg}_p$i -o pmc -- python scripts/prof_ablate.py $dbg > $OUT/d${dbg}_p$i.log 2>&1
    i=$((i+1))
  done
  echo "=== debug=$dbg"; python scripts/summarize_prof.py $OUT/d${dbg}_p1 | grep -A12 "2, 2, 2, 2" | grep -v "^##"; python scripts/summarize_prof.py $OUT/d${dbg}_p2 | grep -A12 "2, 2, 2, 2" | grep -v "^##"
done
