#!/bin/bash
# rocprofv3 passes for the bench workload; run on the GPU box from the repo root:
#   bash scripts/profile.sh <tag> [workload] [precision]
# Writes per-pass outputs under gpurun_out/prof_<tag>/ and compact summaries (the files that get
# committed under profiles/) under gpurun_out/prof_<tag>/summary/.
set -u
TAG=${1:-r01}
WL=${2:-stft}
PREC=${3:-f16x3}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT/summary
CMD="python bench.py --steps 5 --warmup 2 --prewarm-ms 0 --extras 0 --cpu-baseline 0 --traffic off --workload $WL --precision $PREC"
# the first launches after start-up run slower (clock ramp): the timing pass uses the bench's
# default step counts so that the per-kernel averages are the steady-state ones the bench reports
TCMD="python bench.py --extras 0 --cpu-baseline 0 --traffic off --workload $WL --precision $PREC"

# (1) kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $TCMD > $OUT/trace.log 2>&1
# (2..) PMC passes: counters only (never combined with trace domains)
MOPS=SQ_INSTS_VALU_MFMA_MOPS_F32
[ "$PREC" = "bf16x3" ] && MOPS=SQ_INSTS_VALU_MFMA_MOPS_BF16
[ "$PREC" = "f16x3" ] && MOPS=SQ_INSTS_VALU_MFMA_MOPS_F16
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE $MOPS"
P2="FETCH_SIZE"
P3="WRITE_SIZE"
P4="GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o pmc -- $CMD > $OUT/pmc$i.log 2>&1
  i=$((i+1))
done
python scripts/summarize_prof.py $OUT > $OUT/summary/summary_${WL}_$PREC.txt 2>&1
cat $OUT/summary/summary_${WL}_$PREC.txt
