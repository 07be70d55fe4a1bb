#!/bin/bash
# rocprofv3 passes over the chain kernel (cqt_chain.hip) on CQT1992v2 B = 64 x 10 s: bash scripts/chain_pmc.sh <tag>
set -u
TAG=${1:-chain}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
CMD="timeout 120 python scripts/chain_check.py --one"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"
P2="FETCH_SIZE"
P3="WRITE_SIZE"
P4="GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL"
P5="SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT"
P6="TCC_HIT_sum TCC_MISS_sum"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o pmc -- $CMD > $OUT/pmc$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, glob, os
from collections import defaultdict
root = "$OUT"
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("%-80s calls=%s avg_ns=%s" % (r["Name"][:80], r["Calls"], r["AverageNs"]))
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    agg = defaultdict(float); n = defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "cqt_chain_kernel" in r.get("Kernel_Name", ""):
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(agg):
        print("  %-34s per dispatch %.6g (%d dispatches)" % (k, agg[k] / n[k], n[k]))
PY
