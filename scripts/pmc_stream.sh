#!/bin/bash
# instruction-side counters of the streaming octave kernel on the cfg5 shard (GPU box, from the repo root):
# instruction cache, branches, scalar memory, FIFO-full stalls.  One rocprofv3 --pmc pass per group.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_stream
rm -rf $OUT; mkdir -p $OUT
CMD="python scripts/cfg5_fwd.py cqt2010 6"
G1="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQC_DCACHE_MISSES"
G2="SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY"
G3="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
G4="SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
i=1
for G in "$G1" "$G2" "$G3" "$G4"; do
  (cd /tmp && rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -o pmc -- python $OLDPWD/scripts/cfg5_fwd.py cqt2010 6 > $OUT/g$i.log 2>&1)
  i=$((i+1))
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "octave_stream" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60], r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k)
        for c, v in sorted(d.items()):
            print("   %-34s per-dispatch=%.6g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
