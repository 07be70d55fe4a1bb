#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS of octave_stream_kernel (first launch of the cfg5 shard) per ablation
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/stream_conflicts
rm -rf $OUT; mkdir -p $OUT
for D in ${1:-0 1 2 3 257 258 1027}; do
  (cd /tmp && DBG=$D timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/d$D -o pmc -- python $OLDPWD/scripts/stream_conflicts.py > $OUT/d$D.log 2>&1)
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/d$D/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "octave_stream" in r["Kernel_Name"]]
    # the last 4 dispatches of the kernel are the ablated ones
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in by.items():
        agg[c] = v[-4:]
if agg:
    g = lambda c: sum(agg[c]) / max(len(agg[c]), 1)
    print("dbg %5d: conflict %.4g  active %.4g  (%.1f %%)  lds instructions %.4g  wave cycles %.4g" % ($D, g("SQ_LDS_BANK_CONFLICT"), g("SQ_LDS_IDX_ACTIVE"), 100 * g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1), g("SQ_INSTS_LDS"), g("SQ_WAVE_CYCLES")))
else:
    print("dbg $D: no counters (see $OUT/d$D.log)")
PY
done
