#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace) of the chain kernel: one work unit per CU (--small) and the bench shape (--one)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/trace_${1:-chain}
mkdir -p $OUT
for m in small one; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$m -o t -- timeout 120 python scripts/chain_check.py --$m > $OUT/$m.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/$m/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cqt_chain" in r["Name"] or "framed_gemm" in r["Name"]:
            print("$m", r["Name"][:60], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3), "min_us %.1f" % (float(r["MinNs"]) / 1e3))
PY
done
