#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of bench workloads, on the GPU box:
#   bash scripts/kernel_times.sh "cqt2010 f16x3" "mel auto" ...     (default: the four headline pairs)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
if [ $# -eq 0 ]; then set -- "cqt f16x3" "cqt bf16x3" "stft f16x3" "stft bf16x3"; fi
for WP in "$@"; do
  set -- $WP
  OUT=$PWD/gpurun_out/kt_$1_$2
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python bench.py --steps 100 --warmup 20 --extras 0 --cpu-baseline 0 --traffic off --workload $1 --precision $2 > $OUT/log.txt 2>&1
  echo "== $WP"; python - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("  %-70s calls=%s avg_ns=%s"%(r["Name"][:70],r["Calls"],r["AverageNs"]))
PY
done
