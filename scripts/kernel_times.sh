export TMPDIR=/tmp
cd /root/repo
for WP in "cqt f16x3" "cqt bf16x3" "stft f16x3" "stft bf16x3"; do
  set -- $WP
  OUT=/root/repo/gpurun_out/kt_$1_$2
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python bench.py --steps 100 --warmup 20 --extras 0 --cpu-baseline 0 --traffic off --workload $1 --precision $2 > $OUT/log.txt 2>&1
  echo "== $WP"; python - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("  %-70s calls=%s avg_ns=%s"%(r["Name"][:70],r["Calls"],r["AverageNs"]))
PY
done
