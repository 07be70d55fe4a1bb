#!/bin/bash
# quick GPU check of the CQT1992v2 bench path: parity tests, bench (cqt only), per-kernel averages
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "cqt1992 or cfg4 or narrow or support" 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/quickc; mkdir -p $R/gpurun_out/quickc
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/quickc/trace -o t -- python $R/bench.py --workload cqt --extras 0 --cpu-baseline 0 --traffic off --steps 100 --warmup 20 ${1:-} > $R/gpurun_out/quickc/bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/quickc/bench.log | head -1
cut -d, -f1-4 $R/gpurun_out/quickc/trace/t_kernel_stats.csv | head -5
if [ "${PMC:-0}" = "1" ]; then
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/quickc/pmc -o p -- python $R/bench.py --workload cqt --extras 0 --cpu-baseline 0 --traffic off --steps 3 --warmup 1 > /dev/null 2>&1
  python - <<PY
import csv, collections
d = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open("$R/gpurun_out/quickc/pmc/p_counter_collection.csv")):
    k = (r["Kernel_Name"][:60], r["Counter_Name"])
    d[k][0] += 1; d[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(d.items()):
    print(k, "%.4g per dispatch" % (v / n))
PY
fi
# box calibration: the narrow-tile kernel of the benchmarking build (same code in every tree)
timeout 200 python $R/scripts/cqtplan.py 0x800000 2>&1 | grep debug
