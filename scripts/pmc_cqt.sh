#!/bin/bash
# extra counter passes for the CQT bench path: bash scripts/pmc_cqt.sh "<counters>" ["<counters>" ...]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
i=0
for P in "$@"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcx$i
  timeout 200 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/pmcx$i -o p -- python $R/bench.py --workload cqt --extras 0 --cpu-baseline 0 --traffic off --steps 3 --warmup 1 > $R/gpurun_out/pmcx$i.log 2>&1
  python - <<PY
import csv, collections, os
f = "$R/gpurun_out/pmcx$i/p_counter_collection.csv"
if not os.path.exists(f):
    print("pass $i ($P): no output"); print(open("$R/gpurun_out/pmcx$i.log").read()[-600:])
else:
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "strip" not in r["Kernel_Name"] and "narrow" not in r["Kernel_Name"]: continue
        k = r["Counter_Name"]
        d[k][0] += 1; d[k][1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(d.items()):
        print("%-40s %.5g per dispatch" % (k, v / n))
PY
done
