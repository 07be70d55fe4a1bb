#!/bin/bash
# quick GPU check of the cfg5 shard: bench (cqt2010 only) under a kernel trace
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/quick5; mkdir -p $R/gpurun_out/quick5
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/quick5/trace -o t -- python $R/bench.py --workload ${1:-cqt2010} --extras 0 --cpu-baseline 0 --traffic off --steps 100 --warmup 20 > $R/gpurun_out/quick5/bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/quick5/bench.log | head -1
cut -d, -f1-4 $R/gpurun_out/quick5/trace/t_kernel_stats.csv | head -12 | cut -c1-150
