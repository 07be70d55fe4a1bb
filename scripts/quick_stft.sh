#!/bin/bash
# quick GPU check of the STFT bench path: fold tests, bench (stft only), per-kernel averages
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests -m gpu -x -q -k "fold or cfg2 or cfg3" 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/quick; mkdir -p $R/gpurun_out/quick
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/quick/trace -o t -- python $R/bench.py --extras 0 --cpu-baseline 0 --traffic off --steps 100 --warmup 20 ${1:-} > $R/gpurun_out/quick/bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/quick/bench.log | head -1
cut -d, -f1-4 $R/gpurun_out/quick/trace/t_kernel_stats.csv | head -5
