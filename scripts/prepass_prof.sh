#!/bin/bash
# per-kernel averages of the folded STFT / fused mel steps (pre-pass vs contraction), rocprofv3 kernel trace
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
for w in stft mel; do
  rm -rf $R/gpurun_out/prepass_$w; mkdir -p $R/gpurun_out/prepass_$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prepass_$w -o t -- python $R/bench.py --workload $w --extras 0 --cpu-baseline 0 --traffic off --steps 100 --warmup 20 > $R/gpurun_out/prepass_$w/bench.log 2>&1
  echo "== $w: $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/prepass_$w/bench.log | head -1)"
  cut -d, -f1-4 $R/gpurun_out/prepass_$w/t_kernel_stats.csv | head -4
done
