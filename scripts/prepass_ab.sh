#!/bin/bash
# pre-pass of the folded STFT (benchmarking build): without global stores (0x40) / loads (0x80)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
for nfft in 2048 1024; do for bits in 0 0x40 0x80 0xc0 0x200 0x2c0; do
  rm -rf $R/gpurun_out/pab; mkdir -p $R/gpurun_out/pab
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pab -o t -- python $R/scripts/prepass_ab.py $bits $nfft > /dev/null 2>&1
  python - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/pab/t_kernel_stats.csv")):
    if "fold_frames" in r["Name"] or "framed_fold_kernel" in r["Name"]:
        print("n_fft=$nfft bits=$bits %-20s avg %.1f us  min %.1f us" % (r["Name"].split("::")[1][:18], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done; done
