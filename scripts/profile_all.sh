#!/bin/bash
# All rocprofv3 summaries of a round: bash scripts/profile_all.sh r02  (on the GPU box)
set -u
TAG=${1:-r02}
for spec in "stft bf16x3" "stft fp32" "cqt bf16x3" "mel bf16x3" "cqt2010 bf16x3" "vqt bf16x3"; do
  set -- $spec
  bash scripts/profile.sh ${TAG}_$1_$2 $1 $2 > /dev/null 2>&1
  mkdir -p gpurun_out/profiles_$TAG
  cp gpurun_out/prof_${TAG}_$1_$2/summary/summary_$1_$2.txt gpurun_out/profiles_$TAG/rocprofv3_$1_$2_summary.txt
  cp gpurun_out/prof_${TAG}_$1_$2/trace/trace_kernel_stats.csv gpurun_out/profiles_$TAG/rocprofv3_$1_$2_kernel_stats.csv 2>/dev/null
  rm -rf gpurun_out/prof_${TAG}_$1_$2
done
ls -la gpurun_out/profiles_$TAG
