cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "cfg5 or 2010 or vqt or mel or Mel" 2>&1 | tail -2
for WP in "stft f16x3" "cqt f16x3" "mel f16x3" "cqt2010 f16x3" "stft bf16x3" "cqt2010 bf16x3"; do
  set -- $WP
  bash scripts/profile.sh r03_$1_$2 $1 $2 > /dev/null 2>&1
  mkdir -p gpurun_out/r03_summaries
  cp gpurun_out/prof_r03_$1_$2/summary/*.txt gpurun_out/r03_summaries/rocprofv3_$1_$2_summary.txt
  cp $(find gpurun_out/prof_r03_$1_$2/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r03_summaries/rocprofv3_$1_$2_kernel_stats.csv
  rm -rf gpurun_out/prof_r03_$1_$2/trace/*/*kernel_trace.csv gpurun_out/prof_r03_$1_$2/pmc*/
done
ls gpurun_out/r03_summaries
