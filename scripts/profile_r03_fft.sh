# round-3 final profiles of the FFT route: STFT cfg2 and Mel cfg3 as the modules ship (--precision auto)
cd /root/repo
for WL in stft mel; do
  bash scripts/profile.sh r03_${WL}_fft $WL auto > /dev/null 2>&1
  mkdir -p gpurun_out/r03_summaries
  cp gpurun_out/prof_r03_${WL}_fft/summary/*.txt gpurun_out/r03_summaries/rocprofv3_${WL}_fft_summary.txt
  cp $(find gpurun_out/prof_r03_${WL}_fft/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r03_summaries/rocprofv3_${WL}_fft_kernel_stats.csv
  rm -rf gpurun_out/prof_r03_${WL}_fft/trace/*/*kernel_trace.csv
done
ls gpurun_out/r03_summaries
