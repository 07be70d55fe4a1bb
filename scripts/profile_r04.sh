# round-4 profiles: the default bench line, then rocprofv3 kernel stats + PMC passes of the workloads as the
# modules ship them (--precision auto): STFT cfg2 (FFT route), CQT2010v2 / VQT cfg5 shard (streaming octave
# kernel), CQT84 f16x3 (strip kernel), Mel cfg3 (FFT route).  Summaries go to gpurun_out/r04_summaries/.
cd /root/repo
mkdir -p gpurun_out/r04_summaries
python bench.py > gpurun_out/r04_summaries/bench_r04.log 2>&1
tail -1 gpurun_out/r04_summaries/bench_r04.log > gpurun_out/r04_summaries/bench_r04.json
cp bench_detail.json gpurun_out/r04_summaries/bench_detail_r04.json 2>/dev/null
for WP in "stft auto fft" "cqt2010 auto stream" "vqt auto stream" "cqt f16x3 f16x3" "mel auto fft" "mfcc auto fft"; do
  set -- $WP
  bash scripts/profile.sh r04_$1_$3 $1 $2 > /dev/null 2>&1
  cp gpurun_out/prof_r04_$1_$3/summary/*.txt gpurun_out/r04_summaries/rocprofv3_$1_$3_summary.txt
  cp $(find gpurun_out/prof_r04_$1_$3/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r04_summaries/rocprofv3_$1_$3_kernel_stats.csv
  rm -rf gpurun_out/prof_r04_$1_$3/trace gpurun_out/prof_r04_$1_$3/pmc*/
done
ls gpurun_out/r04_summaries
