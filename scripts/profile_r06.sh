# round-6 profiles: the default bench line, then rocprofv3 kernel stats + PMC passes of the workloads as the modules ship them
# (STFT cfg2, Mel cfg3, CQT84 default = the chain kernel, CQT2010v2 cfg5 shard, Gammatonegram).  Summaries go to
# gpurun_out/r06_summaries/ (copied to profiles/r06/ by hand).
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_summaries
timeout 900 python bench.py > gpurun_out/r06_summaries/bench_r06.log 2>&1
tail -1 gpurun_out/r06_summaries/bench_r06.log > gpurun_out/r06_summaries/bench_r06.json
cp bench_detail.json gpurun_out/r06_summaries/bench_detail_r06.json 2>/dev/null
for WP in "stft auto fft" "mel auto fft" "cqt auto default_fp32_chain" "cqt2010 auto stream" "gammatone auto frame_major"; do
  set -- $WP
  timeout 600 bash scripts/profile.sh r06_$1_$3 $1 $2 > /dev/null 2>&1
  cp gpurun_out/prof_r06_$1_$3/summary/*.txt gpurun_out/r06_summaries/rocprofv3_$1_$3_summary.txt
  cp $(find gpurun_out/prof_r06_$1_$3/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r06_summaries/rocprofv3_$1_$3_kernel_stats.csv
  rm -rf gpurun_out/prof_r06_$1_$3/trace gpurun_out/prof_r06_$1_$3/pmc*/
done
ls gpurun_out/r06_summaries
