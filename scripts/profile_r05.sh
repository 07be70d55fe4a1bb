# round-5 profiles: the default bench line, then rocprofv3 kernel stats + PMC passes of the workloads as the modules ship them:
# STFT cfg2 (FFT route), Mel cfg3, CQT84 default (fp32) and the opt-in f16x3, CQT2010v2 cfg5 shard, STFT n_fft = 4096 (composite),
# and the kernel trace of the training step.  Summaries go to gpurun_out/r05_summaries/ (copied to profiles/r05/ by hand).
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_summaries
python bench.py > gpurun_out/r05_summaries/bench_r05.log 2>&1
tail -1 gpurun_out/r05_summaries/bench_r05.log > gpurun_out/r05_summaries/bench_r05.json
cp bench_detail.json gpurun_out/r05_summaries/bench_detail_r05.json 2>/dev/null
for WP in "stft auto fft" "mel auto fft" "cqt auto default_fp32" "cqt f16x3 f16x3" "cqt2010 auto stream" "stft4096 auto fft_composite"; do
  set -- $WP
  bash scripts/profile.sh r05_$1_$3 $1 $2 > /dev/null 2>&1
  cp gpurun_out/prof_r05_$1_$3/summary/*.txt gpurun_out/r05_summaries/rocprofv3_$1_$3_summary.txt
  cp $(find gpurun_out/prof_r05_$1_$3/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r05_summaries/rocprofv3_$1_$3_kernel_stats.csv
  rm -rf gpurun_out/prof_r05_$1_$3/trace gpurun_out/prof_r05_$1_$3/pmc*/
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_train -o trace -- python scripts/train_step.py auto 10 > gpurun_out/r05_summaries/train_step.log 2>&1
python scripts/summarize_prof.py gpurun_out/prof_r05_train > gpurun_out/r05_summaries/rocprofv3_train_step_summary.txt 2>&1
rm -rf gpurun_out/prof_r05_train
ls gpurun_out/r05_summaries
