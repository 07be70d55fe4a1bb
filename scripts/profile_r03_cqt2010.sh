cd /root/repo
bash scripts/profile.sh r03_cqt2010_f16x3 cqt2010 f16x3 > /dev/null 2>&1
mkdir -p gpurun_out/r03_summaries
cp gpurun_out/prof_r03_cqt2010_f16x3/summary/*.txt gpurun_out/r03_summaries/rocprofv3_cqt2010_f16x3_summary.txt
cp $(find gpurun_out/prof_r03_cqt2010_f16x3/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r03_summaries/rocprofv3_cqt2010_f16x3_kernel_stats.csv
rm -rf gpurun_out/prof_r03_cqt2010_f16x3/trace/*/*kernel_trace.csv
head -12 gpurun_out/r03_summaries/rocprofv3_cqt2010_f16x3_summary.txt
