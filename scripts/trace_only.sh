export TMPDIR=/tmp
for WL in stft mel cqt; do
OUT=$PWD/gpurun_out/prof_r01d_$WL
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --extras 0 --cpu-baseline 0 --workload $WL --precision bf16x3 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log > gpurun_out/keep2/bench_under_trace_$WL.json
cp $OUT/trace/*kernel_stats.csv gpurun_out/keep2/r01d_${WL}_kernel_stats.csv
rm -rf $OUT/trace
done
