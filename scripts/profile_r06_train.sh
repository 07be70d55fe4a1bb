# round 6: the STFT contraction route in f16x3 (the basis GEMM: MFMA utilisation) and the kernels of a training step
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_summaries
MISPEC_FFT=0 timeout 600 bash scripts/profile.sh r06_stft_f16x3 stft f16x3 > /dev/null 2>&1
cp gpurun_out/prof_r06_stft_f16x3/summary/*.txt gpurun_out/r06_summaries/rocprofv3_stft_f16x3_summary.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06_train -o trace -- python scripts/train_step.py auto 20 > gpurun_out/r06_summaries/train_step.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_r06_train/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/r06_summaries/train_step_kernel_stats.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python scripts/train_step.py auto 20 (22 steps incl. 2 warm-up): STFT(trainable) cfg2 batch, f16x3\n")
    o.write([l for l in open("gpurun_out/r06_summaries/train_step.log") if l.startswith("train step")][-1])
    for r in rows[:16]:
        o.write("%-96s calls=%5s total_ms=%8.3f avg_us=%8.1f pct=%s\n" % (r["Name"][:96], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
print(open("gpurun_out/r06_summaries/train_step_kernel_stats.txt").read())
PY
rm -rf gpurun_out/prof_r06_train gpurun_out/prof_r06_stft_f16x3/trace gpurun_out/prof_r06_stft_f16x3/pmc*/
