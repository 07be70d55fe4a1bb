#!/bin/bash
# PMC passes over the STFT step on the FFT path (counters only; one small group per pass).
# usage (GPU box): bash scripts/fft_pmc.sh  -> gpurun_out/fft_pmc/*.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fft_pmc; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --pmc-child stft --steps 5 > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY' > $OUT/pass_$i.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-50:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    if "at::" in k: continue
    print(k, {c: acc[k][c] / n[(k, c)] for c in acc[k]})
PY
  else
    tail -3 /tmp/pmc_$i.log > $OUT/pass_$i.txt
  fi
done
cat $OUT/pass_*.txt
