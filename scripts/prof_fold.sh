cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fold1/trace -o t -- python $R/bench.py --extras 0 --cpu-baseline 0 --traffic off --steps 100 --warmup 20 > $R/gpurun_out/prof_fold1_trace.log 2>&1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"
P4="GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
i=1
for P in "$P1" "$P4"; do
rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/prof_fold1/pmc$i -o pmc -- python $R/bench.py --steps 5 --warmup 2 --prewarm-ms 0 --extras 0 --cpu-baseline 0 --traffic off > $R/gpurun_out/prof_fold1_pmc$i.log 2>&1
i=$((i+1))
done
cd $R; python scripts/summarize_prof.py gpurun_out/prof_fold1 | grep -v "^  *void at::\|at::native" | head -80
