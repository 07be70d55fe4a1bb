#!/bin/bash
# the measurements DESIGN.md 3.14 cites, in one GPU call: microbenchmarks, ablation builds, phase clock of one unit
cd /root/repo
O=gpurun_out/chain_evidence; mkdir -p $O
for b in mfma_rate dma_rate corun; do
  (cd experiments/chain && hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm $b.hip -o $b 2>/dev/null; timeout 120 ./$b) > $O/chain_micro_$b.log 2>&1
done
timeout 900 bash scripts/chain_ablate.sh "0 1 2 4 8 12 16" > $O/chain_ablations.log 2>&1
timeout 300 bash scripts/chain_stamps.sh 64 90 > $O/chain_phase_clock.log 2>&1
timeout 120 python scripts/chain_check.py > $O/chain_check.log 2>&1
tail -3 $O/*.log
