# the CQT1992v2 fp32 measurements of round 5 -> gpurun_out/r05_cqt/
cd /root/repo
mkdir -p gpurun_out/r05_cqt
timeout 300 python scripts/cqt1992_verbatim.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_cqt/cqt1992_verbatim.log
timeout 100 experiments/tap_order/mfma_order > gpurun_out/r05_cqt/mfma_order.log 2>&1
timeout 200 python scripts/cqt_fp32_t16.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_cqt/cqt_fp32_t16.log
timeout 200 python scripts/cqt_fp32_ablate.py 76 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_cqt/cqt_fp32_ablate.log
timeout 300 python scripts/cqt_fp32_quantization.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_cqt/cqt_fp32_quantization.log
timeout 200 python scripts/aligned_rows_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_cqt/stft_aligned_rows_probe.log
tail -n 30 gpurun_out/r05_cqt/*.log
