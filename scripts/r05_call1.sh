cd /root/repo
mkdir -p gpurun_out/c1
( cd experiments/valu_rate && timeout 120 ./valu_rate ) > gpurun_out/c1/valu_rate.log 2>&1
timeout 300 python scripts/ref_gpu_path_miss.py > gpurun_out/c1/ref_gpu_path_miss.log 2>&1
timeout 300 python bench.py --steps 50 --warmup 10 > gpurun_out/c1/bench.log 2>&1
tail -5 gpurun_out/c1/valu_rate.log; tail -20 gpurun_out/c1/ref_gpu_path_miss.log; tail -c 600 gpurun_out/c1/bench.log
