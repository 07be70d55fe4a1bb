# the full GPU suite + smoke + the bench line (what the driver runs at round end)
cd /root/repo
tag=${1:-full}
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/$tag/pytest_gpu.log 2>&1
tail -4 gpurun_out/$tag/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$tag/smoke.log 2>&1
tail -3 gpurun_out/$tag/smoke.log
timeout 600 python bench.py > gpurun_out/$tag/bench.log 2>&1
tail -c 1200 gpurun_out/$tag/bench.log
