# quick GPU check of the FFT route: parity tests of the FFT path + the bench line
cd /root/repo
tag=${1:-q}
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_gpu_fft.py -x -q -m gpu > gpurun_out/$tag/pytest_fft.log 2>&1
tail -3 gpurun_out/$tag/pytest_fft.log
timeout 300 python bench.py --steps 100 --warmup 20 > gpurun_out/$tag/bench.log 2>&1
python - <<PY
import json
d = json.loads(open("gpurun_out/$tag/bench.log").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel", d["roofline"]["dominant_kernel"])
print({k: v.get("ms_per_step") for k, v in d["extra"].items()})
PY
