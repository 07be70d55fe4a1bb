#!/bin/bash
# bench-shape CQT1992v2 forward under plan switches (environment): bash scripts/chain_env_sweep.sh "A=1" "B=2 C=3" ...
for e in "$@"; do
  for r in 1 2; do echo -n "[$e] "; env $e timeout 120 python bench.py --workload cqt --steps 100 --warmup 20 --extras 0 --cpu-baseline 0 --traffic off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
done
