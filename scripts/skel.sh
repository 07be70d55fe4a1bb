#!/bin/bash
# build experiments/stft_skeleton/skeleton and run it on the GPU box; log -> gpurun_out/$1/skeleton.log
set -e
cd /root/repo
tag=${1:-skel}
( cd experiments/stft_skeleton && hipcc --offload-arch=gfx950 -O3 -w $SKELFLAGS skeleton.hip -o skeleton )
/usr/local/graft/bin/gpurun --timeout 600 -- "mkdir -p gpurun_out/$tag; cd experiments/stft_skeleton && timeout 120 ./skeleton > ../../gpurun_out/$tag/skeleton.log 2>&1; echo rc \$?" > gpurun_out/${tag}_stdout.log 2>&1 || true
cat gpurun_out/$tag/skeleton.log; tail -2 gpurun_out/${tag}_stdout.log
