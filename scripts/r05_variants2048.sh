# A/B of n_fft = 2048 variants (libmispec_NAME.so) + phase clocks: VARIANTS="a,b" STAMPED="x,y" bash scripts/r05_variants2048.sh tag
cd /root/repo
tag=${1:-vv}
mkdir -p gpurun_out/$tag
VARIANTS=$VARIANTS timeout 400 python scripts/fft2048_variants.py > gpurun_out/$tag/variants.log 2>&1
for v in $(echo $STAMPED | tr ',' ' '); do VARIANT=$v timeout 200 python scripts/fft_stamps.py > gpurun_out/$tag/stamps_$v.log 2>&1; done
cat gpurun_out/$tag/variants.log
