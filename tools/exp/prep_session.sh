#!/bin/bash
# One gpurun call: the prepared-weights route (cfd_conv2d_wprep_batch) and the one-rank RCCL exchange -- parity first, then the
# U-Net / ResNet steps with the route on and off.
set -u
TAG=${1:-prep}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_harness.py tests/test_gpu_dp.py \
    -m gpu -q -k "conv or unet or resnet or UNet or ResNet or prepared or rccl or graph" "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -25 $OUT/pytest.log
grep -q "pytest rc=0" $OUT/pytest.log || { echo "parity tests failed: skipping the timings"; exit 1; }
for prep in 1 0; do
    CFDBENCH_CONV_PREP=$prep timeout 300 python tools/bench_unet.py --graph --steps 30 > $OUT/unet_prep$prep.txt 2>&1; echo "== U-Net, prepared weights = $prep"; grep -E "ms_per_step|frames" $OUT/unet_prep$prep.txt | head -3
    CFDBENCH_CONV_PREP=$prep timeout 300 python tools/bench_resnet.py > $OUT/resnet_prep$prep.txt 2>&1; echo "== ResNet, prepared weights = $prep"; grep -E "ms_per_step|frames" $OUT/resnet_prep$prep.txt | head -3
done
tail -22 $OUT/unet_prep1.txt
