#!/bin/bash
# One gpurun call: the matrix-pipe ConvTranspose2d kernels (convt6.hip) -- parity, random shapes, then the U-Net step with them on / off.
set -u
TAG=${1:-convt}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -m gpu -q -k "convt or unet or UNet or prepared" "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -12 $OUT/pytest.log
grep -q "pytest rc=0" $OUT/pytest.log || { echo "parity tests failed: skipping the timings"; exit 1; }
timeout 300 python tools/exp/fuzz_convt_gpu.py 150 > $OUT/fuzz.txt 2>&1; tail -5 $OUT/fuzz.txt
for mf in 1 0; do
    CFD_CONVT_MFMA=$mf timeout 300 python tools/bench_unet.py --graph --steps 30 > $OUT/unet_mfma$mf.txt 2>&1; echo "== U-Net, convT on the matrix pipe = $mf"; grep -E "ms_per_step" $OUT/unet_mfma$mf.txt | head -2
done
