#!/bin/bash
# One gpurun call for the convolution stack: parity tests that touch conv kernels, per-layer timings, U-Net / ResNet steps.
set -u
TAG=${1:-conv}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0  # a GPU memory fault otherwise spends minutes writing a core file
timeout 900 python -m pytest tests -m gpu -q -k "conv or unet or resnet or UNet or ResNet" "$@" > $OUT/pytest_conv.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_conv.log
tail -15 $OUT/pytest_conv.log
grep -q "pytest rc=0" $OUT/pytest_conv.log || { echo "parity tests failed: skipping the timings"; exit 1; }
timeout 300 python tools/bench_conv_layers.py > $OUT/conv_layers.txt 2>&1; cat $OUT/conv_layers.txt
timeout 300 python tools/bench_unet.py > $OUT/unet.txt 2>&1; tail -25 $OUT/unet.txt
timeout 300 python tools/bench_resnet.py > $OUT/resnet.txt 2>&1; tail -15 $OUT/resnet.txt
