mkdir -p gpurun_out/r02h
(time CFD_FULL_ORACLE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s) > gpurun_out/r02h/fullsize_full_oracle.log 2>&1; grep -E "nMSE|passed|failed|real" gpurun_out/r02h/fullsize_full_oracle.log | tail -40
