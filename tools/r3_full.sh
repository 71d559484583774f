#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_repeats.py tests/test_gpu_scale.py tests/test_gpu_sparse.py -x -q -m gpu -s > gpurun_out/r3_t_all.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_t_all.log
tail -4 gpurun_out/r3_t_all.log
