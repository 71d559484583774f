#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_repeats.py tests/test_gpu_dist.py tests/test_gpu_dense.py -x -q -m gpu -s -k "not tandem_repeat_matches" > gpurun_out/r3_t3.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_t3.log
tail -5 gpurun_out/r3_t3.log
