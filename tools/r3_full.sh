#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/r3_t_all.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_t_all.log
tail -4 gpurun_out/r3_t_all.log
