#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r3_t_all.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r3_t_all.log
tail -4 gpurun_out/r3_t_all.log
[ $rc -ne 0 ] && exit $rc
( timeout -k 10 330 python tools/fuzz_parity.py 120 22 > gpurun_out/r3_fuzz22.log 2>&1; echo "seed 22 rc=$?"; tail -1 gpurun_out/r3_fuzz22.log )
