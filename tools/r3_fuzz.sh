#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 500 python tools/fuzz_parity.py 150 11 > gpurun_out/r3_fuzz11.log 2>&1; echo "seed 11 rc=$?"; tail -1 gpurun_out/r3_fuzz11.log )
( timeout -k 10 500 python tools/fuzz_parity.py 150 12 > gpurun_out/r3_fuzz12.log 2>&1; echo "seed 12 rc=$?"; tail -1 gpurun_out/r3_fuzz12.log )
( timeout -k 10 400 python tools/fuzz_parity.py 24 13 -1 medium > gpurun_out/r3_fuzz13m.log 2>&1; echo "medium seed 13 rc=$?"; tail -1 gpurun_out/r3_fuzz13m.log )
