#!/bin/bash
# round 3, final build, part A: bench lines (+ same-box A/B against the round-2 library with "ab" as argument)
mkdir -p gpurun_out/prof_r3
bash tools/r3_benchlines.sh > gpurun_out/r3_benchlines.log 2>&1
tail -8 gpurun_out/r3_benchlines.log
if [ "$1" = "ab" ]; then
bash tools/r3_ab_bench.sh > gpurun_out/r3_ab_bench.log 2>&1
cat gpurun_out/r3_ab_bench.log
fi
