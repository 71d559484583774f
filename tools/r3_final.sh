#!/bin/bash
# round 3, final build: bench lines, profiles, same-box A/B, shard traces
mkdir -p gpurun_out/prof_r3
bash tools/r3_benchlines.sh > gpurun_out/r3_benchlines.log 2>&1
tail -8 gpurun_out/r3_benchlines.log
bash tools/profile_r3.sh > gpurun_out/r3_prof_full.log 2>&1
echo "profile rc=$?"
bash tools/r3_ab_bench.sh > gpurun_out/r3_ab_bench.log 2>&1
cat gpurun_out/r3_ab_bench.log
bash tools/r3_shard.sh > gpurun_out/r3_shard8.log 2>&1
head -20 gpurun_out/r3_shard8.log
