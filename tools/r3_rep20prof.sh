#!/bin/bash
OUT=$PWD/gpurun_out/prof_r3; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/rep20_stats -o stats -- python3 $REPO/bench.py --workload rep20 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/rep20_bench_under_rocprof.json 2> $OUT/rep20_stats.err
find $OUT -name '*kernel_trace.csv' -delete
head -14 $OUT/rep20_stats/stats_kernel_stats.csv | cut -c1-170
