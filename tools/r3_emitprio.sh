#!/bin/bash
OUT=$PWD/gpurun_out/prof_r3
REPO=$PWD
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/cfg3_stats -o stats -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cfg3_bench_under_rocprof.json 2> $OUT/cfg3_stats.err
find $OUT -name '*kernel_trace.csv' -delete
grep "emit\|Name" $OUT/cfg3_stats/stats_kernel_stats.csv | cut -c1-160
