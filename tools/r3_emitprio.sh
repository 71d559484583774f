#!/bin/bash
OUT=$PWD/gpurun_out/prof_r3
REPO=$PWD
mkdir -p $OUT
for v in 0 1; do
  if [ $v = 1 ]; then export PHMM_EMIT_LOW_PRIORITY=1; fi
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('low_priority=$v', d['ms_per_step'], d['roofline']['avg_launch_us'], d['config']['cold_hint_ms'])"
done
unset PHMM_EMIT_LOW_PRIORITY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/cfg3_stats -o stats -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cfg3_bench_under_rocprof.json 2> $OUT/cfg3_stats.err
find $OUT -name '*kernel_trace.csv' -delete
head -12 $OUT/cfg3_stats/stats_kernel_stats.csv | cut -c1-150
