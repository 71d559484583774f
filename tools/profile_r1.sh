#!/bin/bash
# Runs on the GPU box (from the repo root): kernel stats + PMC passes of the default bench and of
# the calibration tool.  Everything lands under gpurun_out/prof_$1/.
set -e
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/calib_fetch -- $REPO/tools/pmc_calib > $OUT/calib_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/calib_write -- $REPO/tools/pmc_calib > $OUT/calib_write.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/bench_fetch -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/bench_write -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1
cd $REPO
python3 tools/pmc_traffic.py --calib-fetch $OUT/calib_fetch --calib-write $OUT/calib_write --bench-fetch $OUT/bench_fetch --bench-write $OUT/bench_write --last-fraction 0.5 --out $OUT/pmc_traffic.json
# keep only the summaries (the raw traces are large)
find $OUT -name '*kernel_trace.csv' -delete
find $OUT/bench_fetch $OUT/bench_write -name "*counter_collection.csv" -delete
ls -R $OUT | head -50
