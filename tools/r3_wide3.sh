#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 200 python bench.py --workload rep20 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_wide3_rep20_a.json 2> gpurun_out/r3_wide3_rep20_a.err || exit 1
cut -c1-200 gpurun_out/r3_wide3_rep20_a.json
timeout -k 10 200 python bench.py --workload rep20 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_wide3_rep20_b.json 2> gpurun_out/r3_wide3_rep20_b.err || exit 1
cut -c1-200 gpurun_out/r3_wide3_rep20_b.json
rm -rf gpurun_out/r3_wide_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3_wide_prof -o rep20 --output-format csv -- python3 bench.py --workload rep20 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_wide_rep20_prof.log 2>&1
f=$(find gpurun_out/r3_wide_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-150
