#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 python tools/r3_wide_bits.py 16 > gpurun_out/r3_wide_bits.log 2>&1
echo "bits rc=$?"; tail -1 gpurun_out/r3_wide_bits.log; grep -c " ok$" gpurun_out/r3_wide_bits.log; head -4 gpurun_out/r3_wide_bits.log | cut -c1-250
timeout -k 10 200 python bench.py --workload rep20 --steps 2 --warmup 2 --no-cpu-baseline > gpurun_out/r3_wide2_rep20.json 2> gpurun_out/r3_wide2_rep20.err || exit 1
cut -c1-200 gpurun_out/r3_wide2_rep20.json
for v in 0 1; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_wide2_cfg3_$v.json 2> gpurun_out/r3_wide2_cfg3_$v.err || exit 1
  cut -c1-200 gpurun_out/r3_wide2_cfg3_$v.json
done
rm -rf gpurun_out/r3_wide_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3_wide_prof -o rep20 --output-format csv -- python3 bench.py --workload rep20 --steps 2 --warmup 2 --no-cpu-baseline > gpurun_out/r3_wide_rep20_prof.log 2>&1
f=$(find gpurun_out/r3_wide_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -5 "$f" | cut -c1-150
