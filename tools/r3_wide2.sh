#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/r3_wide_bits.py 12 > gpurun_out/r3_wide_bits.log 2>&1
echo "bits rc=$?"; tail -1 gpurun_out/r3_wide_bits.log; grep -c " ok$" gpurun_out/r3_wide_bits.log; head -4 gpurun_out/r3_wide_bits.log | cut -c1-250
for i in 1 2; do
timeout -k 10 200 python bench.py --workload rep20 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r3_wide2_rep20.json 2> gpurun_out/r3_wide2_rep20.err || exit 1
cut -c1-200 gpurun_out/r3_wide2_rep20.json
done
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_wide2_cfg3_0.json 2> gpurun_out/r3_wide2_cfg3_0.err || exit 1
cut -c1-200 gpurun_out/r3_wide2_cfg3_0.json
