#!/bin/bash
# the 448-thread 400-slot classes (forward + backward): against the generic kernels (1e-9), then rep20 / cfg3 / L10k /
# rep with and without them, then the kernel statistics of rep20
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 python tools/r3_wide_bits.py > gpurun_out/r3_wide_bits.log 2>&1
echo "bits rc=$?"; tail -1 gpurun_out/r3_wide_bits.log
for v in 1 0; do
  PHMM_NO_WIDE_CLASS=$v timeout -k 10 200 python bench.py --workload rep20 --steps 2 --warmup 1 > gpurun_out/r3_wide_rep20_$v.json 2> gpurun_out/r3_wide_rep20_$v.err || exit 1
  cut -c1-200 gpurun_out/r3_wide_rep20_$v.json
done
for v in 1 0 1 0; do
  PHMM_NO_WIDE_CLASS=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 > gpurun_out/r3_wide_cfg3_$v.json 2> gpurun_out/r3_wide_cfg3_$v.err || exit 1
  cut -c1-200 gpurun_out/r3_wide_cfg3_$v.json
done
for v in 1 0; do
  PHMM_NO_WIDE_CLASS=$v timeout -k 10 200 python bench.py --workload rep --steps 3 --warmup 1 > gpurun_out/r3_wide_rep_$v.json 2> gpurun_out/r3_wide_rep_$v.err || exit 1
  cut -c1-200 gpurun_out/r3_wide_rep_$v.json
done
rm -rf gpurun_out/r3_wide_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3_wide_prof -o rep20 --output-format csv -- python3 bench.py --workload rep20 --steps 2 --warmup 1 > gpurun_out/r3_wide_rep20_prof.log 2>&1
f=$(find gpurun_out/r3_wide_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-150
