#!/bin/bash
# round 3, first GPU pass: repeat parity tests + bench lines of the new workloads
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_repeats.py -x -q -m gpu -s > gpurun_out/r3_rep_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_rep_tests.log
tail -5 gpurun_out/r3_rep_tests.log
for wl in rep rep20; do
  timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 > gpurun_out/r3_${wl}_bench0.json 2> gpurun_out/r3_${wl}_bench0.err
  echo "$wl rc=$?"; tail -c 600 gpurun_out/r3_${wl}_bench0.json
done
timeout -k 10 300 python bench.py --workload cfg3 --read-len 10000 --steps 3 --warmup 1 > gpurun_out/r3_cfg3_L10k_bench0.json 2> gpurun_out/r3_cfg3_L10k_bench0.err
echo "cfg3 L10k rc=$?"; tail -c 600 gpurun_out/r3_cfg3_L10k_bench0.json
