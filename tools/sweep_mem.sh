#!/bin/bash
for cfg in "0.8 20" "0.9 20" "0.9 18" "0.92 18" "0.92 17" "0.9 16"; do
  set -- $cfg
  echo "== mem fraction $1 warm cols $2"
  PHMM_MEM_FRACTION=$1 PHMM_WARM_COLS=$2 PHMM_TRACE=1 timeout -k 10 200 python bench.py --no-cpu-baseline 2> gpurun_out/sweep_mem_err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.1f  bwd %.0f GB/s (%.0f us)  fwd %.0f GB/s' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'], d['roofline']['fwd_step']['achieved']))"
  grep -c "deferred" gpurun_out/sweep_mem_err.log; grep -c "dense warm-up" gpurun_out/sweep_mem_err.log
done
