#!/bin/bash
# dev: cfg3 bench for a few pipeline settings
for cfg in "1 0" "2 0" "2 16" "2 27" "3 0" "3 12" "3 16" "4 8"; do
  set -- $cfg
  echo "== workers $1 chunk_groups $2"
  PHMM_WORKERS=$1 PHMM_CHUNK_GROUPS=$2 timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.1f  bwd %.0f GB/s (%.0f us)  fwd %.0f GB/s' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'], d['roofline']['fwd_step']['achieved']))"
done
