#!/bin/bash
out=gpurun_out/r2_workers.log
: > $out
for cfg in "1 0" "2 32" "2 21" "2 16" "3 21" "3 11"; do
  set -- $cfg
  echo "== WORKERS=$1 CHUNK_GROUPS=$2" >> $out
  PHMM_WORKERS=$1 PHMM_CHUNK_GROUPS=$2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline >> $out 2>&1 || exit 1
done
python3 tools/show_sweep.py $out
