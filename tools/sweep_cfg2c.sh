#!/bin/bash
out=gpurun_out/r2_cfg2_sweep4.log
: > $out
for w in 16 64; do
for k in 1 3; do
  for npt in 4 8 12 16; do
    echo "== W=$w K=$k NPT=$npt" >> $out
    PHMM_DENSE_W=$w PHMM_DENSE_STREAMS=$k PHMM_DENSE_NPT=$npt timeout -k 10 120 python bench.py --workload cfg2 --steps 3 --warmup 1 --no-cpu-baseline >> $out 2>&1 || exit 1
  done
done
done
python3 tools/show_sweep.py $out
