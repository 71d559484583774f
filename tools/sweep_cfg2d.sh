#!/bin/bash
out=gpurun_out/r2_cfg2_sweep5.log
: > $out
for lib in "" w3 w2; do
for k in 3; do
  for npt in 8 12; do
    echo "== LIB=$lib K=$k NPT=$npt" >> $out
    L=$PWD/dbgphmm_amd/libphmm_amd${lib:+_$lib}.so
    PHMM_AMD_LIB=$L PHMM_DENSE_STREAMS=$k PHMM_DENSE_NPT=$npt timeout -k 10 120 python bench.py --workload cfg2 --steps 3 --warmup 1 --no-cpu-baseline >> $out 2>&1 || exit 1
  done
done
done
python3 tools/show_sweep.py $out
