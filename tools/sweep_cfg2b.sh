#!/bin/bash
out=gpurun_out/r2_cfg2_sweep3.log
: > $out
for lib in "" v1 v2; do
for w in 16 64; do
for k in 1 3; do
  for npt in 8 12 16; do
    echo "== LIB=$lib W=$w K=$k NPT=$npt" >> $out
    L=$PWD/dbgphmm_amd/libphmm_amd${lib:+_$lib}.so
    PHMM_AMD_LIB=$L PHMM_DENSE_W=$w PHMM_DENSE_STREAMS=$k PHMM_DENSE_NPT=$npt timeout -k 10 120 python bench.py --workload cfg2 --steps 3 --warmup 1 --no-cpu-baseline >> $out 2>&1 || exit 1
  done
done
done
done
