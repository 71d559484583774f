#!/bin/bash
mkdir -p gpurun_out
for v in v0 vf vb; do
  PHMM_AMD_LIB=$PWD/_build_$v/lib.so timeout -k 10 200 python tools/r3_diag_det.py cfg3 1000 /tmp/ref_cfg3.npz 2>&1 | grep -v "^$" | tail -8
done
PHMM_AMD_LIB=$PWD/dbgphmm_amd/libphmm_amd.so timeout -k 10 200 python tools/r3_diag_det.py cfg3 1000 /tmp/ref_cfg3.npz 2>&1 | tail -8
for v in v0 vf vb; do
  PHMM_AMD_LIB=$PWD/_build_$v/lib.so PHMM_NO_WIDE_HANDOVER=1 timeout -k 10 200 python tools/r3_diag_det.py cfg3 10000 2>&1 | grep -v "^$" | tail -5
done
