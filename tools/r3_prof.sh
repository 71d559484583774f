#!/bin/bash
PHMM_AMD_LIB=$PWD/_build_vprof/lib.so PHMM_NO_WIDE_HANDOVER=1 timeout -k 10 200 python tools/r3_diag_det.py cfg3 10000 2>&1 | grep "prof" | sort | uniq -c | sort -rn | head -8
