#!/bin/bash
timeout -k 10 200 python tools/r3_diag_det.py cfg3 1000 2>&1 | grep -v "^$" | tail -6
PHMM_NO_WIDE_HANDOVER=1 timeout -k 10 200 python tools/r3_diag_det.py cfg3 10000 2>&1 | grep -v "^$" | tail -6
timeout -k 10 200 python tools/r3_diag_det.py rep 0 2>&1 | grep -v "^$" | tail -6
PHMM_AMD_LIB=$PWD/_build_v0/lib.so timeout -k 10 200 python tools/r3_diag_det.py rep 0 2>&1 | grep -v "^$" | tail -6
