#!/bin/bash
# dev iteration: parity of the sparse flow, then timing of the frontier phases
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py -x -q -m gpu 2>&1 | tail -3
PHMM_AMD_LIB=$PWD/_build_vprof/lib.so PHMM_NO_WIDE_HANDOVER=1 timeout -k 10 200 python tools/r3_diag_det.py cfg3 10000 2>&1 | grep "prof:" | sort | uniq -c | sort -rn | head -4
PHMM_TRACE=1 PHMM_NO_WIDE_HANDOVER=1 timeout -k 10 200 python tools/r3_diag_det.py cfg3 10000 2>&1 | grep "phase B\|sparse backward\|call\|identical\|differ" | tail -12
PHMM_TRACE=1 timeout -k 10 200 python tools/r3_diag_det.py cfg3 0 2>&1 | grep "w0.*phase B\|w0.*sparse backward\|call\|identical\|differ" | tail -12
