#!/bin/bash
PHMM_TRACE=1 timeout -k 10 200 python tools/r3_diag_det.py cfg3 10000 2>&1 | grep -v "remaining lanes\|chunk:\|prof" | tail -75
