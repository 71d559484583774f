#!/bin/bash
PHMM_TRACE=1 timeout -k 10 300 python tools/r3_diag_det.py rep20 0 2>&1 | grep -v "remaining lanes\|chunk:" | tail -150
