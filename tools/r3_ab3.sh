#!/bin/bash
PHMM_AMD_LIB=$PWD/_build_v0/lib.so timeout -k 10 200 python tools/r3_diag_det2.py /tmp/ref2.npz 2>&1 | tail -3
PHMM_AMD_LIB=$PWD/_build_vb/lib.so timeout -k 10 200 python tools/r3_diag_det2.py /tmp/ref2.npz 2>&1 | tail -80
