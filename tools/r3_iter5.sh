#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_repeats.py -x -q -m gpu -s -k "long_reads_on_a_short" 2>&1 | tail -12
