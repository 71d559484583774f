#!/bin/bash
PHMM_TRACE=1 timeout -k 10 300 python tools/trace_shard.py 8 2>&1 | tail -60
