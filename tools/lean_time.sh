#!/bin/bash
# dev: frontier phase times of a warm cfg3 step and of a 1/8 shard
echo "== cfg3"; PHMM_TRACE=1 timeout -k 10 300 python tools/trace_cfg3.py 2>&1 | grep "w0.*phase B\|w0.*sparse backward" | tail -2
echo "== shard of 8"; PHMM_TRACE=1 timeout -k 10 300 python tools/trace_shard.py 8 2>&1 | grep "w0.*phase B\|shard of" | tail -2
