#!/bin/bash
for w in 1 2 3 4; do
  echo "== workers $w"; PHMM_WORKERS=$w PHMM_PIPELINE_MIN_GROUPS=2 timeout -k 10 300 python tools/trace_shard.py 8 2>&1 | grep "shard of" | tail -2
done
for w in 2 4; do
  echo "== shard of 4, workers $w"; PHMM_WORKERS=$w PHMM_PIPELINE_MIN_GROUPS=2 timeout -k 10 300 python tools/trace_shard.py 4 2>&1 | grep "shard of" | tail -2
done
