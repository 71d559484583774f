#!/bin/bash
# dev: one shard of an N-rank strong-scaling run with the chunk pipeline on/off
for world in 8 4; do
  for cfg in "1 0" "2 4" "2 2" "3 3" "2 8" "3 6"; do
    set -- $cfg
    echo "world $world workers $1 chunk_groups $2: $(PHMM_WORKERS=$1 PHMM_CHUNK_GROUPS=$2 PHMM_PIPELINE_MIN_GROUPS=2 timeout -k 10 200 python tools/trace_shard.py $world 2>/dev/null | tail -2 | tr '\n' ' ')"
  done
done
