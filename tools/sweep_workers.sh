#!/bin/bash
# dev: cfg3 generate_mappings time for a few pipeline settings
for cfg in "1 0" "2 0" "3 0" "3 6" "3 4" "4 4" "4 6" "2 8"; do
  set -- $cfg
  echo "== workers $1 chunk_groups $2"
  PHMM_WORKERS=$1 PHMM_CHUNK_GROUPS=$2 timeout -k 10 200 python tools/try_cfg.py 2>&1 | grep "generate_mappings\|sparse full_prob"
done
