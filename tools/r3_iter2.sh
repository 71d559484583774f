#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -3
for spec in "cfg3 10000 3" "rep 0 3" "cfg3 0 5"; do
  set -- $spec
  python bench.py --workload $1 --read-len $2 --steps $3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 L=$2', '%.1f ms/step' % d['ms_per_step'], 'cold %.1f' % d['config']['cold_hint_ms'], d['config']['frontier']['wide_frontier_reads'])"
done
