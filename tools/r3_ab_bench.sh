#!/bin/bash
# same box, same run: round-2 library (_build_v0) against the current one
for spec in "cfg3 0 5" "cfg3 10000 3" "rep 0 3" "rep20 0 2"; do
  set -- $spec
  for lib in _build_v0/lib.so dbgphmm_amd/libphmm_amd.so; do
    PHMM_AMD_LIB=$PWD/$lib python bench.py --workload $1 --read-len $2 --steps $3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 L=$2 $lib', '%.1f ms/step' % d['ms_per_step'], 'cold %.1f' % d['config']['cold_hint_ms'], 'bwd_step us', d['roofline'].get('avg_launch_us'))"
  done
done
