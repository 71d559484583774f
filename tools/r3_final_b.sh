#!/bin/bash
# round 3, final build, part B: first-process warm-up probe, profiles + shard traces
mkdir -p gpurun_out/prof_r3
python bench.py --steps 5 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('first process, warmup 6:', '%.1f ms/step' % d['ms_per_step'])"
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('second process, warmup 1:', '%.1f ms/step' % d['ms_per_step'])"
bash tools/profile_r3.sh > gpurun_out/r3_prof_full.log 2>&1
echo "profile rc=$?"
bash tools/r3_shard.sh > gpurun_out/r3_shard8.log 2>&1
head -20 gpurun_out/r3_shard8.log
