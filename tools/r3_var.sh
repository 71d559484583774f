#!/bin/bash
for i in 1 2 3; do
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i', '%.1f ms/step' % d['ms_per_step'], 'cold %.1f' % d['config']['cold_hint_ms'], 'first %.0f' % d['config']['first_call_ms'], 'bwd %.0f us fwd %.0f us' % (d['roofline']['avg_launch_us'], d['roofline']['fwd_step']['avg_launch_us']))"
done
for i in 1 2; do
  PHMM_EMIT_LOW_PRIORITY=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('low-prio run $i', '%.1f ms/step' % d['ms_per_step'], 'cold %.1f' % d['config']['cold_hint_ms'], 'bwd %.0f us' % d['roofline']['avg_launch_us'])"
done
python bench.py --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default cmd', '%.1f ms/step' % d['ms_per_step'], 'cold %.1f' % d['config']['cold_hint_ms'])"
