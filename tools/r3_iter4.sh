#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests/test_gpu_repeats.py -x -q -m gpu -k "tandem or long_reads_on" 2>&1 | tail -2
python bench.py --workload rep20 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep20', '%.1f ms/step' % d['ms_per_step'], 'cold %.1f' % d['config']['cold_hint_ms'])"
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3', '%.1f ms/step' % d['ms_per_step'], 'cold %.1f' % d['config']['cold_hint_ms'])"
