#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_scale.py -x -q -m gpu > gpurun_out/r3_wide2_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r3_wide2_tests.log | cut -c1-200
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --workload rep20 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_wide2_rep20_1.json 2> gpurun_out/r3_wide2_rep20.err || exit 1
python -c "
import json
d=json.loads(open('gpurun_out/r3_wide2_rep20_1.json').read().strip().splitlines()[-1]); c=d['config']
print('rep20', '%.1f ms/step' % d['ms_per_step'], 'first %.0f' % c['first_call_ms'], 'cold %.1f' % c['cold_hint_ms'], 'spin', c['spin_up_steps'])"
( timeout -k 10 330 python tools/fuzz_parity.py 120 22 > gpurun_out/r3_fuzz22.log 2>&1; echo "seed 22 rc=$?"; tail -1 gpurun_out/r3_fuzz22.log )
