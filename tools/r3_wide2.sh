#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_sparse.py tests/test_abi_host.py -x -q -m gpu > gpurun_out/r3_wide2_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r3_wide2_tests.log | cut -c1-200
for i in 1 2; do
timeout -k 10 200 python bench.py --workload rep20 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_wide2_rep20_$i.json 2> gpurun_out/r3_wide2_rep20.err || exit 1
python -c "
import json
d=json.loads(open('gpurun_out/r3_wide2_rep20_$i.json').read().strip().splitlines()[-1]); c=d['config']
print('rep20', '%.1f ms/step' % d['ms_per_step'], 'first %.0f' % c['first_call_ms'], 'cold %.1f' % c['cold_hint_ms'], 'spin', c['spin_up_steps'])"
done
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_wide2_cfg3_0.json 2> gpurun_out/r3_wide2_cfg3_0.err || exit 1
cut -c1-200 gpurun_out/r3_wide2_cfg3_0.json
