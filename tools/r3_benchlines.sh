#!/bin/bash
# round-3 bench lines (with cpu_baseline) under gpurun_out/prof_r3/
OUT=$PWD/gpurun_out/prof_r3
mkdir -p $OUT
python bench.py --steps 5 --warmup 1 > $OUT/cfg3_bench.json 2> $OUT/cfg3_bench.err; echo cfg3 $?
python bench.py --read-len 10000 --steps 3 --warmup 1 > $OUT/cfg3_L10k_bench.json 2> $OUT/cfg3_L10k_bench.err; echo L10k $?
python bench.py --workload rep --steps 3 --warmup 1 > $OUT/rep_bench.json 2> $OUT/rep_bench.err; echo rep $?
python bench.py --workload rep20 --steps 3 --warmup 2 > $OUT/rep20_bench.json 2> $OUT/rep20_bench.err; echo rep20 $?
python bench.py --workload cfg2 --steps 3 --warmup 1 > $OUT/cfg2_bench.json 2> $OUT/cfg2_bench.err; echo cfg2 $?
python bench.py --mode candidates --candidates 64 --steps 3 --warmup 1 > $OUT/cand64_bench.json 2> $OUT/cand64_bench.err; echo cand64 $?
python bench.py --mode candidates --candidates 256 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cand256_bench.json 2> $OUT/cand256_bench.err; echo cand256 $?
for f in cfg3 cfg3_L10k rep rep20 cfg2 cand64 cand256; do python - <<PY
import json
d=json.loads(open('$OUT/${f}_bench.json').read().strip().splitlines()[-1])
print('$f', '%.4g %s' % (d['value'], d['unit']), '%.1f ms/step' % d['ms_per_step'], 'cold', d['config'].get('cold_hint_ms'), 'frac %.3f' % d['roofline']['frac'], 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('max_abs_dlogp'))
PY
done
