#!/bin/bash
# one GPU call: full -m gpu suite, then the bench lines of this round
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2_gputests.log 2>&1; tail -4 gpurun_out/r2_gputests.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/r2_bench_cfg3.json 2> gpurun_out/r2_bench_cfg3.err || { tail -5 gpurun_out/r2_bench_cfg3.err; exit 1; }
timeout -k 10 300 python bench.py --workload cfg2 --steps 3 --warmup 1 > gpurun_out/r2_bench_cfg2.json 2> gpurun_out/r2_bench_cfg2.err || { tail -5 gpurun_out/r2_bench_cfg2.err; exit 1; }
for c in 1 16 64 256; do
  timeout -k 10 300 python bench.py --mode candidates --candidates $c --steps 3 --warmup 1 $( [ $c != 64 ] && echo --no-cpu-baseline ) > gpurun_out/r2_bench_cand$c.json 2> gpurun_out/r2_bench_cand$c.err || { tail -5 gpurun_out/r2_bench_cand$c.err; exit 1; }
done
timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r2_bench_2rank.json 2> gpurun_out/r2_bench_2rank.err || { tail -5 gpurun_out/r2_bench_2rank.err; exit 1; }
python3 - <<PY
import json
for n in ("cfg3","cfg2","cand1","cand16","cand64","cand256","2rank"):
    j=json.loads(open("gpurun_out/r2_bench_%s.json"%n).read().strip().splitlines()[-1])
    print(n, "value %.4g %s" % (j["value"], j["unit"]), "ms %.2f" % j["ms_per_step"], "frac %.3f" % j["roofline"]["frac"], "n_gpus", j["n_gpus"], {k:v for k,v in j.get("cpu_baseline",{}).items() if k in ("value","max_abs_dlogp","cores")}, {k:j["config"].get(k) for k in ("first_call_ms","cold_hint_ms","per_rank_ms","all_reduce_ms_per_step","hinted_forward_ms") if k in j["config"]})
PY
