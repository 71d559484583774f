#!/bin/bash
# Runs on the GPU box: the round's final evidence -- profiles (tools/profile_r2.sh) and the plain bench lines.
set -e
tools/profile_r2.sh > gpurun_out/prof_r2.log 2>&1
echo profiles done
python bench.py > gpurun_out/r2_cfg3_bench.json 2> gpurun_out/r2_cfg3_bench.err
echo cfg3 done
python bench.py --workload cfg2 > gpurun_out/r2_cfg2_bench.json 2> gpurun_out/r2_cfg2_bench.err
python bench.py --mode candidates --candidates 64 > gpurun_out/r2_cand64_bench.json 2> gpurun_out/r2_cand64_bench.err
python bench.py --mode candidates --candidates 256 --no-cpu-baseline > gpurun_out/r2_cand256_bench.json 2> gpurun_out/r2_cand256_bench.err
python bench.py --gpus 2 --no-cpu-baseline > gpurun_out/r2_2rank_bench.json 2> gpurun_out/r2_2rank_bench.err
for f in cfg3 cfg2 cand64 cand256 2rank; do cut -c1-260 gpurun_out/r2_${f}_bench.json; done
