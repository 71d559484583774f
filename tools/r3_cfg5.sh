#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --workload cfg5 --steps 1 --warmup 1 > gpurun_out/r3_cfg5_mapping_bench.json 2> gpurun_out/r3_cfg5_mapping_bench.err
echo "mapping rc=$?"; tail -c 1500 gpurun_out/r3_cfg5_mapping_bench.json
timeout -k 10 400 python bench.py --workload cfg5 --mode candidates --candidates 64 --steps 2 --warmup 1 > gpurun_out/r3_cfg5_cand64_bench.json 2> gpurun_out/r3_cfg5_cand64_bench.err
echo "cand rc=$?"; tail -c 800 gpurun_out/r3_cfg5_cand64_bench.json
timeout -k 10 400 python bench.py --workload cfg5 --mode map_nodes --steps 2 --warmup 1 > gpurun_out/r3_cfg5_map_nodes_bench.json 2> gpurun_out/r3_cfg5_map_nodes_bench.err
echo "map_nodes rc=$?"; tail -c 800 gpurun_out/r3_cfg5_map_nodes_bench.json
