#!/bin/bash
# copies the round-3 summaries out of gpurun_out/ (scratch) into profiles/ (tracked)
S=gpurun_out/prof_r3
P=profiles
cp $S/cfg3_bench.json $P/r3_cfg3_bench.json
cp $S/cfg3_bench_under_rocprof.json $P/r3_cfg3_bench_under_rocprof.json
cp $S/cfg3_stats/stats_kernel_stats.csv $P/r3_cfg3_kernel_stats.csv
cp $S/pmc_traffic.json $P/r3_cfg3_pmc_traffic.json
cp $S/cfg3_sq_counters.txt $P/r3_cfg3_sq_counters.txt
cp $S/cfg3_L10k_bench.json $P/r3_cfg3_L10k_bench.json
cp $S/cfg3_L10k_stats/stats_kernel_stats.csv $P/r3_cfg3_L10k_kernel_stats.csv
cp $S/rep_bench.json $P/r3_rep_bench.json
cp $S/rep_stats/stats_kernel_stats.csv $P/r3_rep_kernel_stats.csv
cp $S/rep20_bench.json $P/r3_rep20_bench.json
cp $S/rep20_stats/stats_kernel_stats.csv $P/r3_rep20_kernel_stats.csv
cp $S/cfg2_bench.json $P/r3_cfg2_bench.json
cp $S/cand64_bench.json $P/r3_cand64_bench.json
cp $S/cand256_bench.json $P/r3_cand256_bench.json
[ -f gpurun_out/r3_shard8.log ] && grep "shard of\|w0" gpurun_out/r3_shard8.log | tail -20 > $P/r3_shard8_trace.txt
ls $P | grep r3_
