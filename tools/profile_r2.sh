#!/bin/bash
# Runs on the GPU box (from the repo root): round-2 evidence.  Kernel stats + PMC passes of the default bench (cfg3),
# kernel stats of cfg2, kernel stats + SQ counters of the candidate batch.  Everything lands under gpurun_out/prof_r2/.
set -e
OUT=$PWD/gpurun_out/prof_r2
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/cfg3_stats -o stats -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cfg3_bench_under_rocprof.json 2> $OUT/cfg3_stats.err
echo cfg3 stats done
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/calib_fetch -- $REPO/tools/pmc_calib > $OUT/calib_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/calib_write -- $REPO/tools/pmc_calib > $OUT/calib_write.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/bench_fetch -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/bench_write -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1
echo cfg3 pmc done
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/cfg2_stats -o stats -- python3 $REPO/bench.py --workload cfg2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cfg2_bench_under_rocprof.json 2> $OUT/cfg2_stats.err
echo cfg2 stats done
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/cand_stats -o stats -- python3 $REPO/bench.py --mode candidates --candidates 64 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cand64_bench_under_rocprof.json 2> $OUT/cand_stats.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/cand_p1 -- python3 $REPO/bench.py --mode candidates --candidates 64 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/cand_p1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $OUT/cand_p2 -- python3 $REPO/bench.py --mode candidates --candidates 64 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/cand_p2.log 2>&1
echo candidates done
cd $REPO
python3 tools/pmc_traffic.py --calib-fetch $OUT/calib_fetch --calib-write $OUT/calib_write --bench-fetch $OUT/bench_fetch --bench-write $OUT/bench_write --last-fraction 0.5 --out $OUT/pmc_traffic.json
python3 - > $OUT/cand64_sq_counters.txt <<PY
import csv, glob, collections
for p in ("cand_p1","cand_p2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%p, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")
            if "phmm::hinted" not in k: continue
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES","SQ_WAIT_ANY"): cnt[k]+=1
    for k,v in sorted(agg.items()):
        print(p,k,"launches",cnt[k]," ".join("%s=%.4g"%(a,b) for a,b in sorted(v.items())))
PY
cat $OUT/cand64_sq_counters.txt
# keep only the summaries (the raw traces are large)
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name "*counter_collection.csv" -delete
ls -R $OUT | head -60
