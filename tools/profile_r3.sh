#!/bin/bash
# Runs on the GPU box (from the repo root): round-3 evidence under gpurun_out/prof_r3/.
#   cfg3 (default bench): kernel stats, PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes, calibrated), SQ counters
#   cfg3 --read-len 10000, rep, rep20: kernel stats
set -e
OUT=$PWD/gpurun_out/prof_r3
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -x $REPO/tools/pmc_calib ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $REPO/tools/pmc_calib $REPO/tools/pmc_calib.hip
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/cfg3_stats -o stats -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cfg3_bench_under_rocprof.json 2> $OUT/cfg3_stats.err
echo cfg3 stats done
if [ "$1" != "quick" ]; then
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/calib_fetch -- $REPO/tools/pmc_calib > $OUT/calib_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/calib_write -- $REPO/tools/pmc_calib > $OUT/calib_write.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/bench_fetch -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/bench_write -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1
echo cfg3 pmc done
fi
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $OUT/sq_p1 -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq_p1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/sq_p2 -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq_p2.log 2>&1
echo cfg3 sq done
if [ "$1" != "quick" ]; then
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/cfg3_L10k_stats -o stats -- python3 $REPO/bench.py --read-len 10000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cfg3_L10k_bench_under_rocprof.json 2> $OUT/cfg3_L10k_stats.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/rep_stats -o stats -- python3 $REPO/bench.py --workload rep --steps 3 --warmup 1 --no-cpu-baseline > $OUT/rep_bench_under_rocprof.json 2> $OUT/rep_stats.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/rep20_stats -o stats -- python3 $REPO/bench.py --workload rep20 --steps 2 --warmup 2 --no-cpu-baseline > $OUT/rep20_bench_under_rocprof.json 2> $OUT/rep20_stats.err
echo long-read stats done
fi
cd $REPO
if [ "$1" != "quick" ]; then
python3 tools/pmc_traffic.py --calib-fetch $OUT/calib_fetch --calib-write $OUT/calib_write --bench-fetch $OUT/bench_fetch --bench-write $OUT/bench_write --last-fraction 0.5 --out $OUT/pmc_traffic.json
fi
python3 - > $OUT/cfg3_sq_counters.txt <<PY
import csv, glob, collections
for p in ("sq_p1","sq_p2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%p, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")
            if "phmm::" not in k: continue
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES","SQ_WAVE_CYCLES"): cnt[k]+=1
    for k,v in sorted(agg.items()):
        print(p,k,cnt[k]," ".join("%s=%.4g"%(a,b) for a,b in sorted(v.items())))
PY
cat $OUT/cfg3_sq_counters.txt | grep "lean\|emit" || true
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*agent_info.csv" -delete
ls -R $OUT | head -50
