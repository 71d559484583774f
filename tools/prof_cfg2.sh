#!/bin/bash
# dev: kernel-trace stats + SQ counters of the cfg2 dense path at the given knobs (env passes through)
set -e
TAG=${1:-a}
OUT=$PWD/gpurun_out/cfg2prof_$TAG
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- python3 $REPO/bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d $OUT/p1 -- python3 $REPO/bench.py --workload cfg2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/p1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU -d $OUT/p2 -- python3 $REPO/bench.py --workload cfg2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/p2.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2"):
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
head -12 $OUT/stats/*/*kernel_stats.csv 2>/dev/null || find $OUT/stats -name "*kernel_stats.csv" -exec head -12 {} \;
find $OUT -name '*kernel_trace.csv' -size +1M -delete
find $OUT -name '*counter_collection.csv' -size +1M -delete
