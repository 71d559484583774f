#!/bin/bash
# dev: SQ counters of the kernels of one cfg3 step (serial mode)
set -e
OUT=$PWD/gpurun_out/sq_$1
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PHMM_WORKERS=1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $OUT/p1 -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/p1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/p2 -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/p2.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%p, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")
            if "phmm::" not in k: continue
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
            if r["Counter_Name"] in ("SQ_WAVES","SQ_WAVE_CYCLES"): cnt[k]+=1
    for k,v in sorted(agg.items()):
        print(p,k,cnt[k]," ".join("%s=%.4g"%(a,b) for a,b in sorted(v.items())))
PY
find $OUT -name '*.csv' -size +1M -delete
