#!/bin/bash
# SQ counters of the wide kernels on rep20 (one cold call) + a fuzz sweep of the final build
mkdir -p gpurun_out/prof_r3
OUT=$PWD/gpurun_out/prof_r3
REPO=$PWD
( timeout -k 10 500 python tools/fuzz_parity.py 150 21 > gpurun_out/r3_fuzz21.log 2>&1; echo "seed 21 rc=$?"; tail -1 gpurun_out/r3_fuzz21.log )
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $OUT/rep20_sq_p1 -- python3 $REPO/bench.py --workload rep20 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/rep20_sq_p1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/rep20_sq_p2 -- python3 $REPO/bench.py --workload rep20 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/rep20_sq_p2.log 2>&1
cd $REPO
python3 - > $OUT/rep20_sq_counters.txt <<PY
import csv, glob, collections
for p in ("rep20_sq_p1","rep20_sq_p2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%p, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")
            if "phmm::wide" not in k and "phmm::lean" not in k: continue
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES","SQ_WAVE_CYCLES"): cnt[k]+=1
    for k,v in sorted(agg.items()):
        print(p,k,cnt[k]," ".join("%s=%.4g"%(a,b) for a,b in sorted(v.items())))
PY
cat $OUT/rep20_sq_counters.txt
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
