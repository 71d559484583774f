#!/bin/bash
set -e
OUT=$PWD/gpurun_out/probe_pmc
REPO=$PWD
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/f -- $REPO/tools/stream_probe > $OUT/f.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/w -- $REPO/tools/stream_probe > $OUT/w.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
for p,c in (("f","FETCH_SIZE"),("w","WRITE_SIZE")):
    agg=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%p, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==c: agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(c, k, len(v), "%.4g KB avg"%(sum(v)/len(v)))
PY
