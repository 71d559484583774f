"""dev script: PHMM_TRACE phase times of one warm cfg3 generate_mappings step"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dbgphmm_amd as D
arrays, reads, w = bench.build_workload("cfg3", 0)
gm = D.PHMMModel(arrays)
rc = D.ReadCollection(reads)
for it in range(3):
    print("=== call", it, file=sys.stderr, flush=True)
    mp, nf = gm.generate_mappings(rc, None, True)
