"""dev script: one cfg3 shard of an N-GPU strong-scaling run on this GPU (timing of rank 0's work)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dbgphmm_amd as D
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
arrays, reads, w = bench.build_workload("cfg3", 0, world, "strong")
gm = D.PHMMModel(arrays)
rc = D.ReadCollection(reads)
for it in range(4):
    print("=== call", it, file=sys.stderr, flush=True)
    t = time.perf_counter()
    mp, nf = gm.generate_mappings(rc, None, True)
    print("shard of", world, "reads", len(reads), "ms %.1f" % ((time.perf_counter() - t) * 1e3), flush=True)
