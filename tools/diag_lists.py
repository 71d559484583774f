"""dev script: list-length statistics of the cfg3 mappings (which capacity class do the reads fall in?)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import dbgphmm_amd as D
arrays, reads, w = bench.build_workload("cfg3", 0)
gm = D.PHMMModel(arrays)
rc = D.ReadCollection(reads)
mp, nf = gm.generate_mappings(rc, None, True)
po, nd, lp = mp.arrays()
cnt = np.diff(po.astype(np.int64))
off = rc.offsets.astype(np.int64)
rmax = np.array([cnt[off[r]:off[r + 1]].max() for r in range(len(reads))])
print("reads", len(reads), "max list per read: <=8", (rmax <= 8).sum(), "<=16", (rmax <= 16).sum(), "<=32", (rmax <= 32).sum(), "<=64", (rmax <= 64).sum(), "<=128", (rmax <= 128).sum(), "400", (rmax == 400).sum())
print("positions: mean", cnt.mean(), "hist", np.bincount(np.minimum(cnt, 70))[:70].tolist())
for p in (0, 1, 2, 5, 10, 15, 20, 30, 50):
    c = np.array([cnt[off[r] + p] for r in range(len(reads)) if off[r + 1] - off[r] > p])
    print("pos", p, "list len mean %.1f max %d" % (c.mean(), c.max()))
print("fraction of positions with list <= 8:", (cnt <= 8).mean(), "<=16", (cnt <= 16).mean())
import torch
for _ in range(2):
    torch.cuda.synchronize(); t = time.perf_counter(); gm.to_full_prob_reads(rc, mp); torch.cuda.synchronize(); print("hinted ms", (time.perf_counter() - t) * 1e3)
