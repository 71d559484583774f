"""dev: where do two builds / two calls differ (lists printed)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import dbgphmm_amd as D
ref = sys.argv[1]
arrays, reads, w = bench.build_workload("cfg3", 0, 1, "strong", 0)
gm = D.PHMMModel(arrays)
rc = D.ReadCollection(reads)
off = rc.offsets.astype(np.int64)
z = np.load(ref) if os.path.exists(ref) else None
for it in range(4):
    mp, nf = gm.generate_mappings(rc, None, True)
    po, nd, lp = mp.arrays()
    if z is None:
        np.savez(ref, po=po, nd=nd, lp=lp)
        print("saved")
        break
    zp, zn, zl = z["po"], z["nd"], z["lp"]
    assert np.array_equal(zp, po)
    d = np.flatnonzero((zn != nd) | (np.abs(zl - lp) > 1e-9))
    pos = np.unique(np.searchsorted(po.astype(np.int64), d, side="right") - 1)
    print("call", it, "positions differing:", pos.size)
    for g in pos[:6]:
        r = np.searchsorted(off, g, side="right") - 1
        a, b = int(po[g]), int(po[g + 1])
        print("  read", r, "len", len(reads[r]), "pos", g - off[r])
        print("     ref:", list(zip(zn[a:b].tolist(), np.round(zl[a:b], 6).tolist())))
        print("     new:", list(zip(nd[a:b].tolist(), np.round(lp[a:b], 6).tolist())))
        print("     ref bits:", [hex(x) for x in zl[a:b].view(np.uint64)][-3:], "new bits:", [hex(x) for x in lp[a:b].view(np.uint64)][-3:], "ids hex", [hex(x) for x in nd[a:b]])
        print("     diff new-ref per entry:", (lp[a:b] - zl[a:b]).tolist())
        print("     exp(new)-exp(ref) / exp(ref[last]):", ((np.exp(lp[a:b]) - np.exp(zl[a:b])) / np.exp(zl[b-1])).tolist())
        a2, b2 = int(po[g - 1]), int(po[g])
        print("     new, position before:", list(zip(nd[a2:b2].tolist(), np.round(lp[a2:b2], 6).tolist())))
