"""dev script: where do two generate_mappings results on cfg3 differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import dbgphmm_amd as D
from dbgphmm_amd import _ffi

arrays, reads, w = bench.build_workload("cfg3", 0)
gm = D.PHMMModel(arrays)
rc = D.ReadCollection(reads)
off = rc.offsets.astype(np.int64)


def diff(a, b, tag, cols):
    po1, nd1, lp1 = a
    po2, nd2, lp2 = b
    print(tag, "pos_off equal", np.array_equal(po1, po2), "nodes", np.array_equal(nd1, nd2), "logp", np.array_equal(lp1, lp2), flush=True)
    if np.array_equal(po1, po2):
        bad = np.flatnonzero((nd1 != nd2) | ~((lp1 == lp2) | (np.isnan(lp1) & np.isnan(lp2))))
        print("  differing entries", bad.size)
        if bad.size:
            pos = np.unique(np.searchsorted(po1, bad, side="right") - 1)
            for p in pos[:12]:
                r = int(np.searchsorted(off, p, side="right") - 1)
                s0, s1 = int(po1[p]), int(po1[p + 1])
                print("   read", r, "pos", int(p - off[r]), "cols", int(cols[r]), "n", s1 - s0)
                print("     A", nd1[s0:s1][:8], lp1[s0:s1][:8])
                print("     B", nd2[s0:s1][:8], lp2[s0:s1][:8])
    else:
        c1, c2 = np.diff(po1.astype(np.int64)), np.diff(po2.astype(np.int64))
        pos = np.flatnonzero(c1 != c2)
        print("  positions with different counts", pos.size)
        for p in pos[:12]:
            r = int(np.searchsorted(off, p, side="right") - 1)
            print("   read", r, "pos", int(p - off[r]), "cols", int(cols[r]), "nA", int(c1[p]), "nB", int(c2[p]))


mp1, nf1 = gm.generate_mappings(rc, None, True)
cols, flags = rc.last_call_info()
print("deferred", int((flags & 1).sum()), "wide", int((flags & 2 > 0).sum()), "forced", int((flags & 4 > 0).sum()), "ws", _ffi.lib().phmm_workspace_bytes() >> 20, "MiB", flush=True)
a1 = tuple(x.copy() for x in mp1.arrays())
mp2, nf2 = gm.generate_mappings(rc, None, True)
diff(a1, mp2.arrays(), "same model, call 2:", cols)
mp3, nf3 = gm.generate_mappings(rc, None, True)
diff(a1, mp3.arrays(), "same model, call 3:", cols)
gs = D.PHMMModel(D.vectorised_to_phmm(bench.cfg_seq_graph("cfg3"), arrays.param, 0))
print("arrays equal", np.array_equal(gs.arrays.init_logp, arrays.init_logp), np.array_equal(gs.arrays.trans_logp, arrays.trans_logp))
mp4, nf4 = gs.generate_mappings(rc, None, True)
diff(a1, mp4.arrays(), "second model:", cols)
print("nf equal", np.array_equal(nf1, nf2), np.array_equal(nf1, nf4), "max", np.abs(nf1 - nf4).max())
