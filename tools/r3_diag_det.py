"""dev: determinism + timing of generate_mappings on a workload for the library in PHMM_AMD_LIB.
usage: python tools/r3_diag_det.py workload read_len [ref.npz to compare with / to write]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import dbgphmm_amd as D

wl = sys.argv[1]
rl = int(sys.argv[2])
ref = sys.argv[3] if len(sys.argv) > 3 else None
arrays, reads, w = bench.build_workload(wl, 0, 1, "strong", rl)
gm = D.PHMMModel(arrays)
rc = D.ReadCollection(reads)
res = []
for it in range(3):
    t = time.perf_counter()
    mp, nf = gm.generate_mappings(rc, None, True)
    dt = (time.perf_counter() - t) * 1e3
    po, nd, lp = mp.arrays()
    res.append((po.copy(), nd.copy(), lp.copy(), nf.copy(), mp.read_logp()[1].copy()))
    print(f"[{os.environ.get('PHMM_AMD_LIB','default')}] {wl} L={rl} call {it}: {dt:.1f} ms", flush=True)
off = rc.offsets.astype(np.int64)
cols, flags = rc.last_call_info()


def cmp(a, b, tag):
    if not np.array_equal(a[0], b[0]):
        bad = np.flatnonzero(np.diff(a[0].astype(np.int64)) != np.diff(b[0].astype(np.int64)))
        owner = np.searchsorted(off, bad, side="right") - 1
        print(tag, "list LENGTHS differ at", bad.size, "positions; reads", np.unique(owner)[:10], "pos in read", (bad - off[owner])[:10],
              "flags", flags[np.unique(owner)[:10]], "cols", cols[np.unique(owner)[:10]])
        return
    d = np.flatnonzero((a[1] != b[1]) | (a[2] != b[2]))
    if d.size:
        pos = np.searchsorted(a[0].astype(np.int64), d, side="right") - 1
        owner = np.searchsorted(off, pos, side="right") - 1
        print(tag, "entries differ:", d.size, "reads", np.unique(owner)[:10], "pos in read", (pos - off[owner])[:10], "max dlp", np.abs(a[2][d] - b[2][d]).max(),
              "flags", flags[np.unique(owner)[:10]])
    else:
        print(tag, "identical lists;", "nf equal" if np.array_equal(a[3], b[3]) else f"nf differs {np.abs(a[3]-b[3]).max()}",
              "logp equal" if np.array_equal(a[4], b[4]) else "logp differs")


cmp(res[0], res[1], "call0 vs call1:")
cmp(res[1], res[2], "call1 vs call2:")
if ref and os.path.exists(ref):
    z = np.load(ref)
    cmp((z["po"], z["nd"], z["lp"], z["nf"], z["rl"]), res[1], "reference file vs call1:")
elif ref:
    np.savez(ref, po=res[1][0], nd=res[1][1], lp=res[1][2], nf=res[1][3], rl=res[1][4])
