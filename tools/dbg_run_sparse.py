import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import dbgphmm_amd as D
from helpers import small_dbg_model
from oracle import oracle as O
O.build()
gl, k, p, seed, n_active, rl = 600, 12, 0.01, 3, 40, 150
arrays, sg = small_dbg_model(gl, k, p, seed=seed, min_copy_num=1)
arrays.param = arrays.param.with_(n_active_nodes=n_active)
reads = D.sample_reads(arrays, 10 ** 9, rl, seed=seed + 1, max_reads=24)
reads = [r[: max(2, len(r) - (j * 11) % (rl - 5))] for j, r in enumerate(reads)]
gm, om = D.PHMMModel(arrays), O.Model(arrays)
nw = arrays.param.n_warmup
print("n_warmup", nw, "N", arrays.n_nodes)
for j, r in enumerate(reads):
    lf, lb, nf = gm.run_sparse(D.ReadCollection([r]))
    o = om.run_sparse(r)
    onf = o.to_node_freqs()
    d = np.abs(nf - onf)
    if d.max() > 1e-9:
        v = int(d.argmax())
        print(f"read {j} len {len(r)} maxdiff {d.max():.3e} node {v} gpu {nf[v]:.6e} orc {onf[v]:.6e} ndiff {(d>1e-9).sum()}")
        # per merged index contribution of the oracle at node v
        for jj in range(len(r) + 1):
            m, i, dd, s = o.to_emit_probs(jj)
            c = np.exp(m[v]) + np.exp(i[v]) + np.exp(dd[v])
            if c > 1e-6:
                print(f"    j={jj} oracle contrib {c:.6e}  (m {np.exp(m[v]):.3e} i {np.exp(i[v]):.3e} d {np.exp(dd[v]):.3e})")
    else:
        print(f"read {j} len {len(r)} ok")
print("---- read 12 detail")
r = reads[12]
lf, lb, nf = gm.run_sparse(D.ReadCollection([r]))
o = om.run_sparse(r)
onf = o.to_node_freqs()
unc = np.zeros(arrays.n_nodes)
for jj in range(len(r) + 1):
    m, i, dd, s = o.to_emit_probs(jj)
    unc += np.exp(m) + np.exp(i) + np.exp(dd)
d = np.abs(nf - onf)
for v in np.nonzero(d > 1e-9)[0]:
    print(f"node {v} gpu {nf[v]:.6e} orc {onf[v]:.6e} uncapped {unc[v]:.6e}")
print("max |gpu - uncapped|", np.abs(nf - unc).max())
