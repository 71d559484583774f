"""dev script: time the sparse flow on a cfg3-like workload"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dbgphmm_amd as D
from dbgphmm_amd import _ffi
G = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t = time.time()
hap = D.random_genome(G, seed=3)
hapb = D.diverge(hap, 0.01, seed=4)
sg = D.dbg_from_haplotypes([hap, hapb], 40)
par = D.PHMMParams.uniform(0.001).with_(n_warmup=40)
a_map = D.vectorised_to_phmm(sg, par, 1)
a_sc = D.vectorised_to_phmm(sg, par, 0)
print("graph", a_map.n_nodes, a_map.n_edges, time.time() - t, flush=True)
t = time.time()
reads = D.sample_reads(a_sc, cov * (len(hap) + len(hapb)), 1000, seed=0)
nb = sum(map(len, reads))
print("reads", len(reads), nb, time.time() - t, flush=True)
L = _ffi.lib(); L.phmm_enable_timing(1)
t = time.time(); gm = D.PHMMModel(a_map); print("model", time.time() - t, flush=True)
rc = D.ReadCollection(reads)
for it in range(2):
    t = time.time(); mp, nf = gm.generate_mappings(rc, None, True); dt = time.time() - t
    po, nd, lp = mp.arrays()
    print(f"generate_mappings {dt:.3f}s  {nb/dt:.3e} bases/s entries={len(nd)} mean_list={len(nd)/nb:.2f} max_list={np.diff(po).max()} nfsum={nf.sum():.1f}", flush=True)
gs = D.PHMMModel(a_sc)
for it in range(2):
    t = time.time(); tot, lps = gs.to_full_prob_reads(rc, mp); dt = time.time() - t
    print(f"hinted full_prob {dt:.4f}s {nb/dt:.3e} bases/s tot={tot:.3f}", flush=True)
t = time.time(); tot2, lps2 = gs.to_full_prob_reads(rc, None); dt = time.time() - t
print(f"sparse full_prob {dt:.3f}s {nb/dt:.3e} bases/s tot={tot2:.3f} maxdiff={np.abs(lps-lps2).max():.2e}", flush=True)
