"""dev script: time backward_sparse (to_full_prob_sparse_backward) next to the fixed top-k forward"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dbgphmm_amd as D
G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
hap = D.random_genome(G, seed=3)
hapb = D.diverge(hap, 0.01, seed=4)
sg = D.dbg_from_haplotypes([hap, hapb], 40)
par = D.PHMMParams.uniform(0.001).with_(n_warmup=40)
a = D.vectorised_to_phmm(sg, par, 0)
reads = D.sample_reads(a, 20 * (len(hap) + len(hapb)), 1000, seed=0)
nb = sum(map(len, reads))
print("graph", a.n_nodes, "reads", len(reads), nb, flush=True)
gm = D.PHMMModel(a)
rc = D.ReadCollection(reads)
for it in range(2):
    t = time.time(); ftot, flp = gm.to_full_prob_reads(rc, None, False); dt = time.time() - t
    print(f"forward fixed top-k  {dt:.3f}s {nb/dt:.3e} bases/s tot={ftot:.3f}", flush=True)
for it in range(2):
    t = time.time(); btot, blp = gm.to_full_prob_sparse_backward(rc); dt = time.time() - t
    print(f"backward_sparse      {dt:.3f}s {nb/dt:.3e} bases/s tot={btot:.3f} max|f-b|={np.abs(flp-blp).max():.3e}", flush=True)
for it in range(2):
    t = time.time(); lf, lb, nf = gm.run_sparse(rc); dt = time.time() - t
    print(f"run_sparse           {dt:.3f}s {nb/dt:.3e} bases/s nfsum={nf.sum():.1f} max|lf-flp|={np.abs(lf-flp).max():.2e} max|lb-blp|={np.abs(lb-blp).max():.2e}", flush=True)
