"""dev script: hinted forward time vs number of reads (latency- or throughput-bound?)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dbgphmm_amd as D
hap = D.random_genome(100000, seed=3)
hapb = D.diverge(hap, 0.01, seed=4)
sg = D.dbg_from_haplotypes([hap, hapb], 40)
par = D.PHMMParams.uniform(0.001).with_(n_warmup=40)
a_map = D.vectorised_to_phmm(sg, par, 1)
a_sc = D.vectorised_to_phmm(sg, par, 0)
reads = D.sample_reads(a_sc, 20 * (len(hap) + len(hapb)), 1000, seed=0)
gm = D.PHMMModel(a_map)
gs = D.PHMMModel(a_sc)
for frac in (1.0, 0.5, 0.25, 0.06):
    sub = reads[: max(1, int(len(reads) * frac))]
    rc = D.ReadCollection(sub)
    mp, nf = gm.generate_mappings(rc, None, True)
    nb = sum(map(len, sub))
    ts = []
    for it in range(4):
        t = time.time(); tot, lps = gs.to_full_prob_reads(rc, mp); ts.append(time.time() - t)
    print(f"reads {len(sub)} bases {nb} hinted best {min(ts)*1e3:.2f} ms  {nb/min(ts):.3e} bases/s", flush=True)
