"""dev check: long reads (15-25 kb) through the adaptive flow, the hinted flow and a candidate batch, against the oracle
on a few of them.  usage: python tools/long_reads_check.py [read_len] [n_reads]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dbgphmm_amd as D  # noqa: E402
from helpers import compare_mappings_tie_aware, subset_csr  # noqa: E402
from oracle import oracle as O  # noqa: E402

rl = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 40
O.build()
hap = D.random_genome(60000, 5)
sg = D.dbg_from_haplotypes([hap, D.diverge(hap, 0.001, 6)], 40)
param = D.PHMMParams.uniform(0.001).with_(n_warmup=40)
arrays = D.vectorised_to_phmm(sg, param, 1)
reads = D.sample_reads(arrays, 10 ** 9, rl, seed=3, max_reads=nr)
print("N", arrays.n_nodes, "reads", len(reads), "lengths", min(map(len, reads)), max(map(len, reads)), flush=True)
gm, om = D.PHMMModel(arrays), O.Model(arrays)
rc = D.ReadCollection(reads)
t = time.time()
mp, nf = gm.generate_mappings(rc, None, True)
print("generate_mappings %.2f s" % (time.time() - t), "flags", np.bincount(rc.last_call_info()[1], minlength=8).tolist(), flush=True)
pick = [0, len(reads) // 2, len(reads) - 1]
sub = [reads[r] for r in pick]
olp = om.full_prob_reads(sub, None, True, n_threads=16)
glp = mp.read_logp()[1][pick]
print("adaptive lnP gpu", glp, "oracle", olp, "max diff", np.abs(glp - olp).max(), flush=True)
assert np.abs(glp - olp).max() < 1e-6
omp, _ = om.generate_mappings(sub, None, True, n_threads=16)
gsub = subset_csr(rc.offsets.astype(np.int64), mp.arrays(), pick)
print("lists: tie / overflow reads", compare_mappings_tie_aware(O, om, sub, gsub, omp), flush=True)
_, lph = gm.to_full_prob_reads(rc, mp)
olph = om.full_prob_reads(sub, gsub, True, n_threads=16)
print("hinted max diff", np.abs(lph[pick] - olph).max(), flush=True)
assert np.abs(lph[pick] - olph).max() < 1e-9
# a candidate that cuts read 0 far from its start
off = rc.offsets.astype(np.int64)
po, nd, lp = mp.arrays()
victim = int(nd[int(po[off[pick[0]] + len(sub[0]) // 2])])
cn = np.stack([sg.copy_num, sg.copy_num]).astype(np.uint32)
cn[1, victim] = 0
tot, lpc = gm.to_full_prob_reads_copy_nums(rc, mp, cn, 0)
with np.errstate(divide="ignore"):
    a1 = D.vectorised_to_phmm(D.SeqGraph(cn[1], sg.base, sg.edge_src, sg.edge_dst, None), param, 0)
ol1 = O.Model(a1).full_prob_reads(sub, gsub, True, n_threads=16)
print("cut candidate: gpu", lpc[1][pick], "oracle", ol1, flush=True)
assert np.abs(lpc[1][pick] - ol1).max() < 1e-6
print("long reads ok")
