"""BASELINE.json configs[4]'s shape on ONE GPU: the 1 Mb diploid k = 40 graph (N = 1.3e6 nodes) with a tenth of its 20x
read set, through the three steps of the `infer` loop that are on this path (multi_dbg/posterior.rs:698-826):
  A  generate_mappings on the new graph (:735) -- chunked: one read group of 64 takes ~40 GB of dense warm-up tables,
  B  a candidate batch on its mappings (:483-515),
  C  the mappings carried to the k+1 graph (hint_kp1_from_hint_k, multi_dbg.rs:1325-1335).
The oracle runs on a sample that holds every read that took a rarer route (phmm_reads_last_call_info) plus random
ones; `bench.py --workload cfg5` times the full 20x set (profiles/).  The whole infer loop (50 iterations x k = 40 ->
20 000, copy-number proposals, min-flow) is the reference's control plane and stays in the reference.
"""
import numpy as np
import pytest

import dbgphmm_amd as D
from dbgphmm_amd import _ffi
from helpers import compare_mappings, subset_csr

pytestmark = pytest.mark.gpu


def test_cfg5_shape_one_gpu(gpu_lib, oracle):
    import bench
    w = bench.WORKLOADS["cfg5"]
    haps = bench.cfg_haplotypes("cfg5")
    sg1, map_off, map_nodes, sg = D.kp1_node_map(haps, w["k"])
    param = D.PHMMParams.uniform(w["p"]).with_(n_warmup=w["k"])
    arrays = D.vectorised_to_phmm(sg, param, 1)
    assert arrays.n_nodes > 1_200_000
    total = w["coverage"] * w["genome"] * w["haplotypes"] // 10
    reads = D.sample_reads(arrays, total, w["read_len"], seed=1000)
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    # ---- A: generate_mappings (many chunks of read groups)
    mp, nf = gm.generate_mappings(rc, None, True)
    cols, flags = rc.last_call_info()
    po, nd, lp = mp.arrays()
    cnt = np.diff(po.astype(np.int64))
    assert cnt.min() >= 1 and cnt.max() <= 400 and np.all(np.isfinite(lp))
    assert abs(nf.sum() - rc.total_bases()) < 0.01 * rc.total_bases()
    tot_s, lp_s = gm.to_full_prob_reads(rc, None, True)
    assert np.array_equal(lp_s, mp.read_logp()[1])  # the score-only flow: same plans, same bits
    rng = np.random.default_rng(55)
    special = np.flatnonzero(flags != 0)
    sample = np.unique(np.concatenate([rng.choice(len(reads), 16, replace=False), special[:16]])).astype(int)
    sample = np.array([r for r in sample if not (flags[r] & _ffi.PHMM_READ_FORCED_SWITCH)], dtype=int)
    sub = [reads[r] for r in sample]
    om = oracle.Model(arrays)
    olp = om.full_prob_reads(sub, None, True, n_threads=16)
    assert np.max(np.abs(mp.read_logp()[1][sample] - olp)) < 1e-6
    omp, onf = om.generate_mappings(sub, None, True, n_threads=16)
    off = rc.offsets.astype(np.int64)
    gsub = subset_csr(off, (po, nd, lp), sample)
    compare_mappings(sub, gsub, omp)
    # ---- B: a candidate batch on these mappings; candidate 0 = the graph's own copy numbers
    C = 6
    cn = np.repeat(sg.copy_num.astype(np.uint32)[None, :], C, axis=0)
    for c in range(1, C):
        ix = rng.integers(0, cn.shape[1], size=160)
        cn[c, ix] = np.maximum(cn[c, ix].astype(np.int64) + rng.choice([-1, 1], size=160), 0).astype(np.uint32)
    tot, lpc = gm.to_full_prob_reads_copy_nums(rc, mp, cn, 0)
    assert np.all(np.isfinite(lpc))
    _, lp1 = gm.to_full_prob_reads_copy_nums(rc, mp, cn[4:5], 0)
    assert np.array_equal(lp1[0], lpc[4])
    _, lp_h = gm.to_full_prob_reads(rc, mp)
    assert np.max(np.abs(lpc[0] - lp_h)) < 1e-9
    olp_h = om.full_prob_reads(sub, gsub, True, n_threads=16)
    assert np.max(np.abs(lp_h[sample] - olp_h)) < 1e-9
    with np.errstate(divide="ignore"):
        a4 = D.vectorised_to_phmm(D.SeqGraph(cn[4].astype(np.int64), sg.base, sg.edge_src, sg.edge_dst, None), param, 0)
    ol4 = oracle.Model(a4).full_prob_reads(sub, gsub, True, n_threads=16)
    assert np.max(np.abs(ol4 - lpc[4][sample])) < 1e-6
    # ---- C: the lists carried to the k+1 graph
    arrays1 = D.vectorised_to_phmm(sg1, param.with_(n_warmup=w["k"] + 1), 1)
    gm1 = D.PHMMModel(arrays1)
    mp1 = mp.map_nodes(gm1, map_off, map_nodes)
    p1, n1, l1 = mp1.arrays()
    assert np.array_equal(np.diff(p1.astype(np.int64)) > 0, cnt > 0) and n1.max() < arrays1.n_nodes
    few = sample[:4]
    exp = oracle.map_nodes(subset_csr(off, (po, nd, lp), few), map_off, map_nodes)
    got = subset_csr(off, (p1, n1, l1), few)
    assert np.array_equal(got[0], exp[0])
    for i in range(len(exp[0]) - 1):
        s0, s1 = int(exp[0][i]), int(exp[0][i + 1])
        assert np.max(np.abs(got[2][s0:s1] - exp[2][s0:s1]), initial=0.0) < 1e-9
        assert sorted(got[1][s0:s1].tolist()) == sorted(exp[1][s0:s1].tolist())
    # the carried lists do their job: the hinted likelihood on the k+1 graph is finite and near the k one (the two
    # graphs differ in their first k bases' states: ~0.15 nats per read here)
    tot1, lpk1 = gm1.to_full_prob_reads(rc, mp1)
    assert np.all(np.isfinite(lpk1)) and abs(tot1 - lp_h.sum()) < 0.5 * len(reads)
    print(f"\ncfg5 (1/10 reads): N={arrays.n_nodes} reads={len(reads)} sample={len(sample)} deferred="
          f"{int((flags & 1).sum())} wide={int(((flags & 2) != 0).sum())} mean dense cols={cols.mean():.1f} "
          f"entries k={nd.shape[0]} k+1={n1.shape[0]}")
