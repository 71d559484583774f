"""GPU parity at BASELINE.json sizes: the HIP path against the oracle on samples of the REAL cfg3 / cfg2 workloads.

Per-read results do not depend on how reads are grouped (the pipeline tests assert bit-equality across groupings),
so the whole read set runs on the GPU and the oracle is run on exactly the sampled reads.  The sample holds the
reads that went through the code only a full-size run reaches (phmm_reads_last_call_info): every read the main
plan deferred (still dense after its kept warm-up columns), the reads whose frontier outgrew the one-lane-per-node
class, the reads with an over-full dense head (more than 400 nodes inside the ratio: histogram top-400), plus
random ones.  Dataset shape as in the reference's hmmv2/tests/dbg.rs:44-45, 85-114 (20x, 1000-bp reads,
p = 0.001, k = 40); bars: |d ln P| < 1e-6 per read (BASELINE.json), lists identical down to the ratio cut.
"""
import numpy as np
import pytest

import dbgphmm_amd as D
from dbgphmm_amd import _ffi
from helpers import compare_mappings, same_mappings, subset_csr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg3(gpu_lib):
    import bench
    arrays, reads, w = bench.build_workload("cfg3", 0)
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    mp, nf = gm.generate_mappings(rc, None, True)
    cols, flags = rc.last_call_info()
    return dict(arrays=arrays, reads=reads, gm=gm, rc=rc, mp=mp, nf=nf, cols=cols.copy(), flags=flags.copy())


def test_cfg3_sample_matches_oracle(cfg3, oracle):
    arrays, reads, rc, mp = cfg3["arrays"], cfg3["reads"], cfg3["rc"], cfg3["mp"]
    cols, flags = cfg3["cols"], cfg3["flags"]
    R = len(reads)
    po, nd, lp = mp.arrays()
    cnt = np.diff(po.astype(np.int64))
    off = rc.offsets.astype(np.int64)
    deferred = np.flatnonzero(flags & _ffi.PHMM_READ_DEFERRED)
    wide = np.flatnonzero(flags & _ffi.PHMM_READ_WIDE_FRONTIER)
    forced = np.flatnonzero(flags & _ffi.PHMM_READ_FORCED_SWITCH)
    # over-full dense head: a full 400-entry list at a position inside the read's dense warm-up
    full_pos = np.flatnonzero(cnt == 400)
    owner = np.searchsorted(off, full_pos, side="right") - 1
    overfull = np.unique(owner[(full_pos - off[owner]) < cols[owner]])
    # the scale-only paths really ran (cfg3: ~1 % of the reads deferred, a handful of wide / over-full ones)
    assert deferred.size >= 8 and wide.size >= 1 and overfull.size >= 1, (deferred.size, wide.size, overfull.size)
    assert np.all(cols[deferred] >= 18) and cols.max() <= arrays.param.n_warmup
    rng = np.random.default_rng(20261004)
    sample = np.unique(np.concatenate([rng.choice(R, 48, replace=False), deferred[:224], wide, forced, overfull]))
    sub = [reads[r] for r in sample]
    om = oracle.Model(arrays)
    omp, onf = om.generate_mappings(sub, None, True, n_threads=16)
    olp = om.full_prob_reads(sub, None, True, n_threads=16)
    glp = mp.read_logp()[1][sample]
    assert np.max(np.abs(glp - olp)) < 1e-6, (np.abs(glp - olp).max(), sample[np.argmax(np.abs(glp - olp))])
    gsub = subset_csr(off, (po, nd, lp), sample)
    compare_mappings(sub, gsub, omp)
    # node usage of the sampled reads (Mappings::to_node_freqs restricted to them)
    gnf = np.bincount(gsub[1], weights=np.exp(gsub[2]), minlength=arrays.n_nodes)
    capped = np.diff(gsub[0].astype(np.int64)).max() == 400
    if capped:  # a capped list keeps an arbitrary subset of the nodes tied at the cut
        assert abs(gnf.sum() - onf.sum()) < 1e-6 * len(sub)
    else:
        assert np.max(np.abs(gnf - onf)) < 1e-6
    # the score-only flow (to_full_prob_reads without mappings) walks the same plans: bit-equal to the mapping flow
    _, lp_s = cfg3["gm"].to_full_prob_reads(rc, None, True)
    assert np.array_equal(lp_s, mp.read_logp()[1])
    # and the hinted likelihood of the sampled reads on the GPU's own lists against the oracle's on the same lists
    _, lp_h = cfg3["gm"].to_full_prob_reads(rc, mp)
    olp_h = om.full_prob_reads(sub, gsub, True, n_threads=16)
    assert np.max(np.abs(lp_h[sample] - olp_h)) < 1e-9


def test_cfg3_two_live_models(cfg3):
    """`infer` keeps a mapping model (to_non_zero_phmm) and a scoring model (to_phmm) on the same graph
    (multi_dbg/posterior.rs:247-255, 609-630).  The DP workspaces belong to the device, so the second model runs
    in the memory the first one used; with the true copy numbers (all >= 1) the two models are equal and so are
    the bits of their results."""
    import bench
    arrays, reads, rc, mp = cfg3["arrays"], cfg3["reads"], cfg3["rc"], cfg3["mp"]
    sg_arrays = D.vectorised_to_phmm(bench.cfg_seq_graph("cfg3"), arrays.param, 0)
    gs = D.PHMMModel(sg_arrays)  # cfg3["gm"] stays alive
    held = _ffi.lib().phmm_workspace_bytes()
    assert held > 1 << 30
    tot_s, lp_s = gs.to_full_prob_reads(rc, None, True)
    assert np.array_equal(lp_s, mp.read_logp()[1])
    tot_h, lp_h = gs.to_full_prob_reads(rc, mp)
    _, lp_h0 = cfg3["gm"].to_full_prob_reads(rc, mp)
    assert np.array_equal(lp_h, lp_h0)
    mp2, nf2 = gs.generate_mappings(rc, None, True)
    assert same_mappings(mp.arrays(), mp2.arrays()) and np.array_equal(nf2, cfg3["nf"])
    # give the memory back and go again: the pool regrows on demand
    _ffi.check(_ffi.lib().phmm_release_workspace())
    assert _ffi.lib().phmm_workspace_bytes() == 0
    _, lp_s2 = gs.to_full_prob_reads(rc, None, True)
    assert np.array_equal(lp_s2, lp_s)


def test_cfg3_candidate_batch_packs_candidates(cfg3, monkeypatch):
    """The inner loop of `infer` (multi_dbg/posterior.rs:483-515) on the cfg3 mappings: a batch of candidates runs
    several candidates per wave (sparse.hip: hinted_packed_kernel); every candidate's per-read ln P has the bits of
    its one-candidate evaluation, whatever its place in the batch."""
    import bench
    arrays, rc, mp, gm = cfg3["arrays"], cfg3["rc"], cfg3["mp"], cfg3["gm"]
    sg = bench.cfg_seq_graph("cfg3")
    rng = np.random.default_rng(11)
    C = 11  # (not a multiple of the candidates per wave: the last wave of a read is partly idle)
    cn = np.repeat(sg.copy_num.astype(np.uint32)[None, :], C, axis=0)
    for c in range(1, C):
        ix = rng.integers(0, cn.shape[1], size=40)
        cn[c, ix] = np.maximum(cn[c, ix].astype(np.int64) + rng.choice([-1, 1], size=40), 0).astype(np.uint32)
    tot, lp = gm.to_full_prob_reads_copy_nums(rc, mp, cn, 0)
    # A k-mer set to 0 cuts the reads through it: every node of their lists dies there.  The reference still carries
    # the InsBegin chain (p_random p_II per base), which re-enters the graph behind the cut: a finite ln P of about -7
    # per base of the cut-off prefix.  The scaled kernels lose that chain (it underflows after ~105 bases); such
    # pairs are recomputed in the reference's log-space arithmetic (sparse.hip: hinted_exact_kernel).
    assert np.all(np.isfinite(lp)) and (lp[1:] < -500.0).any()
    dead = np.argwhere(lp < -500.0)
    for c, r in dead[:: max(1, len(dead) // 4)][:4]:
        with np.errstate(divide="ignore"):
            a2 = D.vectorised_to_phmm(D.SeqGraph(cn[c], sg.base, sg.edge_src, sg.edge_dst, None), arrays.param, 0)
        from oracle import oracle as O
        O.build()
        sub = subset_csr(rc.offsets.astype(np.int64), mp.arrays(), [int(r)])
        ol = O.Model(a2).full_prob_reads([cfg3["reads"][int(r)]], sub, True, n_threads=1)[0]
        assert abs(ol - lp[c, r]) < 1e-6, (int(c), int(r), ol, lp[c, r])
    for c in (0, 4, 10):
        t1, lp1 = gm.to_full_prob_reads_copy_nums(rc, mp, cn[c:c + 1], 0)  # one candidate: the one-candidate kernels
        assert np.array_equal(lp1[0], lp[c]) and t1[0] == tot[c]
    monkeypatch.setenv("PHMM_NO_PACKED", "1")
    tot2, lp2 = gm.to_full_prob_reads_copy_nums(rc, mp, cn, 0)
    assert np.array_equal(lp2, lp)
    # candidate 0 carries the graph's own copy numbers: the mapping model's hinted likelihood
    _, lp_h = gm.to_full_prob_reads(rc, mp)
    assert np.max(np.abs(lp[0] - lp_h)) < 1e-9


def test_cfg3_small_workspace_limit(cfg3):
    """The same read set under a 24 GB table budget (many chunks; deferred reads' plan inside its fixed share):
    same bits as the one-chunk run."""
    L = _ffi.lib()
    rc, mp = cfg3["rc"], cfg3["mp"]
    _ffi.check(L.phmm_release_workspace())
    _ffi.check(L.phmm_set_workspace_limit(24 << 30))
    try:
        mp2, nf2 = cfg3["gm"].generate_mappings(rc, None, True)
        assert L.phmm_workspace_bytes() < (60 << 30)
    finally:
        _ffi.check(L.phmm_set_workspace_limit(0))
    assert same_mappings(mp.arrays(), mp2.arrays()) and np.array_equal(nf2, cfg3["nf"])


def test_cfg2_sample_matches_oracle(gpu_lib, oracle):
    """BASELINE.json configs[1] (10 kb haploid, k = 40, dense forward + backward + node posteriors): the whole
    read set on the GPU; a sample of reads cut to ~100 bases again next to the oracle (a cut read is a different
    read: both sides run the same bytes)."""
    import bench
    arrays, reads, w = bench.build_workload("cfg2", 0)
    gm = D.PHMMModel(arrays)
    rng = np.random.default_rng(7)
    pick = rng.choice(len(reads), 16, replace=False)
    sub = [reads[r][: 90 + int(j) * 2] for j, r in enumerate(pick)]
    lf, lb, nf = gm.run_dense(D.ReadCollection(sub))
    om = oracle.Model(arrays)
    olf, olb, onf = om.run_dense_reads(sub, n_threads=16)
    assert np.max(np.abs(lf - olf)) < 1e-9 and np.max(np.abs(lb - olb)) < 1e-9
    assert np.max(np.abs(nf - onf)) < 1e-8
    # full size through properties: forward and backward totals of a dense run differ only by the Del-chain
    # truncation (a read with a deletion: 1e-3), node usage sums to about one node per base
    rc = D.ReadCollection(reads)
    lf_all, lb_all, nf_all = gm.run_dense(rc)
    assert np.all(np.isfinite(lf_all)) and np.max(np.abs(lf_all - lb_all)) < 0.01  # (hmmv2/tests/dbg.rs:45)
    assert abs(nf_all.sum() - rc.total_bases()) < 0.01 * rc.total_bases()
    # the sample's full-length siblings inside the big batch and alone (another read-group shape, hence another
    # order of the additions): equal to rounding
    lf1, lb1, _ = gm.run_dense(D.ReadCollection([reads[r] for r in pick]))
    assert np.max(np.abs(lf1 - lf_all[pick])) < 1e-9 and np.max(np.abs(lb1 - lb_all[pick])) < 1e-9
