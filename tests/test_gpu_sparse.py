"""GPU parity: hinted (mapping-restricted) forward = the inner loop of `infer`
(forward_with_mapping_score_only, forward.rs:79-89; to_full_prob_reads, freq.rs:175-192)."""
import numpy as np
import pytest

import dbgphmm_amd as D
from helpers import compare_mappings as _compare_mappings, finite_close, same_mappings, small_dbg_model

pytestmark = pytest.mark.gpu
TOL_LOGP = 1e-9  # BASELINE.json bar: |delta ln P(R|X)| < 1e-6 per read


def _setup(oracle, genome_len=600, k=12, p=0.01, n_reads=24, seed=3, read_len=120):
    arrays, sg = small_dbg_model(genome_len, k, p, seed=seed)
    reads = D.sample_reads(arrays, 10 ** 9, read_len, seed=seed + 1, max_reads=n_reads)
    reads = [r[: max(5, len(r) - (j * 5) % 23)] for j, r in enumerate(reads)]
    om = oracle.Model(arrays)
    (po, nd, lp), nf = om.generate_mappings(reads, None, True, n_threads=8)
    return arrays, sg, reads, om, (po, nd, lp)


def test_hinted_forward_matches_oracle(gpu_lib, oracle):
    arrays, sg, reads, om, mp = _setup(oracle)
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    gmp = D.Mappings.from_arrays(rc, *mp)
    tot, lp = gm.to_full_prob_reads(rc, gmp)
    olp = om.full_prob_reads(reads, mp, True, n_threads=8)
    assert np.max(np.abs(lp - olp)) < TOL_LOGP
    assert abs(tot - olp.sum()) < TOL_LOGP * len(reads)
    # the hinted score is close to the dense one (reference: hmmv2/tests/dbg.rs:44-45, 1e-4 on the total)
    lf, _, _ = gm.run_dense(rc, False, False)
    assert abs(lf.sum() - tot) < 1e-4 * len(reads)


def test_hinted_forward_top_k_lists(gpu_lib, oracle):
    """short fixed-size lists (to_mapping(n_active)) exercise nodes outside the list."""
    arrays, sg, reads, om, _ = _setup(oracle, n_reads=10, seed=9)
    gm = D.PHMMModel(arrays)
    pos_off, nodes = [0], []
    for r in reads:
        m = om.run(r).to_mapping(3)
        for i in range(len(r)):
            nodes.extend(m.nodes(i))
            pos_off.append(len(nodes))
    mp = (np.array(pos_off, dtype=np.uint64), np.array(nodes, dtype=np.uint32), np.zeros(len(nodes)))
    rc = D.ReadCollection(reads)
    tot, lp = gm.to_full_prob_reads(rc, D.Mappings.from_arrays(rc, *mp))
    olp = om.full_prob_reads(reads, mp, True, n_threads=8)
    assert np.max(np.abs(lp - olp)) < TOL_LOGP


def test_candidate_batch(gpu_lib, oracle):
    """C candidate copy-number vectors on one topology (posterior.rs:483-515)."""
    arrays, sg, reads, om, mp = _setup(oracle, n_reads=12, seed=5)
    rng = np.random.default_rng(0)
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    gmp = D.Mappings.from_arrays(rc, *mp)
    inits, transs, expect = [], [], []
    for c in range(5):
        sg2 = D.SeqGraph(sg.copy_num.copy(), sg.base, sg.edge_src, sg.edge_dst, None)
        flip = rng.integers(0, sg2.copy_num.shape[0], size=8)
        sg2.copy_num[flip] = rng.integers(0, 4, size=8)
        a2 = D.vectorised_to_phmm(sg2, arrays.param, 0)
        inits.append(a2.init_logp)
        transs.append(a2.trans_logp)
        expect.append(oracle.Model(a2).full_prob_reads(reads, mp, True, n_threads=8))
    tot, lp = gm.to_full_prob_reads_candidates(rc, gmp, np.stack(inits), np.stack(transs))
    expect = np.stack(expect)
    both_inf = np.isneginf(lp) & np.isneginf(expect)
    with np.errstate(invalid="ignore"):
        assert np.all(both_inf | (np.abs(lp - expect) < TOL_LOGP))
    # set_probs + single evaluation agrees with the batched one
    gm.set_probs(inits[2], transs[2])
    t2, lp2 = gm.to_full_prob_reads(rc, gmp)
    with np.errstate(invalid="ignore"):
        assert np.all((np.isneginf(lp2) & np.isneginf(lp[2])) | (np.abs(lp2 - lp[2]) < 1e-12))


@pytest.mark.parametrize("cpl", ["1", "2"])
def test_candidate_batch_packed_kernels_all_widths(gpu_lib, oracle, cpl, monkeypatch):
    """Candidate batches on reads whose longest lists fall in each packed class (<= 8, <= 16, <= 32 nodes) and beyond
    (one-candidate kernels), one and two candidates per lane: every (candidate, read) equals the oracle's score and
    has the bits of the one-candidate evaluation."""
    monkeypatch.setenv("PHMM_PACKED_CPL", cpl)
    arrays, sg, reads, om, _ = _setup(oracle, genome_len=500, n_reads=8, seed=13, read_len=60)
    gm = D.PHMMModel(arrays)
    rng = np.random.default_rng(2)
    # lists of the oracle's dense run cut to a per-read width: 3, 8, 9, 16, 17, 32, 33, 70 nodes
    widths = [3, 8, 9, 16, 17, 32, 33, 70]
    pos_off, nodes = [0], []
    for r, wd in zip(reads, widths):
        mpp = om.run(r).to_mapping(wd)
        for i in range(len(r)):
            nodes.extend(mpp.nodes(i))
            pos_off.append(len(nodes))
    mp = (np.array(pos_off, dtype=np.uint64), np.array(nodes, dtype=np.uint32), np.zeros(len(nodes)))
    rc = D.ReadCollection(reads)
    gmp = D.Mappings.from_arrays(rc, *mp)
    C = 19
    cns = []
    for c in range(C):
        cn = sg.copy_num.copy()
        flip = rng.integers(0, cn.shape[0], size=12)
        cn[flip] = rng.integers(0, 4, size=12)
        cns.append(cn)
    tot, lp = gm.to_full_prob_reads_copy_nums(rc, gmp, np.stack(cns), 0)
    for c in (0, 7, 18):
        t1, lp1 = gm.to_full_prob_reads_copy_nums(rc, gmp, np.stack(cns[c:c + 1]), 0)
        assert np.array_equal(lp1[0], lp[c])
        with np.errstate(divide="ignore"):
            a2 = D.vectorised_to_phmm(D.SeqGraph(cns[c], sg.base, sg.edge_src, sg.edge_dst, None), arrays.param, 0)
        ol = oracle.Model(a2).full_prob_reads(reads, mp, True, n_threads=8)
        with np.errstate(invalid="ignore"):
            assert np.all((np.isneginf(ol) & np.isneginf(lp[c])) | (np.abs(ol - lp[c]) < TOL_LOGP))


def test_reads_cut_by_a_zero_copy_kmer_keep_the_begin_chain(gpu_lib, oracle, monkeypatch):
    """A candidate that sets a k-mer on a read's path to copy number 0 kills every node of the read's lists at that
    position.  The reference's ln P stays finite: the InsBegin chain (p_random p_II per base) re-enters the graph
    behind the cut (forward.rs:337-359 from_begin, 541-545).  In the scaled linear domain that chain underflows
    ~105 bases into the read; such pairs are recomputed in log space (sparse.hip: hinted_exact_kernel) -- for all
    three kernel classes, for candidates given as copy numbers, as probability vectors, and for a model built with
    the zero already in it."""
    arrays, sg, reads, om, mp = _setup(oracle, genome_len=900, k=12, p=0.003, n_reads=40, seed=21, read_len=420)
    reads = [r for r in reads if len(r) > 380][:24]
    (po, nd, lpm), _ = om.generate_mappings(reads, None, True, n_threads=8)
    mp = (po, nd, lpm)
    rc = D.ReadCollection(reads)
    gm = D.PHMMModel(arrays)
    gmp = D.Mappings.from_arrays(rc, *mp)
    # the best node of base 300 of read 0 goes to zero: reads through it are cut ~300 bases in (past the underflow)
    off = rc.offsets.astype(np.int64)
    victim = int(nd[int(po[off[0] + 300])])
    cn0 = sg.copy_num.copy()
    cn1 = cn0.copy()
    cn1[victim] = 0
    with np.errstate(divide="ignore"):
        a1 = D.vectorised_to_phmm(D.SeqGraph(cn1, sg.base, sg.edge_src, sg.edge_dst, None), arrays.param, 0)
    ol = oracle.Model(a1).full_prob_reads(reads, mp, True, n_threads=8)
    healthy = om.full_prob_reads(reads, mp, True, n_threads=8)
    cut = np.flatnonzero(ol < healthy - 200.0)
    assert cut.size >= 2 and np.all(np.isfinite(ol))
    for env in ({}, {"PHMM_NO_PACKED": "1"}, {"PHMM_NO_PACKED": "1", "PHMM_NO_LEAN": "1"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        tot, lp = gm.to_full_prob_reads_copy_nums(rc, gmp, np.stack([cn0, cn1, cn1]), 0)
        assert np.max(np.abs(lp[1] - ol)) < 1e-9 and np.array_equal(lp[1], lp[2]), env
        assert np.max(np.abs(lp[0] - healthy)) < 1e-9
        assert abs(tot[1] - ol.sum()) < 1e-6
        t2, lp2 = gm.to_full_prob_reads_candidates(rc, gmp, a1.init_logp[None, :], a1.trans_logp[None, :])
        assert np.max(np.abs(lp2[0] - ol)) < 1e-9
        _, lp3 = D.PHMMModel(a1).to_full_prob_reads(rc, gmp)
        assert np.max(np.abs(lp3 - ol)) < 1e-9
        for k_ in env:
            monkeypatch.delenv(k_)
    # without the fallback the cut reads come back -inf
    monkeypatch.setenv("PHMM_NO_EXACT_HINTED", "1")
    _, lp4 = gm.to_full_prob_reads_copy_nums(rc, gmp, np.stack([cn1]), 0)
    assert np.all(np.isneginf(lp4[0][cut]))
    monkeypatch.delenv("PHMM_NO_EXACT_HINTED")
    # generate_mappings WITH lists needs the forward columns of every read: a model that cuts a read is refused with a
    # clear error (the reference maps on to_non_zero_phmm, multi_dbg/posterior.rs:609-618), the healthy model is not
    with pytest.raises(D.PhmmError, match="to_non_zero_phmm"):
        D.PHMMModel(a1).generate_mappings(rc, gmp, True)
    mph, _ = gm.generate_mappings(rc, gmp, True)
    assert np.all(np.isfinite(mph.arrays()[2]))


def test_candidate_copy_numbers_on_device(gpu_lib, oracle):
    """candidates as copy-number vectors: init / trans built on the device (seq_graph.rs:160-209) give the
    same likelihoods as the host-built probability vectors, for to_phmm (min 0) and to_non_zero_phmm (min 1)."""
    arrays, sg, reads, om, mp = _setup(oracle, n_reads=12, seed=5)
    rng = np.random.default_rng(1)
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    gmp = D.Mappings.from_arrays(rc, *mp)
    for min_cn in (0, 1):
        cns, inits, transs = [], [], []
        for c in range(4):
            cn = sg.copy_num.copy()
            flip = rng.integers(0, cn.shape[0], size=10)
            cn[flip] = rng.integers(0, 4, size=10)
            sg2 = D.SeqGraph(cn, sg.base, sg.edge_src, sg.edge_dst, None)
            with np.errstate(divide="ignore"):
                a2 = D.vectorised_to_phmm(sg2, arrays.param, min_cn)
            cns.append(cn)
            inits.append(a2.init_logp)
            transs.append(a2.trans_logp)
        t1, lp1 = gm.to_full_prob_reads_candidates(rc, gmp, np.stack(inits), np.stack(transs))
        t2, lp2 = gm.to_full_prob_reads_copy_nums(rc, gmp, np.stack(cns), min_cn)
        with np.errstate(invalid="ignore"):
            assert np.all((np.isneginf(lp1) & np.isneginf(lp2)) | (np.abs(lp1 - lp2) < TOL_LOGP))
        # and against the oracle on the host-built model of one candidate
        ol = oracle.Model(D.vectorised_to_phmm(D.SeqGraph(cns[1], sg.base, sg.edge_src, sg.edge_dst, None),
                                               arrays.param, min_cn)).full_prob_reads(reads, mp, True, n_threads=8)
        with np.errstate(invalid="ignore"):
            assert np.all((np.isneginf(ol) & np.isneginf(lp2[1])) | (np.abs(ol - lp2[1]) < TOL_LOGP))


def test_long_lists_use_bigger_class(gpu_lib, oracle):
    """node lists longer than 64/128 entries run in the 128/400-slot kernels."""
    arrays, sg, reads, om, _ = _setup(oracle, genome_len=500, n_reads=3, seed=13, read_len=40)
    gm = D.PHMMModel(arrays)
    N = arrays.n_nodes
    rng = np.random.default_rng(1)
    for width in (100, 300):
        pos_off, nodes = [0], []
        for r in reads:
            m = om.run(r).to_mapping(min(width, 390))
            for i in range(len(r)):
                nodes.extend(m.nodes(i))
                pos_off.append(len(nodes))
        mp = (np.array(pos_off, dtype=np.uint64), np.array(nodes, dtype=np.uint32), np.zeros(len(nodes)))
        rc = D.ReadCollection(reads)
        tot, lp = gm.to_full_prob_reads(rc, D.Mappings.from_arrays(rc, *mp))
        olp = om.full_prob_reads(reads, mp, True, n_threads=4)
        assert np.max(np.abs(lp - olp)) < TOL_LOGP, width


def test_mapping_errors(gpu_lib, oracle):
    arrays, sg, reads, om, mp = _setup(oracle, n_reads=2)
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    bad = mp[1].copy()
    bad[0] = arrays.n_nodes + 5
    with pytest.raises(D.PhmmError):
        gm.to_full_prob_reads(rc, D.Mappings.from_arrays(rc, mp[0], bad, mp[2]))
    other = D.ReadCollection(reads[:1])
    with pytest.raises(D.PhmmError):
        gm.to_full_prob_reads(other, D.Mappings.from_arrays(rc, *mp))


@pytest.mark.parametrize("cfg", [(300, 12, 0.001, 9), (600, 16, 0.001, 3), (600, 12, 0.01, 3), (900, 16, 0.003, 21)])
def test_adaptive_sparse_forward_matches_oracle(gpu_lib, oracle, cfg):
    """to_full_prob_reads without mappings = forward_sparse_score_only(use_max_ratio=true)
    (forward.rs:158-206): dense warm-up, per-read switch, adaptive frontier."""
    gl, k, p, seed = cfg
    arrays, sg = small_dbg_model(gl, k, p, seed=seed)
    reads = D.sample_reads(arrays, 10 ** 9, 150, seed=seed + 1, max_reads=40)
    reads = [r[: max(3, len(r) - (j * 11) % 140)] for j, r in enumerate(reads)]  # some end inside the warm-up
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    rc = D.ReadCollection(reads)
    tot, lp = gm.to_full_prob_reads(rc, None, True)
    olp = om.full_prob_reads(reads, None, True, n_threads=8)
    assert np.max(np.abs(lp - olp)) < 1e-6, np.abs(lp - olp).max()  # BASELINE.json: |d lnP| < 1e-6 per read
    lf, _, _ = gm.run_dense(rc, False, False)
    assert abs(lf.sum() - tot) < 1e-4 * len(reads)  # hmmv2/tests/dbg.rs:44-45


@pytest.mark.parametrize("cfg", [(600, 12, 0.01, 3, 40), (900, 16, 0.003, 21, 12), (300, 12, 0.001, 9, 5)])
def test_fixed_top_k_forward_matches_oracle(gpu_lib, oracle, cfg):
    """to_full_prob_reads(reads, None, false) = forward_sparse_score_only(use_max_ratio = false)
    (forward.rs:158-206): dense for the first n_warmup positions, then the n_active_nodes best."""
    gl, k, p, seed, n_active = cfg
    arrays, sg = small_dbg_model(gl, k, p, seed=seed)
    arrays.param = arrays.param.with_(n_active_nodes=n_active)
    reads = D.sample_reads(arrays, 10 ** 9, 150, seed=seed + 1, max_reads=30)
    reads = [r[: max(3, len(r) - (j * 11) % 145)] for j, r in enumerate(reads)]  # some end inside the warm-up
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    rc = D.ReadCollection(reads)
    tot, lp = gm.to_full_prob_reads(rc, None, False)
    olp = om.full_prob_reads(reads, None, False, n_threads=8)
    assert np.max(np.abs(lp - olp)) < 1e-6, np.abs(lp - olp).max()


@pytest.mark.parametrize("cfg", [(600, 12, 0.01, 3, 40), (300, 12, 0.001, 9, 6)])
def test_fixed_top_k_generate_mappings_matches_oracle(gpu_lib, oracle, cfg):
    """generate_mappings(reads, None, false) = run_sparse_adaptive(false) + to_mapping(n_active_nodes)
    (hint.rs:193-220, 124-131)."""
    gl, k, p, seed, n_active = cfg
    arrays, sg = small_dbg_model(gl, k, p, seed=seed, min_copy_num=1)
    arrays.param = arrays.param.with_(n_active_nodes=n_active)
    reads = D.sample_reads(arrays, 10 ** 9, 120, seed=seed + 1, max_reads=16)
    reads = [r[: max(1, len(r) - (j * 13) % 115)] for j, r in enumerate(reads)]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    rc = D.ReadCollection(reads)
    mp, nf = gm.generate_mappings(rc, None, False)
    omp, onf = om.generate_mappings(reads, None, False, n_threads=8)
    _compare_mappings(reads, mp.arrays(), omp, top_k=n_active)
    assert abs(nf.sum() - onf.sum()) < 1e-6 * max(1.0, onf.sum())


def test_generate_mappings_toy_kat(gpu_lib):
    """multi_dbg/posterior/test.rs:544-576 (hint_for_toy): best node per base on toy::repeat()."""
    import json, os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_hmmv2.json")))["toy_repeat_hints"]
    sg, k = D.toy_repeat()
    gm = D.PHMMModel(sg.to_non_zero_phmm(D.PHMMParams.uniform(kat["p"]).with_(n_warmup=k)))
    for read, best in kat["reads"].items():
        rc = D.ReadCollection([read.encode()])
        mp, nf = gm.generate_mappings(rc, None, True)
        assert [mp.nodes(0, i)[0] for i in range(len(read))] == best


def test_generate_mappings_toy_kat_from_the_documented_dbg_file(gpu_lib):
    """The same KAT on the graph read from the DBG text the reference documents (README.md:176-190)."""
    import json, os
    from dbgphmm_amd import formats as F
    here = os.path.dirname(__file__)
    kat = json.load(open(os.path.join(here, "golden", "kat_hmmv2.json")))["toy_repeat_hints"]
    dbg = F.read_dbg(os.path.join(here, "golden", "toy_repeat_readme.dbg"))
    gm = D.PHMMModel(dbg.to_seq_graph().to_non_zero_phmm(D.PHMMParams.uniform(kat["p"]).with_(n_warmup=dbg.k)))
    for read, best in kat["reads"].items():
        mp, nf = gm.generate_mappings(D.ReadCollection([read.encode()]), None, True)
        assert [mp.nodes(0, i)[0] for i in range(len(read))] == best


@pytest.mark.parametrize("case", ["case1", "case2"])
def test_map_nodes_kat(gpu_lib, case):
    """hint.rs:233-263 (mapping_node_convert) through the device kernel."""
    from test_oracle_kat import _map_nodes_case
    (po, nd, lp), mo, mn, exp = _map_nodes_case(case)
    gm = D.PHMMModel(D.mock_linear().to_phmm(D.PHMMParams.uniform(0.01)))  # any model with >= 6 nodes
    rc = D.ReadCollection([b"AC"])
    got = D.Mappings.from_arrays(rc, po, nd, lp).map_nodes(gm, mo, mn).arrays()
    assert got[1].tolist() == sum(exp["nodes"], [])
    assert np.max(np.abs(np.exp(got[2]) - np.array(sum(exp["probs"], [])))) < 1e-15


@pytest.mark.parametrize("cfg", [(300, 12, 0.001, 9, 30), (600, 16, 0.001, 3, 40), (600, 12, 0.01, 3, 24), (900, 16, 0.003, 21, 70)])
def test_generate_mappings_matches_oracle(gpu_lib, oracle, cfg):
    """generate_mappings(reads, None, true) = run_sparse_adaptive + to_mapping_by_score_ratio
    (hint.rs:193-220; freq.rs:60-68; backward.rs:101-142)."""
    gl, k, p, seed, n_reads = cfg
    arrays, sg = small_dbg_model(gl, k, p, seed=seed, min_copy_num=1)
    reads = D.sample_reads(arrays, 10 ** 9, 150, seed=seed + 1, max_reads=n_reads)
    reads = [r[: max(1, len(r) - (j * 13) % 149)] for j, r in enumerate(reads)]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    rc = D.ReadCollection(reads)
    mp, nf = gm.generate_mappings(rc, None, True)
    omp, onf = om.generate_mappings(reads, None, True, n_threads=8)
    _compare_mappings(reads, mp.arrays(), omp)
    if np.diff(mp.arrays()[0].astype(np.int64)).max() < 400:
        assert np.max(np.abs(nf - onf)) < 1e-6
    else:  # a capped list keeps an arbitrary subset of the nodes tied at the cut (see _compare_mappings)
        assert abs(nf.sum() - onf.sum()) < 1e-6
    assert np.max(np.abs(mp.to_node_freqs(arrays.n_nodes) - nf)) < 1e-12
    # the mappings drive the hinted likelihood (hmmv2/tests/dbg.rs:85-114): 1e-4 on the total
    # (reads of a few bases are left out: their first lists hold 400 of the N nodes inside the ratio)
    _, lp_hint = gm.to_full_prob_reads(rc, mp)
    _, lp_sparse = gm.to_full_prob_reads(rc, None)
    long_enough = np.array([len(r) >= 2 * k for r in reads])
    assert np.max(np.abs(lp_hint - lp_sparse)[long_enough], initial=0.0) < 1e-4


@pytest.mark.parametrize("cfg", [(600, 12, 0.01, 3, 24, True, 40), (900, 16, 0.003, 21, 40, True, 40),
                                 (600, 12, 0.01, 5, 16, False, 6), (300, 12, 0.001, 9, 12, False, 40)])
def test_generate_mappings_from_mappings_matches_oracle(gpu_lib, oracle, cfg):
    """generate_mappings(reads, Some(mappings), use_max_ratio) = run_with_mapping (freq.rs:72-76:
    forward_with_mapping + backward_with_mapping) + to_mapping_by_score_ratio / to_mapping(n_active)
    (hint.rs:193-220)."""
    gl, k, p, seed, n_reads, use_ratio, n_active = cfg
    arrays, sg = small_dbg_model(gl, k, p, seed=seed, min_copy_num=1)
    arrays.param = arrays.param.with_(n_active_nodes=n_active)
    reads = D.sample_reads(arrays, 10 ** 9, 150, seed=seed + 1, max_reads=n_reads)
    reads = [r[: max(1, len(r) - (j * 13) % 149)] for j, r in enumerate(reads)]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    rc = D.ReadCollection(reads)
    omp0, _ = om.generate_mappings(reads, None, True, n_threads=8)
    gmp0 = D.Mappings.from_arrays(rc, *omp0)
    mp, nf = gm.generate_mappings(rc, gmp0, use_ratio)
    omp, onf = om.generate_mappings(reads, omp0, use_ratio, n_threads=8)
    _compare_mappings(reads, mp.arrays(), omp, top_k=0 if use_ratio else n_active)
    assert np.max(np.abs(nf - onf)) < 1e-6
    # the forward score of the pass that produced them is kept with the mappings
    olp = om.full_prob_reads(reads, omp0, True, n_threads=8)
    assert np.max(np.abs(mp.read_logp()[1] - olp)) < TOL_LOGP
    # device-resident round trip: mappings made on the GPU feed the next call without a host copy
    mp1, _ = gm.generate_mappings(rc, None, True)
    mp2, nf2 = gm.generate_mappings(rc, mp1, True)
    omp2, onf2 = om.generate_mappings(reads, om.generate_mappings(reads, None, True, n_threads=8)[0], True, n_threads=8)
    assert abs(nf2.sum() - onf2.sum()) < 1e-6


def test_chunk_pipeline_matches_single_stream(gpu_lib, oracle, monkeypatch):
    """The sparse flow cuts the read groups into chunks that worker threads run concurrently on their
    own streams (sparse_dyn.hip).  Forced here on a small read set: same mappings, same ln P."""
    arrays, sg = small_dbg_model(900, 16, 0.003, seed=21, min_copy_num=1)
    reads = D.sample_reads(arrays, 10 ** 9, 150, seed=5, max_reads=200)
    reads = [r[: max(1, len(r) - (j * 13) % 149)] for j, r in enumerate(reads)]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    rc = D.ReadCollection(reads)
    monkeypatch.setenv("PHMM_WORKERS", "1")
    mp1, nf1 = gm.generate_mappings(rc, None, True)
    _, lp1 = gm.to_full_prob_reads(rc, None, True)
    monkeypatch.setenv("PHMM_WORKERS", "3")
    monkeypatch.setenv("PHMM_CHUNK_GROUPS", "1")
    monkeypatch.setenv("PHMM_PIPELINE_MIN_GROUPS", "2")
    mp3, nf3 = gm.generate_mappings(rc, None, True)
    _, lp3 = gm.to_full_prob_reads(rc, None, True)
    a1, a3 = mp1.arrays(), mp3.arrays()
    assert all(np.array_equal(x, y) for x, y in zip(a1, a3))
    assert np.array_equal(lp1, lp3) and np.array_equal(nf1, nf3)
    assert np.array_equal(mp1.read_logp()[1], mp3.read_logp()[1])
    omp, onf = om.generate_mappings(reads, None, True, n_threads=8)
    _compare_mappings(reads, a3, omp)


def test_full_size_properties_cfg3(gpu_lib):
    """BASELINE.json configs[2] at full size (100 kb diploid, 20x, L = 1000, k = 40: N = 1.3e5, 4e6 bases)
    through size-independent properties -- the oracle needs minutes for this, the GPU a second."""
    import bench
    arrays, reads, w = bench.build_workload("cfg3", 0)
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    n_bases = rc.total_bases()
    mp, nf = gm.generate_mappings(rc, None, True)
    po, nd, lp = mp.arrays()
    cnt = np.diff(po.astype(np.int64))
    # every position has a non-empty, descending list inside the ratio
    assert cnt.min() >= 1 and cnt.max() <= 400
    first = lp[po[:-1].astype(np.int64)]
    last = lp[po[1:].astype(np.int64) - 1]
    assert np.all(first - last < 30.0 + 1e-9)
    seg = np.repeat(np.arange(cnt.shape[0]), cnt)
    inner = np.ones(lp.shape[0], dtype=bool)
    inner[po[:-1].astype(np.int64)] = False  # first entry of each list
    assert np.all(np.diff(lp)[inner[1:]] <= 1e-12)
    mass = np.bincount(seg, weights=np.exp(lp), minlength=cnt.shape[0])
    # (Match + Ins emit the base: <= 1; the silent Del states crossed at this index add to it)
    assert mass.max() < 1.0 + 4.0 and np.median(mass) > 0.99 and mass.min() > 0.0
    # node usage: about one node per base; equals the sum over the lists
    assert abs(nf.sum() - n_bases) < 0.01 * n_bases
    assert abs(nf.sum() - np.exp(lp).sum()) < 1e-6 * n_bases
    assert np.max(np.abs(mp.to_node_freqs(arrays.n_nodes) - nf)) < 1e-9
    # the forward score kept with the mappings is the adaptive forward score; the hinted score walks a
    # subset of its paths (the first positions of a read keep 400 of the ~N nodes inside the ratio): a
    # little lower, by ~1e-3 per read
    tot_s, lp_s = gm.to_full_prob_reads(rc, None, True)
    assert np.array_equal(mp.read_logp()[1], lp_s)
    tot_h, lp_h = gm.to_full_prob_reads(rc, mp)
    assert np.all(lp_h <= lp_s + 1e-9) and 0.0 <= tot_s - tot_h < 5e-3 * len(reads)
    assert np.all(np.isfinite(lp_s)) and lp_s.max() < 0.0
    # a second call groups the reads differently (warm-up hints) and must return the same bits
    mp2, nf2 = gm.generate_mappings(rc, None, True)
    assert same_mappings(mp.arrays(), mp2.arrays()) and np.array_equal(nf, nf2)


def test_mappings_map_nodes_matches_oracle(gpu_lib, oracle):
    """Mapping::map_nodes (hint.rs:60-88): identity map, the parents map of hint_kp1_from_hint_k
    (multi_dbg.rs:1325-1335) and a purge-style map with dropped nodes (multi_dbg.rs:1783-1793)."""
    arrays, sg, reads, om, mp = _setup(oracle, n_reads=8, seed=11)
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    gmp = D.Mappings.from_arrays(rc, *mp)
    N = arrays.n_nodes
    # identity: nothing changes (up to the order of equal-probability entries)
    ident = gmp.map_nodes(gm, np.arange(N + 1), np.arange(N))
    a, b = ident.arrays(), mp
    assert np.array_equal(a[0], b[0]) and np.allclose(a[2], b[2], atol=1e-12)
    # node -> its parents (fan-out 0..3, shared images add up)
    order = np.argsort(arrays.edge_dst, kind="stable")
    par_off = np.concatenate([[0], np.cumsum(np.bincount(arrays.edge_dst, minlength=N))])
    par_nodes = arrays.edge_src[order]
    # purge: every third node disappears, the others are renumbered
    keep = np.arange(N) % 3 != 0
    newid = np.cumsum(keep) - 1
    pur_off = np.concatenate([[0], np.cumsum(keep.astype(np.int64))])
    pur_nodes = newid[keep]
    for mo, mn in ((par_off, par_nodes), (pur_off, pur_nodes)):
        got = gmp.map_nodes(gm, mo, mn).arrays()
        exp = oracle.map_nodes(mp, mo, mn)
        assert np.array_equal(got[0], exp[0])
        g = 0
        for i in range(len(exp[0]) - 1):
            s0, s1 = int(exp[0][i]), int(exp[0][i + 1])
            assert np.max(np.abs(got[2][s0:s1] - exp[2][s0:s1]), initial=0.0) < 1e-9
            assert sorted(got[1][s0:s1].tolist()) == sorted(exp[1][s0:s1].tolist())
    with pytest.raises(D.PhmmError):
        gmp.map_nodes(gm, np.arange(N + 1), np.full(N, N + 7))


def test_mappings_map_nodes_wide_fan_out(gpu_lib, oracle):
    """A node map with fan-out 12: the full first lists of a read (400 entries) have 4 800 images, more distinct
    ones than the LDS table of one pass holds.  The reference merges any number of images and keeps the 400 best
    (hint.rs:65-86); so does the kernel (key classes in several passes) -- no capacity error, no spinning probe."""
    arrays, sg = small_dbg_model(4000, 12, 0.01, seed=11)
    N = arrays.n_nodes
    rng = np.random.default_rng(3)
    reads = [b"ACGTACGTAC", b"TTGACA"]
    # synthetic lists: 400 / 37 / 0 / 400 ... distinct nodes per position with descending probabilities
    sizes = [400, 37, 0, 400, 1, 400, 250, 400, 64, 400, 400, 3, 400, 128, 399, 400]
    pos_off, nodes, logp = [0], [], []
    for sz in sizes:
        nodes.extend(rng.choice(N, sz, replace=False).tolist())
        logp.extend(np.sort(rng.uniform(-30.0, 0.0, sz))[::-1].tolist())
        pos_off.append(len(nodes))
    mp = (np.array(pos_off, dtype=np.uint64), np.array(nodes, dtype=np.uint32), np.array(logp))
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    gmp = D.Mappings.from_arrays(rc, *mp)
    assert N > 4500
    fan = 12
    mo = np.arange(N + 1, dtype=np.int64) * fan
    mn = ((np.arange(N)[:, None] * 7 + np.arange(fan)[None, :] * 509) % N).reshape(-1)
    got = gmp.map_nodes(gm, mo, mn).arrays()
    exp = oracle.map_nodes(mp, mo, mn)
    assert np.array_equal(got[0], exp[0]) and np.diff(got[0].astype(np.int64)).max() == 400
    for i in range(len(exp[0]) - 1):
        s0, s1 = int(exp[0][i]), int(exp[0][i + 1])
        assert np.max(np.abs(got[2][s0:s1] - exp[2][s0:s1]), initial=0.0) < 1e-9, i
        if s1 - s0 < 400:
            assert sorted(got[1][s0:s1].tolist()) == sorted(exp[1][s0:s1].tolist()), i
        else:  # capped: the nodes tied at the cut are an arbitrary subset
            above = exp[2][s0:s1] > exp[2][s1 - 1] + 1e-9
            assert set(exp[1][s0:s1][above].tolist()) <= set(got[1][s0:s1].tolist()), i


def test_edge_cases_match_oracle(gpu_lib, oracle):
    """odd inputs through the adaptive flow: bases outside ACGT, a read far longer than the rest, a read
    that matches nowhere, one-base reads -- and an empty read set."""
    arrays, sg = small_dbg_model(500, 12, 0.01, seed=17, min_copy_num=1)
    rng = np.random.default_rng(5)
    reads = D.sample_reads(arrays, 10 ** 9, 120, seed=3, max_reads=6)
    long_read = b"".join(D.sample_reads(arrays, 10 ** 9, 400, seed=9, max_reads=6))  # 2400 bases, chimeric
    junk = bytes(rng.choice(list(b"ACGT"), size=90).tolist())
    with_n = bytearray(reads[0])
    with_n[10] = ord("N")
    with_n[40] = ord("n")
    reads = reads + [long_read, junk, bytes(with_n), b"A", b"G"]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    rc = D.ReadCollection(reads)
    tot, lp = gm.to_full_prob_reads(rc, None, True)
    olp = om.full_prob_reads(reads, None, True, n_threads=8)
    junk_ix = len(reads) - 4
    ok = np.ones(len(reads), dtype=bool)
    ok[junk_ix] = False
    assert np.max(np.abs(lp - olp)[ok]) < 1e-6
    # The read that matches nowhere keeps MORE than 400 nodes inside the score ratio at every sparse position:
    # the regime where the reference's 400-slot SparseVec overflows and whose semantics are unpinned
    # (DESIGN.md section 2).  GPU and oracle both drop inserts there but not the same ones; what can
    # be asserted: the sparse scores are lower bounds of the dense one and the GPU keeps at least as much mass.
    lfd, _, _ = gm.run_dense(D.ReadCollection([junk]), False, False)
    assert olp[junk_ix] - 1e-6 <= lp[junk_ix] <= lfd[0] + 1e-9 and abs(lp[junk_ix] - olp[junk_ix]) < 1.0
    keep = [r for j, r in enumerate(reads) if j != junk_ix]
    mp, nf = gm.generate_mappings(D.ReadCollection(keep), None, True)
    omp, onf = om.generate_mappings(keep, None, True, n_threads=8)
    _compare_mappings(keep, mp.arrays(), omp)
    mpj, _ = gm.generate_mappings(rc, None, True)  # with the junk read: runs, lists stay inside the ratio
    assert np.diff(mpj.arrays()[0].astype(np.int64)).max() <= 400
    lf, lb, nfd = gm.run_dense(D.ReadCollection(reads[-4:]))
    olf, olb, onfd = om.run_dense_reads(reads[-4:], n_threads=8)
    assert np.max(np.abs(lf - olf)) < TOL_LOGP and np.max(np.abs(nfd - onfd)) < 1e-8
    # The chimeric read under the DENSE recursion leaves the range of the scaled linear domain:
    # the placement that wins after a junction was, at the junction, thousands of nats below the then-best path.
    # The dense driver notices (certify_dense) and recomputes the read in the log domain (exact_dense.hip).
    # (The sparse modes drop such paths in the reference too -- 30 nats below the best -- which is why the
    # adaptive scores above agree.)
    lfc, _, _ = gm.run_dense(D.ReadCollection([long_read]), False, False)
    olfc, _, _ = om.run_dense_reads([long_read], n_threads=1)
    assert abs(lfc[0] - olfc[0]) < 1e-6
    # no reads: a zero total, and generate_mappings refuses (nothing to map)
    empty = D.ReadCollection([])
    t0, l0 = gm.to_full_prob_reads(empty, None, True)
    assert t0 == 0.0 and l0.shape == (0,)
    with pytest.raises(D.PhmmError):
        gm.generate_mappings(empty, None, True)


def test_impossible_reads_zero_error_model(gpu_lib, oracle):
    """p = 0 parameters (PHMMParams::zero_error): a read with one wrong base has probability 0.  ln P = -inf
    comes back as -inf, its mapping lists are empty or all -inf, the other reads are unaffected."""
    sg = D.mock_linear()
    arrays = sg.to_phmm(D.PHMMParams.zero_error().with_(n_warmup=2, warmup_threshold=3))
    reads = [b"CGATC", b"CGATT", b"TTCGAT"]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    rc = D.ReadCollection(reads)
    _, lp = gm.to_full_prob_reads(rc, None, True)
    olp = om.full_prob_reads(reads, None, True, n_threads=2)
    assert np.isneginf(lp[1]) and np.isneginf(olp[1])
    assert np.max(np.abs(lp[[0, 2]] - olp[[0, 2]])) < 1e-9
    mp, nf = gm.generate_mappings(rc, None, True)
    po, nd, l2 = mp.arrays()
    off = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    bad = slice(int(po[off[1]]), int(po[off[2]]))
    assert np.all(np.isneginf(l2[bad])) or bad.start == bad.stop
    assert np.all(np.isfinite(nf)) and abs(nf.sum() - (len(reads[0]) + len(reads[2]))) < 1e-6


def test_files_to_likelihood_end_to_end(gpu_lib, oracle, tmp_path):
    """what the reference's `infer` resumes from (bin/infer.rs:47-48, 90-94): DBG + MAP + FASTA files -> model,
    reads, mappings -> hinted read-set likelihood; equal to the in-memory path bit for bit."""
    from dbgphmm_amd import formats as F
    from test_formats import _kmers_first_occurrence
    k = 12
    hap = D.random_genome(400, seed=4)
    haps = [hap, D.diverge(hap, 0.02, seed=5)]
    sg = D.dbg_from_haplotypes(haps, k)
    param = D.PHMMParams.uniform(0.01).with_(n_warmup=k)
    arrays = D.vectorised_to_phmm(sg, param, 1)
    reads = D.sample_reads(arrays, 10 ** 9, 100, seed=2, max_reads=10)
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    mp, _ = gm.generate_mappings(rc, None, True)
    tot0, lp0 = gm.to_full_prob_reads(rc, mp)
    kmers, cns = _kmers_first_occurrence(haps, k)
    F.write_dbg(str(tmp_path / "g.dbg"), F.dbg_from_seq_graph_kmers(kmers, cns, k))
    F.write_map(str(tmp_path / "m.mpz"), reads, mp.arrays(), k=k, n_edges_full=arrays.n_nodes)
    F.write_fasta(str(tmp_path / "r.fa"), reads)
    sg2 = F.read_dbg(str(tmp_path / "g.dbg")).to_seq_graph()
    reads2 = F.read_fasta(str(tmp_path / "r.fa"))
    _, arrs = F.read_map(str(tmp_path / "m.mpz"))
    gm2 = D.PHMMModel(D.vectorised_to_phmm(sg2, param, 1))
    rc2 = D.ReadCollection(reads2)
    tot1, lp1 = gm2.to_full_prob_reads(rc2, D.Mappings.from_arrays(rc2, *arrs))
    assert reads2 == reads and np.max(np.abs(lp1 - lp0)) < 1e-12
    olp = oracle.Model(D.vectorised_to_phmm(sg2, param, 1)).full_prob_reads(reads2, arrs, True, n_threads=4)
    assert np.max(np.abs(lp1 - olp)) < TOL_LOGP


@pytest.mark.parametrize("over", [dict(n_warmup=5, warmup_threshold=50), dict(active_node_max_ratio=15.0),
                                  dict(n_warmup=30, warmup_threshold=399), dict(warmup_threshold=3, active_node_max_ratio=40.0)])
def test_unusual_frontier_parameters_match_oracle(gpu_lib, oracle, over):
    """n_warmup / warmup_threshold / active_node_max_ratio away from their defaults: forced switches at a short
    warm-up, a threshold next to the 400 cap, a narrow and a wide ratio (forward.rs:107-137, table.rs:134-149)."""
    arrays, sg = small_dbg_model(700, 12, 0.01, seed=41, min_copy_num=1)
    arrays.param = arrays.param.with_(**over)
    reads = D.sample_reads(arrays, 10 ** 9, 140, seed=3, max_reads=20)
    reads = [r[: max(2, len(r) - (j * 11) % 131)] for j, r in enumerate(reads)]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    rc = D.ReadCollection(reads)
    _, lp = gm.to_full_prob_reads(rc, None, True)
    olp = om.full_prob_reads(reads, None, True, n_threads=8)
    mp, nf = gm.generate_mappings(rc, None, True)
    omp, onf = om.generate_mappings(reads, None, True, n_threads=8)
    ratio = arrays.param.active_node_max_ratio
    gpo, gnd, glp = mp.arrays()
    opo, ond, olp2 = omp
    if over.get("n_warmup", 12) >= 12:
        assert np.max(np.abs(lp - olp)) < 1e-6, np.abs(lp - olp).max()
        # (entries within rounding of the ratio cut-off may fall on either side: compare what is above e^-20)
        # (with a ratio of 40 the frontier keeps nodes 17 nats under the best whose posterior is a sum of terms
        # near the sparse cut-off: their log carries the dropped tail, 1e-5 there; the top entries agree to 1e-9)
        # (a threshold of 3 forces the switch at n_warmup with more than 400 nodes inside a ratio of 40: the first sparse
        # columns overflow the 400-slot vector and the two restatements keep different tails -- entries 25+ nats down
        # then differ in the third digit; everything above e^-20 agrees)
        _compare_mappings(reads, mp.arrays(), omp, ratio=ratio, tol=1e-6 if ratio <= 30 else 1e-4, deep=ratio <= 30)
        assert abs(nf.sum() - onf.sum()) < 1e-6 * max(1.0, onf.sum())
    else:
        # A warm-up of 5 columns forces the switch while thousands of nodes are inside the ratio: 400 are kept
        # and the frontier overflows at every step of the first dozen positions -- the unpinned regime of the
        # 400-slot SparseVec.  Both restatements then lose some reads' true path (their
        # scores fall 20-60 nats under the dense one, by different amounts); where the oracle keeps it the GPU
        # agrees to 1e-3, and a sparse score never exceeds the dense one.
        lfd, _, _ = gm.run_dense(rc, False, False)
        kept = np.abs(olp - lfd) < 1e-3
        assert kept.sum() >= len(reads) // 2 and np.max(np.abs(lp - olp)[kept]) < 1e-3
        assert np.all(lp <= lfd + 1e-9)
    first = glp[gpo[:-1].astype(np.int64)]
    last = glp[gpo[1:].astype(np.int64) - 1]
    assert np.all(first - last < ratio + 1e-9)


@pytest.mark.parametrize("cfg", [(600, 12, 0.01, 3, 40), (900, 16, 0.003, 21, 12), (300, 12, 0.001, 9, 5)])
def test_backward_sparse_scores_match_oracle(gpu_lib, oracle, cfg):
    """to_full_prob_sparse_backward (freq.rs:153-163) = backward_sparse(read).full_prob() (backward.rs:146-185):
    dense over the last n_warmup positions, then top_nodes(n_active_nodes) + adaptive b_step."""
    gl, k, p, seed, n_active = cfg
    arrays, sg = small_dbg_model(gl, k, p, seed=seed)
    arrays.param = arrays.param.with_(n_active_nodes=n_active)
    reads = D.sample_reads(arrays, 10 ** 9, 150, seed=seed + 1, max_reads=30)
    reads = [r[: max(3, len(r) - (j * 11) % 145)] for j, r in enumerate(reads)]  # some are all warm-up
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    tot, lp = gm.to_full_prob_sparse_backward(D.ReadCollection(reads))
    olp = np.array([om.backward(r, oracle.BWD_SPARSE).full_prob() for r in reads])
    assert np.max(np.abs(lp - olp)) < 1e-6, np.abs(lp - olp).max()
    assert abs(tot - olp.sum()) < 1e-6 * len(reads)
    # forward and backward totals of the sparse modes agree to 0.01 on the read set (hmmv2/tests/dbg.rs:44-45)
    ftot, _ = gm.to_full_prob_reads(D.ReadCollection(reads), None, False)
    assert abs(ftot - tot) < 0.01 * len(reads)


@pytest.mark.parametrize("cfg", [(400, 12, 0.01, 3, 40, 100), (300, 12, 0.003, 9, 6, 70), (300, 12, 0.01, 5, 40, 9),
                                 (600, 12, 0.01, 7, 300, 60)])  # the last: the grown sets reach the 400-slot capacity
def test_backward_sparse_tables_match_oracle(gpu_lib, oracle, cfg):
    """Every column of backward_sparse: the same elements (m, i on to_parents_and_us(top); d on the grown sets)
    with the same values, the dense tail, and the Begin-state scalars."""
    gl, k, p, seed, n_active, rl = cfg
    arrays, sg = small_dbg_model(gl, k, p, seed=seed)
    arrays.param = arrays.param.with_(n_active_nodes=n_active)
    read = D.sample_reads(arrays, 10 ** 9, rl, seed=seed + 2, max_reads=1)[0]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    gt, dense = gm.backward_sparse(read)
    ot = om.backward(read, oracle.BWD_SPARSE)
    L = len(read)
    assert dense.tolist() == [ot.is_dense(i) for i in range(L)]
    for i in range(L):
        m, ins, d, s = ot.table(i)
        for name, g, o in (("m", gt.m[i], m), ("i", gt.i[i], ins), ("d", gt.d[i], d)):
            floor = np.max(o) - 600.0 if dense[i] else None  # the dense kernel's scaled linear range
            assert np.all(finite_close(g, o, 1e-8, floor)), (i, name, dense[i])
        assert abs(gt.scal[i, 0] - s[0]) < 1e-8 and abs(gt.scal[i, 1] - s[1]) < 1e-8, (i, gt.scal[i], s)
        assert np.isneginf(gt.scal[i, 2]) and np.isneginf(s[2])
    assert abs(gt.scal[0, 0] - ot.full_prob()) < 1e-8


def test_backward_sparse_rejects_zero_warmup(gpu_lib):
    """n_warmup = 0: the reference's backward_sparse panics in last_table() (table.rs:388)."""
    arrays, _ = small_dbg_model(300, 12, 0.01, seed=5)
    arrays.param = arrays.param.with_(n_warmup=0)
    with pytest.raises(D.PhmmError):  # (rejected with the parameters already: check_params)
        D.PHMMModel(arrays).to_full_prob_sparse_backward(D.ReadCollection([b"ACGTACGTAC"]))


@pytest.mark.parametrize("cfg", [(600, 12, 0.01, 3, 40, 150), (300, 12, 0.003, 9, 6, 70), (400, 16, 0.01, 5, 20, 40)])
def test_run_sparse_node_freqs_match_oracle(gpu_lib, oracle, cfg):
    """run_sparse (freq.rs:51-55) + to_node_freqs (freq.rs:245-255) summed over the reads: dense x sparse,
    sparse x dense, sparse x sparse and (short reads) dense x dense merged indices."""
    gl, k, p, seed, n_active, rl = cfg
    arrays, sg = small_dbg_model(gl, k, p, seed=seed, min_copy_num=1)
    arrays.param = arrays.param.with_(n_active_nodes=n_active)
    reads = D.sample_reads(arrays, 10 ** 9, rl, seed=seed + 1, max_reads=24)
    reads = [r[: max(2, len(r) - (j * 11) % (rl - 5))] for j, r in enumerate(reads)]  # all-warm-up and overlap cases
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    lf, lb, nf = gm.run_sparse(D.ReadCollection(reads))
    onf = np.zeros(arrays.n_nodes)
    nw = arrays.param.n_warmup
    for j, r in enumerate(reads):
        o = om.run_sparse(r)
        assert abs(lf[j] - o.to_full_prob_forward()) < 1e-6, j
        assert abs(lb[j] - o.to_full_prob_backward()) < 1e-6, j
        # state_probs = sum over merged indices of the emit probs (freq.rs:236-243), summed here without the
        # 400-element capacity of a sparse accumulator
        one = np.zeros(arrays.n_nodes)
        for jj in range(len(r) + 1):
            m, i, d, _ = o.to_emit_probs(jj)
            one += np.exp(m) + np.exp(i) + np.exp(d)
        if not (nw < len(r) < 2 * nw and arrays.n_nodes > 400):
            assert np.max(np.abs(one - o.to_node_freqs())) < 1e-9, j
        # (nw < len < 2 nw: dense emit-prob tables are added INTO a sparse accumulator; with N > 400 the
        # oracle's accumulator drops elements and the reference's SparseVec behaviour there is unpinned)
        onf += one
    assert np.max(np.abs(nf - onf)) < 1e-9 * len(reads), np.abs(nf - onf).max()
    assert abs(nf.sum() - sum(map(len, reads))) < 0.02 * sum(map(len, reads))


def _hub_graph(seed=11, fan=7, backbone=260, branch=18):
    """A backbone with a hub: node 60 branches into `fan` arms of `branch` nodes that rejoin at one node, so that
    one node has `fan` children and another `fan` parents (more than the 5 the packed frontier records hold:
    the one-lane-per-node kernels step aside for the generic vector kernels)."""
    rng = np.random.default_rng(seed)
    bases, src, dst = [], [], []

    def add(b):
        bases.append(b)
        return len(bases) - 1

    prev = None
    hub = join = None
    for k in range(backbone):
        v = add(int(rng.choice(list(b"ACGT"))))
        if prev is not None:
            src.append(prev)
            dst.append(v)
        prev = v
        if k == 60:
            hub = v
            ends = []
            for _ in range(fan):
                p = hub
                for _ in range(branch):
                    w = add(int(rng.choice(list(b"ACGT"))))
                    src.append(p)
                    dst.append(w)
                    p = w
                ends.append(p)
            join = add(int(rng.choice(list(b"ACGT"))))
            for e in ends:
                src.append(e)
                dst.append(join)
            prev = join
    n = len(bases)
    return D.SeqGraph(np.ones(n, dtype=np.int64), np.array(bases, dtype=np.uint8), np.array(src, dtype=np.uint32),
                      np.array(dst, dtype=np.uint32), None)


@pytest.mark.parametrize("no_lean", [False, True])
def test_generic_frontier_class_matches_oracle(gpu_lib, oracle, no_lean, monkeypatch):
    """The generic vector kernels of the <= 64 / 128-node class (sparse_forward_kernel<128>, sparse_backward_kernel<64>)
    only run where the one-lane-per-node kernels cannot: nodes of degree above 5, or PHMM_NO_LEAN.  Both ways
    must give the oracle's scores, mapping lists and node usage."""
    if no_lean:
        monkeypatch.setenv("PHMM_NO_LEAN", "1")
        arrays, sg = small_dbg_model(600, 12, 0.01, seed=3, min_copy_num=1)
    else:
        sg = _hub_graph()
        arrays = D.vectorised_to_phmm(sg, D.PHMMParams.uniform(0.01).with_(n_warmup=12), 1)
    reads = D.sample_reads(arrays, 10 ** 9, 150, seed=4, max_reads=24)
    reads = [r[: max(5, len(r) - (j * 7) % 60)] for j, r in enumerate(reads)]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    rc = D.ReadCollection(reads)
    tot, lp = gm.to_full_prob_reads(rc, None, True)
    olp = om.full_prob_reads(reads, None, True, n_threads=8)
    assert np.max(np.abs(lp - olp)) < 1e-6
    mp, nf = gm.generate_mappings(rc, None, True)
    omp, onf = om.generate_mappings(reads, None, True, n_threads=8)
    _compare_mappings(reads, mp.arrays(), omp)
    assert np.max(np.abs(nf - onf)) < 1e-6 * len(reads)


def test_overfull_dense_column_selection_paths_agree(gpu_lib, monkeypatch):
    """A read that fits nowhere keeps more than 400 nodes inside the score ratio in the dense head of its mapping:
    the 400 best come from a histogram pass + boundary-bin ranking, with an eight-pass radix select as fallback
    (mapping_flow.hip: block_top_from_column).  Both must give the same lists."""
    arrays, sg = small_dbg_model(5000, 12, 0.01, seed=17, min_copy_num=1)
    rng = np.random.default_rng(5)
    junk = [bytes(rng.choice(list(b"ACGT"), size=15).tolist()) for _ in range(3)]  # short: nothing narrows them down
    reads = D.sample_reads(arrays, 10 ** 9, 100, seed=3, max_reads=5) + junk
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    mp1, nf1 = gm.generate_mappings(rc, None, True)
    po1, nd1, lp1 = mp1.arrays()
    assert np.diff(po1.astype(np.int64)).max() == 400  # the junk reads fill the list capacity
    monkeypatch.setenv("PHMM_FORCE_RADIX", "1")
    mp2, nf2 = gm.generate_mappings(rc, None, True)
    po2, nd2, lp2 = mp2.arrays()
    assert np.array_equal(po1, po2) and np.array_equal(nd1, nd2) and np.array_equal(lp1, lp2)
    assert np.array_equal(nf1, nf2)


def test_destroyed_mappings_leave_their_device_buffers_for_the_next_call(gpu_lib):
    """The device CSR of a Mappings that is destroyed is kept by the device's pool (DevicePool::recycle: buffers of
    1 MB and more, at most 6 / 8 GB) and handed to the next generate_mappings call instead of a hipFree + hipMalloc
    pair; it counts as workspace and phmm_release_workspace() gives it back."""
    from dbgphmm_amd import _ffi
    L = _ffi.lib()
    arrays, sg = small_dbg_model(3000, 16, 0.003, seed=33, min_copy_num=1)
    reads = D.sample_reads(arrays, 10 ** 9, 400, seed=5, max_reads=150)
    gm, rc = D.PHMMModel(arrays), D.ReadCollection(reads)
    _ffi.check(L.phmm_release_workspace())  # (spares of earlier tests in this process)
    mp1, nf1 = gm.generate_mappings(rc, None, True)
    a1 = [x.copy() for x in mp1.arrays()]
    assert a1[2].nbytes >= 1 << 20  # (the ln p array is big enough to be kept)
    held = L.phmm_workspace_bytes()
    del mp1
    spare = L.phmm_workspace_bytes() - held
    assert spare >= a1[2].nbytes
    mp2, nf2 = gm.generate_mappings(rc, None, True)
    assert L.phmm_workspace_bytes() - held < 1 << 16  # the spares went into mp2 (a hinted call may grow a control array)
    assert same_mappings(a1, mp2.arrays()) and np.array_equal(nf1, nf2)
    del mp2
    assert spare <= L.phmm_workspace_bytes() - held < spare + (1 << 16)
    _ffi.check(L.phmm_release_workspace())
    assert L.phmm_workspace_bytes() == 0
    mp3, _ = gm.generate_mappings(rc, None, True)
    assert same_mappings(a1, mp3.arrays())
    # the cache is bounded: the oldest spares go when more than six are waiting
    keep = [gm.generate_mappings(rc, None, True)[0] for _ in range(5)]
    held = L.phmm_workspace_bytes()
    del keep, mp3
    assert 0 < L.phmm_workspace_bytes() - held <= 6 * a1[2].nbytes
