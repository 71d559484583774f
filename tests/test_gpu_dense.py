"""GPU parity: dense forward / backward / posteriors through the C ABI vs the oracle and
the reference's own known-answer values (tests/golden/kat_hmmv2.json)."""
import json
import os

import numpy as np
import pytest

import dbgphmm_amd as D
from helpers import finite_close, small_dbg_model

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_hmmv2.json")))
# |delta ln P| per read: BASELINE.json's bar is 1e-6; the scaled-f64 path is far inside it
TOL_LOGP = 1e-9
TOL_TABLE = 1e-9   # per table entry, log space, where finite (tests/hmm.rs:69-70)
TOL_FREQ = 1e-9


def test_forward_kat_zero_error(gpu_lib):
    phmm = D.PHMMModel(D.mock_linear().to_phmm(D.PHMMParams.zero_error()))
    r = phmm.forward(b"CGATC")
    k = KAT["forward_zero_error"]
    assert abs(r.m[2][5] - k["t2_m5"]) < 1e-5
    assert abs(r.m[3][6] - k["t3_m6"]) < 1e-5
    assert abs(r.m[4][7] - k["t4_m7"]) < 1e-5
    assert abs(r.e[4] - k["t4_e"]) < 1e-5
    assert np.all(np.isneginf(r.i)) and np.all(np.isneginf(r.d))
    assert np.isneginf(phmm.forward(b"CGATT").e[4])


def test_forward_backward_kat_high_error(gpu_lib):
    phmm = D.PHMMModel(D.mock_linear().to_phmm(D.PHMMParams.high_error()))
    k = KAT["high_error"]
    r = phmm.forward(b"CGATC")
    assert abs(r.e[4] - k["fwd_CGATC_e"]) < 1e-5
    assert abs(r.m[4][7] - k["fwd_CGATC_t4_m7"]) < 1e-5
    r2 = phmm.forward(b"CGATT")
    assert abs(r2.e[4] - k["fwd_CGATT_e"]) < 1e-5
    assert abs(r2.e[3] - r.e[3]) < 1e-5
    b = phmm.backward(b"CGATC")
    assert abs(b.m[0][2] - k["bwd_CGATC_t0_m2"]) < 1e-5
    assert abs(b.mb[0] - k["bwd_CGATC_mb"]) < 1e-5
    assert abs(phmm.backward(b"CGATT").mb[0] - k["bwd_CGATT_mb"]) < 1e-5


def test_backward_kat_zero_error(gpu_lib):
    phmm = D.PHMMModel(D.mock_linear().to_phmm(D.PHMMParams.zero_error()))
    b = phmm.backward(b"CGATC")
    k = KAT["backward_zero_error"]
    assert abs(b.mb[0] - k["t0_mb"]) < 1e-5
    for (t, n, v) in k["m"]:
        assert abs(b.m[t][n] - v) < 1e-5
    assert np.isneginf(phmm.backward(b"CGATT").mb[0])


def test_node_freq_kat(gpu_lib):
    phmm = D.PHMMModel(D.mock_linear().to_phmm(D.PHMMParams.zero_error()))
    o = phmm.run(b"CGATC")
    assert abs(o.to_full_prob_forward() - o.to_full_prob_backward()) < 1e-7
    nf = o.to_node_freqs()
    assert np.allclose(nf, [0, 0, 0, 1, 1, 1, 1, 1, 0, 0], atol=1e-5)
    phmm = D.PHMMModel(D.mock_linear().to_phmm(D.PHMMParams.default()))
    nf = phmm.run(b"ATTCGTCGT").to_node_freqs()  # one deletion: every node used once
    assert np.allclose(nf, 1.0, atol=0.01)


@pytest.mark.parametrize("case", ["linear_high", "crossing_on", "crossing_off", "toy_repeat", "dbg_diploid"])
def test_tables_match_oracle(gpu_lib, oracle, case):
    rng = np.random.default_rng(3)
    if case == "linear_high":
        arrays = D.mock_linear().to_phmm(D.PHMMParams.high_error())
        reads = [b"CGATC", b"ATTCGTCGT", b"T", b"GGGGGGGG"]
    elif case.startswith("crossing"):
        arrays = D.mock_crossing(case.endswith("on")).to_phmm(D.PHMMParams.default())
        reads = [b"ATTAGGAGCAGCTGATAGGG", b"ATTAGGAGCA", b"TGCTCTGGCGCGAAGATGAG"]
    elif case == "toy_repeat":
        sg, k = D.toy_repeat()
        arrays = sg.to_uniform_phmm(D.PHMMParams.uniform(0.01).with_(n_warmup=k))
        reads = [b"CCCAG", b"GCAGCAGG", b"TCCCAGCAGCAGCAGGAA"]
    else:
        arrays, _ = small_dbg_model(200, 10, 0.02, seed=11)
        reads = D.sample_reads(arrays, 300, 60, seed=5)[:4]
        reads.append(bytes(rng.choice(list(b"ACGT"), size=40).tolist()))
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    for read in reads:
        f, b = gm.forward(read), gm.backward(read)
        of, ob = om.forward(read), om.backward(read)
        for i in range(len(read)):
            m, ins, d, s = of.table(i)
            floor = max(m.max(), ins.max(), d.max()) - 600.0
            assert finite_close(f.m[i], m, TOL_TABLE, floor).all(), (case, read, i, "Fm")
            assert finite_close(f.i[i], ins, TOL_TABLE, floor).all(), (case, read, i, "Fi")
            assert finite_close(f.d[i], d, TOL_TABLE, floor).all(), (case, read, i, "Fd")
            assert finite_close(f.scal[i], s, TOL_TABLE).all(), (case, read, i, "Fscal", f.scal[i], s)
            m, ins, d, s = ob.table(i)
            floor = max(m.max(), ins.max(), d.max()) - 600.0
            assert finite_close(b.m[i], m, TOL_TABLE, floor).all(), (case, read, i, "Bm")
            assert finite_close(b.i[i], ins, TOL_TABLE, floor).all(), (case, read, i, "Bi")
            assert finite_close(b.d[i], d, TOL_TABLE, floor).all(), (case, read, i, "Bd")
            assert finite_close(b.scal[i], s, TOL_TABLE).all(), (case, read, i, "Bscal", b.scal[i], s)


@pytest.mark.parametrize("n_reads", [1, 3, 8, 37, 70])
def test_run_dense_matches_oracle(gpu_lib, oracle, n_reads):
    arrays, _ = small_dbg_model(400, 12, 0.01, seed=21)
    reads = D.sample_reads(arrays, 10 ** 9, 80, seed=n_reads, max_reads=n_reads)
    # ragged lengths
    reads = [r[: max(1, len(r) - (j * 7) % 31)] for j, r in enumerate(reads)]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    lf, lb, nf = gm.run_dense(D.ReadCollection(reads))
    olf, olb, onf = om.run_dense_reads(reads, n_threads=8)
    assert np.max(np.abs(lf - olf)) < TOL_LOGP
    assert np.max(np.abs(lb - olb)) < TOL_LOGP
    assert np.max(np.abs(nf - onf)) < TOL_FREQ * max(1, n_reads)
    # posterior mass: about one node per emitted base (Del states and the trailing-deletion
    # asymmetry between fe and b_init move it by ~p)
    tot = sum(len(r) for r in reads)
    assert abs(nf.sum() - tot) < 0.03 * tot
    # bit-reproducible
    lf2, lb2, nf2 = gm.run_dense(D.ReadCollection(reads))
    assert np.array_equal(lf, lf2) and np.array_equal(lb, lb2) and np.array_equal(nf, nf2)


def test_workspace_chunking(gpu_lib, oracle):
    """a tiny workspace limit forces several chunks; results must not change."""
    arrays, _ = small_dbg_model(300, 12, 0.01, seed=5)
    reads = D.sample_reads(arrays, 10 ** 9, 50, seed=9, max_reads=40)
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    lf, lb, nf = gm.run_dense(rc)
    gpu_lib.phmm_set_workspace_limit(4 << 20)
    try:
        lf2, lb2, nf2 = gm.run_dense(rc)
    finally:
        gpu_lib.phmm_set_workspace_limit(0)
    assert np.allclose(lf, lf2, atol=1e-12) and np.allclose(lb, lb2, atol=1e-12)
    assert np.allclose(nf, nf2, atol=1e-10)


def test_errors(gpu_lib):
    with pytest.raises(D.PhmmError):
        D.ReadCollection([b"ACGT", b""])  # empty read: the reference panics
    arrays = D.mock_linear().to_phmm(D.PHMMParams.default())
    arrays.edge_dst = arrays.edge_dst.copy()
    arrays.edge_dst[0] = 99
    with pytest.raises(D.PhmmError):
        D.PHMMModel(arrays)


def test_edge_freq_kats(gpu_lib):
    """freq.rs:517-609: transition posteriors on the 10-node linear mock."""
    gm = D.PHMMModel(D.mock_linear().to_phmm(D.PHMMParams.zero_error()))
    _, ef, _ = gm.run_dense_edge_freqs(D.ReadCollection([b"CGATC"]))
    assert np.all(ef[[0, 1, 2, 7, 8]] < 1e-4) and np.all(ef[3:7] > 0.9999)
    gm = D.PHMMModel(D.mock_linear().to_phmm(D.PHMMParams.default()))
    _, ef, _ = gm.run_dense_edge_freqs(D.ReadCollection([b"ATTCGTCGT"]))
    assert np.allclose(ef, 0.99, atol=0.01)


@pytest.mark.parametrize("n_reads", [1, 5, 70])
def test_edge_and_init_freqs_match_oracle(gpu_lib, oracle, n_reads):
    """PHMMOutput::to_edge_and_init_freqs (freq.rs:276-298, 332-389) summed over the reads."""
    arrays, _ = small_dbg_model(300, 12, 0.01, seed=5)
    reads = D.sample_reads(arrays, 10 ** 9, 60, seed=n_reads, max_reads=n_reads)
    reads = [r[: max(1, len(r) - (j * 7) % 31)] for j, r in enumerate(reads)]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    lf, ef, nf = gm.run_dense_edge_freqs(D.ReadCollection(reads))
    oef, onf = np.zeros(arrays.n_edges), np.zeros(arrays.n_nodes)
    for r in reads:
        e1, n1 = om.run(r).to_edge_and_init_freqs()
        oef += e1
        onf += n1
    assert np.max(np.abs(ef - oef)) < TOL_FREQ * max(1, n_reads)
    assert np.max(np.abs(nf - onf)) < TOL_FREQ * max(1, n_reads)
    # every read leaves the Begin states about once (the forward `e` and the backward b_init treat a
    # trailing deletion differently, as for the node posteriors above)
    assert abs(nf.sum() - len(reads)) < 0.01 * len(reads)


def test_edge_and_init_freqs_of_a_read_below_exp_range(gpu_lib, oracle):
    """A long noisy read whose ln P is below -709 (exp(-ln P) overflows a double): the Begin-state posteriors multiply
    the InsBegin chain -- itself below the double range from base ~105 on -- with that weight; the product is formed
    in the exponent (0 * inf was NaN; found by the randomised cases)."""
    arrays, _ = small_dbg_model(2600, 12, 0.05, seed=31)
    read = D.sample_reads(arrays, 10 ** 9, 2400, seed=4, max_reads=1)[0]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    lf, ef, nf = gm.run_dense_edge_freqs(D.ReadCollection([read]))
    o = om.run(read)
    assert o.to_full_prob_forward() < -720.0 and abs(lf[0] - o.to_full_prob_forward()) < 1e-6
    oe, on = o.to_edge_and_init_freqs()
    assert np.all(np.isfinite(ef)) and np.all(np.isfinite(nf))
    assert np.max(np.abs(ef - oe)) < 1e-6 and np.max(np.abs(nf - on)) < 1e-6


def test_chimeric_read_takes_the_exact_path(gpu_lib, oracle):
    """A read whose halves come from different places: cells thousands of nats below the column maximum grow back
    after the junction.  The scaled linear kernels flush them; the driver's certificate (exact_dense.hip:
    certify_dense) sends the read through the log-domain recursion, the others stay on the fast path."""
    arrays, sg = small_dbg_model(500, 12, 0.01, seed=17, min_copy_num=1)
    chim = b"".join(D.sample_reads(arrays, 10 ** 9, 400, seed=9, max_reads=6))  # six reads' worth of bases from 6 places
    normal = D.sample_reads(arrays, 10 ** 9, 120, seed=3, max_reads=4)
    reads = [normal[0], chim, normal[1], normal[2]]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    rc = D.ReadCollection(reads)
    lf, lb, nf = gm.run_dense(rc)
    olf, olb, onf = om.run_dense_reads(reads, n_threads=8)
    assert np.max(np.abs(lf - olf)) < 1e-6 and np.max(np.abs(lb - olb)) < 1e-6
    assert np.max(np.abs(nf - onf)) < 1e-6
    lf2, _, _ = gm.run_dense(rc, False, False)  # forward only: the certificate fetches the backward maxima itself
    assert np.max(np.abs(lf2 - olf)) < 1e-6
    # transition posteriors (freq.rs:276-298) of the same batch
    lf3, ef, inf = gm.run_dense_edge_freqs(rc)
    oef, oinf = np.zeros(arrays.n_edges), np.zeros(arrays.n_nodes)
    for r in reads:
        e1, n1 = om.run(r).to_edge_and_init_freqs()
        oef += e1
        oinf += n1
    assert np.max(np.abs(lf3 - olf)) < 1e-6
    assert np.max(np.abs(ef - oef)) < 1e-6 and np.max(np.abs(inf - oinf)) < 1e-6
    # tables of the chimeric read alone (phmm_dense_tables)
    out = gm.run(chim)
    oo = om.run(chim)
    L = len(chim)
    for i in (0, L // 6 - 1, L // 6, L // 6 + 1, L // 2, L - 1):
        m, ins, d, sc = oo.forward.table(i)
        assert np.all(finite_close(out.forward.m[i], m, 1e-6)) and np.all(finite_close(out.forward.d[i], d, 1e-6))
        assert abs(out.forward.scal[i, 2] - sc[2]) < 1e-6
        m, ins, d, sc = oo.backward.table(i)
        assert np.all(finite_close(out.backward.m[i], m, 1e-6)) and np.all(finite_close(out.backward.i[i], ins, 1e-6))
        assert abs(out.backward.scal[i, 0] - sc[0]) < 1e-6
    assert np.max(np.abs(out.to_node_freqs() - oo.to_node_freqs())) < 1e-6


def test_q_score_exact(gpu_lib, oracle):
    """q_score_exact (q.rs:66-96): init = sum init_freq * ln init, trans = sum edge_freq * ln trans over emittable
    nodes / edges between emittable nodes, prior = 0.  The reference holds no KAT for it: the expected value is the
    formula evaluated on the ORACLE's transition posteriors."""
    arrays, _ = small_dbg_model(300, 12, 0.01, seed=5, min_copy_num=1)
    reads = D.sample_reads(arrays, 10 ** 9, 60, seed=3, max_reads=20)
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    _, ef, nf = gm.run_dense_edge_freqs(D.ReadCollection(reads))
    oef, onf = np.zeros(arrays.n_edges), np.zeros(arrays.n_nodes)
    for r in reads:
        e1, n1 = om.run(r).to_edge_and_init_freqs()
        oef += e1
        onf += n1
    emit = arrays.emission != ord("n")
    ok_e = emit[arrays.edge_src] & emit[arrays.edge_dst]
    want_init = float(np.sum(onf[emit] * arrays.init_logp[emit]))
    want_trans = float(np.sum(oef[ok_e] * arrays.trans_logp[ok_e]))
    qi, qt, qp = gm.q_score_exact(ef, nf)
    assert qp == 0.0
    assert abs(qi - want_init) < 1e-6 * max(1.0, abs(want_init))
    assert abs(qt - want_trans) < 1e-6 * max(1.0, abs(want_trans))
    assert qi < 0 and qt <= 0
    # a zero-probability node on the walk is the reference's assert (q.rs:79)
    a0, _ = small_dbg_model(300, 12, 0.01, seed=5, min_copy_num=0)
    if np.any(np.isinf(a0.init_logp[a0.emission != ord("n")])):
        with pytest.raises(D.PhmmError):
            D.PHMMModel(a0).q_score_exact(ef, nf)


@pytest.mark.parametrize("gaps", [0, 1, 2, 4, 5, 6])
def test_n_max_gaps_variants_match_oracle(gpu_lib, oracle, gaps):
    """n_max_gaps != 4: up to 4 the per-hop window with masked coefficients, above it merged closure entries
    (dense.hip); the frontier kernels loop over the Del levels.  Dense, adaptive and hinted scores vs the oracle."""
    arrays, _ = small_dbg_model(400, 12, 0.01, seed=31, min_copy_num=1)
    arrays.param = arrays.param.with_(n_max_gaps=gaps)
    reads = D.sample_reads(arrays, 10 ** 9, 90, seed=7, max_reads=9)
    reads = [r[: max(1, len(r) - (j * 7) % 31)] for j, r in enumerate(reads)]
    gm, om = D.PHMMModel(arrays), oracle.Model(arrays)
    rc = D.ReadCollection(reads)
    lf, lb, nf = gm.run_dense(rc)
    olf, olb, onf = om.run_dense_reads(reads, n_threads=8)
    assert np.max(np.abs(lf - olf)) < TOL_LOGP and np.max(np.abs(lb - olb)) < TOL_LOGP
    assert np.max(np.abs(nf - onf)) < TOL_FREQ * len(reads)
    _, lp = gm.to_full_prob_reads(rc, None, True)
    olp = om.full_prob_reads(reads, None, True, n_threads=8)
    assert np.max(np.abs(lp - olp)) < 1e-6
    mp, _ = gm.generate_mappings(rc, None, True)
    omp, _ = om.generate_mappings(reads, None, True, n_threads=8)
    _, lh = gm.to_full_prob_reads(rc, mp)
    olh = om.full_prob_reads(reads, omp, True, n_threads=8)
    assert np.max(np.abs(lh - olh)) < 1e-6
