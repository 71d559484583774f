"""CPU: the oracle (oracle/phmm_oracle.c) against every known-answer value the reference's
own tests hold for this path (tests/golden/kat_hmmv2.json; tolerance 1e-5 as in the reference)
and against the reference's property tests."""
import json
import os

import numpy as np
import pytest

import dbgphmm_amd as D
from helpers import small_dbg_model

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_hmmv2.json")))
EPS = 1e-5


def lin(p):
    return D.mock_linear().to_phmm(p)


def test_prob_logadd(oracle):
    # prob.rs:181-197 incl. the -inf and equal branches
    assert oracle.logadd(-np.inf, -3.0) == -3.0
    assert oracle.logadd(-3.0, -np.inf) == -3.0
    assert oracle.logadd(-np.inf, -np.inf) == -np.inf
    assert oracle.logadd(-2.0, -2.0) == -2.0 + np.log(2.0)
    assert abs(oracle.logadd(np.log(0.3), np.log(0.2)) - np.log(0.5)) < 1e-15


def test_params_uniform():
    # params.rs:73-125
    p = D.PHMMParams.uniform(0.1)
    assert abs(np.exp(p.p_MM) - (1 - 0.2 - 1e-5)) < 1e-15
    assert abs(np.exp(p.p_IM) - (1 - 0.1 - 0.1 - 1e-5)) < 1e-15
    assert p.p_DM == p.p_IM and p.p_MI == p.p_MD == p.p_ID == p.p_DI == np.log(0.1)
    assert p.n_active_nodes == 40 and p.n_warmup == 50 and p.warmup_threshold == 200 and p.n_max_gaps == 4
    assert p.active_node_max_ratio == 30.0 and abs(np.exp(p.p_random) - 0.25) < 1e-16
    z = D.PHMMParams.zero_error()
    assert z.p_mismatch == -np.inf and z.p_match == 0.0
    with pytest.raises(AssertionError):
        D.PHMMParams.new(0.1, 0.1, 0.1, 1e-5, 400, 50)  # params.rs:83


def test_forward_zero_error(oracle):  # forward.rs:576-597
    om = oracle.Model(lin(D.PHMMParams.zero_error()))
    r = om.forward(b"CGATC")
    k = KAT["forward_zero_error"]
    assert abs(r.table(2)[0][5] - k["t2_m5"]) <= EPS
    assert abs(r.table(3)[0][6] - k["t3_m6"]) <= EPS
    assert abs(r.table(4)[0][7] - k["t4_m7"]) <= EPS
    assert abs(r.table(4)[3][2] - k["t4_e"]) <= EPS
    for i in range(5):
        m, ins, d, s = r.table(i)
        assert np.all(np.isneginf(ins)) and np.all(np.isneginf(d))
    assert np.isneginf(om.forward(b"CGATT").table(4)[3][2])


def test_forward_backward_high_error(oracle):  # forward.rs:599-619, backward.rs:607-628
    om = oracle.Model(lin(D.PHMMParams.high_error()))
    k = KAT["high_error"]
    r, r2 = om.forward(b"CGATC"), om.forward(b"CGATT")
    assert abs(r.table(4)[3][2] - k["fwd_CGATC_e"]) <= EPS
    assert abs(r.table(4)[0][7] - k["fwd_CGATC_t4_m7"]) <= EPS
    assert abs(r2.table(4)[3][2] - k["fwd_CGATT_e"]) <= EPS
    assert abs(r2.table(3)[3][2] - r.table(3)[3][2]) <= EPS
    b = om.backward(b"CGATC")
    assert abs(b.table(0)[0][2] - k["bwd_CGATC_t0_m2"]) <= EPS
    assert abs(b.table(0)[3][0] - k["bwd_CGATC_mb"]) <= EPS
    assert abs(om.backward(b"CGATT").table(0)[3][0] - k["bwd_CGATT_mb"]) <= EPS


def test_backward_zero_error(oracle):  # backward.rs:577-605
    om = oracle.Model(lin(D.PHMMParams.zero_error()))
    b = om.backward(b"CGATC")
    k = KAT["backward_zero_error"]
    assert abs(b.table(0)[3][0] - k["t0_mb"]) <= EPS
    for (t, n, v) in k["m"]:
        assert abs(b.table(t)[0][n] - v) <= EPS
    assert np.isneginf(om.backward(b"CGATT").table(0)[3][0])


def test_freq_kats(oracle):  # freq.rs:434-514
    om = oracle.Model(lin(D.PHMMParams.zero_error()))
    o = om.run(b"CGATC")
    assert abs(o.to_full_prob_forward() - o.to_full_prob_backward()) < 1e-7
    for i in range(5):
        assert abs(np.exp(o.to_emit_probs(i + 1)[0][3 + i]) - 1.0) < EPS
    assert abs(np.exp(o.to_emit_probs(0)[3][0]) - 1.0) < EPS
    assert np.allclose(o.to_node_freqs(), KAT["node_freq_zero_error"]["freq"], atol=EPS)
    om = oracle.Model(lin(D.PHMMParams.default()))
    nf = om.run(b"CGATC").to_node_freqs()
    assert np.all(nf[[0, 1, 2, 8, 9]] < 0.01) and np.all(nf[3:8] > 0.98)
    o = om.run(b"ATTCGTCGT")
    assert abs(o.to_full_prob_forward() - o.to_full_prob_backward()) < 1e-5
    assert np.allclose(o.to_node_freqs(), 1.0, atol=0.01)


def test_trans_probs_kats(oracle):  # freq.rs:517-609
    om = oracle.Model(lin(D.PHMMParams.zero_error()))
    o = om.run(b"CGATC")
    tp0, _ = o.to_trans_and_init_probs(0)
    assert np.all(np.isneginf(tp0))
    for i in range(1, 5):
        tp, _ = o.to_trans_and_init_probs(i)
        assert abs(np.exp(tp[2 + i][0]) - 1.0) < EPS  # mm on edge e_{2+i}
    assert np.all(np.isneginf(o.to_trans_and_init_probs(5)[0]))
    ef, _ = o.to_edge_and_init_freqs()
    assert np.all(ef[[0, 1, 2, 7, 8]] < 1e-4) and np.all(ef[3:7] > 0.9999)
    om = oracle.Model(lin(D.PHMMParams.default()))
    o = om.run(b"ATTCGTCGT")
    tps = [o.to_trans_and_init_probs(i)[0] for i in range(1, 10)]
    for (i, e, col) in [(0, 0, 0), (1, 1, 0), (2, 2, 0), (3, 3, 0), (4, 4, 3), (4, 5, 2), (5, 6, 0), (6, 7, 0), (7, 8, 0)]:
        assert np.exp(tps[i][e][col]) > 0.9
    ef, _ = o.to_edge_and_init_freqs()
    assert np.allclose(ef, 0.99, atol=0.01)


def test_hint_top3(oracle):  # forward.rs:640-669, backward.rs:630-651
    om = oracle.Model(lin(D.PHMMParams.high_error()))
    o = om.run(b"CGATC")
    hint = o.to_mapping(3)
    assert [hint.nodes(i) for i in range(5)] == KAT["hint_top3_high_error"]["nodes"]
    p1 = om.forward(b"CGATC").full_prob()
    p2 = om.forward(b"CGATC", oracle.FWD_MAPPING, hint).full_prob()
    assert abs(p1 - p2) < 0.1
    h5 = o.to_mapping(5)
    assert abs(om.backward(b"CGATC").full_prob() - om.backward(b"CGATC", oracle.BWD_MAPPING, h5).full_prob()) < 0.1


def test_crossing(oracle):  # common.rs:381-417, seq_graph.rs:440-503
    rb = KAT["crossing"]["read"].encode()
    a1 = D.mock_crossing(False).to_phmm(D.PHMMParams.default())
    a2 = D.mock_crossing(True).to_phmm(D.PHMMParams.default())
    assert a1.n_nodes == 40 and a1.n_edges == 40
    assert np.allclose(np.exp(a1.trans_logp[36:40]), 0.5)
    assert np.allclose(np.exp(a2.trans_logp[36:40]), [1, 0, 0, 1])
    o1, o2 = oracle.Model(a1).run(rb), oracle.Model(a2).run(rb)
    assert o1.to_full_prob_forward() > KAT["crossing"]["logp_without_edge_copy_num_gt"]
    assert o2.to_full_prob_forward() < KAT["crossing"]["logp_with_edge_copy_num_lt"]
    assert abs(o1.to_full_prob_forward() - o1.to_full_prob_backward()) < 0.1
    ef, _ = o1.to_edge_and_init_freqs()
    assert ef[36] < 1e-4 and ef[37] > 0.9 and ef[38] < 1e-4 and ef[39] < 1e-4
    ef2, _ = o2.to_edge_and_init_freqs()
    assert ef2[37] == 0.0 and ef2[38] == 0.0


def test_toy_repeat_hints(oracle):  # multi_dbg/posterior/test.rs:544-576
    sg, k = D.toy_repeat()
    t = KAT["toy_repeat_hints"]
    om = oracle.Model(sg.to_non_zero_phmm(D.PHMMParams.uniform(t["p"]).with_(n_warmup=k)))
    for read, best in t["reads"].items():
        (po, nd, lp), _ = om.generate_mappings([read.encode()], None, True)
        assert [int(nd[po[i]]) for i in range(len(read))] == best


def test_dense_vs_sparse_property(oracle):
    """forward.rs:621-638 / tests/hmm.rs:59-71: dense vs sparse tables differ < 1e-9 in
    probability; hmmv2/tests/dbg.rs:44-45: totals agree within 1e-4, fwd vs bwd within 0.01."""
    arrays, _ = small_dbg_model(300, 12, 0.001, seed=4)
    om = oracle.Model(arrays)
    reads = D.sample_reads(arrays, 10 ** 9, 60, seed=2, max_reads=6)
    for r in reads:
        f1, f2 = om.forward(r), om.forward(r, oracle.FWD_SPARSE_TOPK)
        for i in range(len(r)):
            a, b = f1.table(i), f2.table(i)
            diff = sum(np.abs(np.exp(a[q]) - np.exp(b[q])).sum() for q in range(3))
            assert diff < 1e-9
    (mp, nf) = om.generate_mappings(reads, None, True)
    p0 = om.full_prob_reads(reads, None, True).sum()
    p1 = om.full_prob_reads(reads, mp, True).sum()
    lf, lb, _ = om.run_dense_reads(reads)
    assert abs(p0 - lf.sum()) < 1e-4 and abs(p1 - lf.sum()) < 1e-4
    assert abs(lf.sum() - lb.sum()) < 0.01 * len(reads)


def _map_nodes_case(case):
    k = KAT["map_nodes"]
    po = np.array([0, 2, 4], dtype=np.uint64)
    nd = np.array(sum(k["nodes"], []), dtype=np.uint32)
    lp = np.log(np.array(sum(k["probs"], [])))
    n_old = 5
    if case == "case1":  # v -> [v+1]
        mo, mn = np.arange(n_old + 1), np.arange(1, n_old + 1)
    else:                # v -> [v, v+1]
        mo, mn = np.arange(n_old + 1) * 2, np.stack([np.arange(n_old), np.arange(1, n_old + 1)], axis=1).reshape(-1)
    return (po, nd, lp), mo, mn, k[case]


@pytest.mark.parametrize("case", ["case1", "case2"])
def test_map_nodes_kat(oracle, case):  # hint.rs:233-263
    mp, mo, mn, exp = _map_nodes_case(case)
    po, nd, lp = oracle.map_nodes(mp, mo, mn)
    assert po.tolist() == [0, len(exp["nodes"][0]), len(exp["nodes"][0]) + len(exp["nodes"][1])]
    assert nd.tolist() == sum(exp["nodes"], [])
    if case == "case1":
        assert np.array_equal(lp, mp[2])  # the reference asserts Mapping equality (hint.rs:262)
    assert np.max(np.abs(np.exp(lp) - np.array(sum(exp["probs"], [])))) < 1e-15
