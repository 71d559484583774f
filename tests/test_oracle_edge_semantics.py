"""CPU: two behaviours of the reference's recursion that the GPU path has to reproduce (oracle only, no GPU)."""
import ctypes as C
import math

import numpy as np

import dbgphmm_amd as D
from helpers import compare_mappings_tie_aware, small_dbg_model


def test_cut_read_keeps_the_begin_chain(oracle):
    """forward_with_mapping on a model in which a k-mer of the read's path has copy number 0 (a candidate of `infer`,
    posterior.rs:483-515): every node of the read's lists dies at the cut, but ln P stays finite -- the InsBegin chain
    (fib, forward.rs:541-545: p_random p_II per base) re-enters the graph behind the cut through from_begin
    (forward.rs:337-359).  The price is about ln(p_random p_II) per base of the prefix before the cut."""
    arrays, sg = small_dbg_model(900, 12, 0.003, seed=21)
    reads = [r for r in D.sample_reads(arrays, 10 ** 9, 420, seed=22, max_reads=40) if len(r) > 380][:6]
    om = oracle.Model(arrays)
    (po, nd, lp), _ = om.generate_mappings(reads, None, True, n_threads=4)
    healthy = om.full_prob_reads(reads, (po, nd, lp), True, n_threads=4)
    cut_at = 300
    victim = int(nd[int(po[cut_at])])  # best node of base 300 of read 0
    cn = sg.copy_num.copy()
    cn[victim] = 0
    with np.errstate(divide="ignore"):
        a1 = D.vectorised_to_phmm(D.SeqGraph(cn, sg.base, sg.edge_src, sg.edge_dst, None), arrays.param, 0)
    cut = oracle.Model(a1).full_prob_reads(reads, (po, nd, lp), True, n_threads=4)
    assert np.all(np.isfinite(cut))
    per_base = arrays.param.p_random + arrays.param.p_II  # log values
    # read 0 pays for ~300 bases of InsBegin (a little more: it re-enters a few bases behind the cut)
    assert healthy[0] > -60.0
    assert cut_at * per_base - 80.0 < cut[0] < cut_at * per_base + 20.0, (cut[0], cut_at * per_base)
    # reads that do not cross the k-mer keep their score (up to the renormalised initial probabilities)
    away = np.flatnonzero(np.abs(cut - healthy) < 1.0)
    assert away.size >= 1


def test_tie_rule_hook_only_moves_ties(oracle):
    """orc_set_tie_rule (oracle test hook): another order of near-equal values in the top-k cuts leaves every score
    within 1e-6 and every list equal under the tie-aware comparison the GPU tests use."""
    arrays, sg = small_dbg_model(700, 10, 0.01, seed=5)
    reads = D.sample_reads(arrays, 10 ** 9, 150, seed=6, max_reads=12)
    om = oracle.Model(arrays)
    L = oracle.lib()
    L.orc_set_tie_rule.argtypes = [C.c_double, C.c_int]
    base_map, _ = om.generate_mappings(reads, None, True, n_threads=4)
    base_lp = om.full_prob_reads(reads, None, True, n_threads=4)
    try:
        L.orc_set_tie_rule(1e-9, 1)
        alt_map, _ = om.generate_mappings(reads, None, True, n_threads=4)
        alt_lp = om.full_prob_reads(reads, None, True, n_threads=4)
    finally:
        L.orc_set_tie_rule(0.0, 0)
    assert np.max(np.abs(alt_lp - base_lp)) < 1e-6
    # (the alternative order plays the part of the GPU's lists)
    retried, overflow = compare_mappings_tie_aware(oracle, om, reads, alt_map, base_map)
    assert retried + overflow <= len(reads)  # (it passes: reads in the 400-slot overflow regime are counted, not compared)
