"""CPU side of the tandem-repeat coverage (no GPU): the dataset generators restated from the reference, the k -> k+1
node map, and the reference's OWN assertions on these datasets (hmmv2/tests/dbg.rs:44-45, 85-114, 195-238) evaluated
on the oracle -- so that what tests/test_gpu_repeats.py holds the HIP path to is itself checked against the
reference's stated properties."""
import numpy as np

import dbgphmm_amd as D
from repeat_cases import GENOMES, dataset


def test_tandem_repeat_generator_shape():
    """genome.rs:294-340: shared unique ends, a repeat of n_unit units, exactly round(len * rate) edit operations
    per mutation pass (random_seq.rs:163-181)."""
    unit, n_unit, _, h0, _, end, n_hap, h, _ = GENOMES["u20"]
    haps = D.tandem_repeat_polyploid_with_unique_homo_ends(*GENOMES["u20"])
    assert len(haps) == n_hap
    a, b = haps
    assert np.array_equal(a[:end], b[:end]) and np.array_equal(a[-end:], b[-end:])  # homozygous ends
    rep = a[end:-end]
    assert rep.shape[0] == unit * n_unit and np.array_equal(rep[:unit], rep[unit:2 * unit])  # H0 = 0: a perfect repeat
    # hap b: 10 operations on the 1000-base repeat; its length moves by (#ins - #del)
    assert abs(len(b) - len(a)) <= 10 and not np.array_equal(a, b)
    # deterministic in its seeds, different for another div_seed
    again = D.tandem_repeat_polyploid_with_unique_homo_ends(*GENOMES["u20"])
    assert all(np.array_equal(x, y) for x, y in zip(haps, again))
    other = D.tandem_repeat_polyploid_with_unique_homo_ends(*GENOMES["u20"][:-1], 7)
    assert not np.array_equal(other[1], haps[1])
    # mutate_exact: exactly that many operations (a substitution-only profile cannot be asked for: count edits)
    rng = np.random.default_rng(0)
    s = D.random_genome(2000, 5)
    t = D.mutate_exact(s, 0.01, rng)
    assert abs(len(t) - len(s)) <= 20 and not np.array_equal(s, t)
    assert np.array_equal(D.mutate_exact(s, 0.0, rng), s)


def test_repeat_graph_collapses_the_units():
    """k = 40 > unit = 20: the repeat's k-mers collapse onto one cycle of 20 nodes with copy number ~ 2 x 50."""
    arrays, reads, sg, haps = dataset("u20", 40)
    assert arrays.n_nodes < 700 and sg.copy_num.max() >= 60
    assert sum(len(r) for r in reads) >= 20 * sum(len(h) for h in haps)  # 20x
    assert max(len(r) for r in reads) <= 1000 and all(set(r) <= set(b"ACGT") for r in reads)


def test_kp1_node_map_is_hint_kp1_from_hint_k():
    """multi_dbg.rs:1325-1335: a k-mer goes to the (k+1)-mers that end with it; every (k+1)-mer whose last k bases
    are a node of the k graph is the image of exactly that node."""
    h = D.random_genome(1500, 3)
    haps = [h, D.diverge(h, 0.02, 4)]
    sg1, off, nodes, sg = D.kp1_node_map(haps, 12)
    n_k, n_k1 = sg.base.shape[0], sg1.base.shape[0]
    assert off.shape[0] == n_k + 1 and nodes.max() < n_k1
    src = np.repeat(np.arange(n_k), np.diff(off.astype(np.int64)))
    assert np.array_equal(sg1.base[nodes], sg.base[src])  # same last base
    assert np.unique(nodes).shape[0] == nodes.shape[0]    # a (k+1)-mer has one suffix k-mer
    # nearly every (k+1)-mer is an image (all but the trailing 'Xn..n' one per haplotype end)
    assert n_k1 - nodes.shape[0] <= 2 * len(haps)


def test_reference_properties_hold_on_the_oracle(oracle):
    """dbg.rs:85-114 (with / without mapping: read-set total within 1e-4) and dbg.rs:195-238 (fixed warm-up with
    n_active_nodes = 200 against the adaptive run per read within 1e-4; forward against backward of the adaptive run
    within 0.01) on u100 and u20 at k = 40, evaluated on the oracle."""
    for name in ("u100", "u20"):
        arrays, reads, sg, haps = dataset(name, 40)
        reads = reads[:24]
        om = oracle.Model(arrays)
        mp, _ = om.generate_mappings(reads, None, True, n_threads=8)
        p0 = om.full_prob_reads(reads, None, True, n_threads=8)
        p1 = om.full_prob_reads(reads, mp, True, n_threads=8)
        assert abs(p0.sum() - p1.sum()) < 1e-4, name
        o200 = oracle.Model(D.vectorised_to_phmm(sg, arrays.param.with_(n_active_nodes=200), 1))
        pf = o200.full_prob_reads(reads, None, False, n_threads=8)
        assert np.max(np.abs(pf - p0)) < 1e-4, name
        for r in reads[:6]:
            out = om.run_sparse_adaptive(r, True)
            assert abs(out.to_full_prob_forward() - out.to_full_prob_backward()) < 0.01, name
