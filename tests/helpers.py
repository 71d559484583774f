"""Shared synthetic-input builders for the tests (seeded, small)."""
import numpy as np

from dbgphmm_amd import PHMMParams, dbg_from_haplotypes, diverge, random_genome, sample_reads, vectorised_to_phmm


def small_dbg_model(genome_len=300, k=12, p=0.01, seed=7, diploid=True, min_copy_num=0):
    hap_a = random_genome(genome_len, seed)
    haps = [hap_a, diverge(hap_a, 0.02, seed + 1)] if diploid else [hap_a]
    sg = dbg_from_haplotypes(haps, k)
    param = PHMMParams.uniform(p).with_(n_warmup=k)
    return vectorised_to_phmm(sg, param, min_copy_num), sg


def finite_close(a, b, atol, floor=None):
    """compare log arrays: both -inf, or |a-b| <= atol; entries where the reference value is
    below `floor` may be -inf on the GPU (underflow of the scaled linear domain)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    with np.errstate(invalid="ignore"):
        return _finite_close(a, b, atol, floor)


def _finite_close(a, b, atol, floor):
    both_inf = np.isneginf(a) & np.isneginf(b)
    ok = both_inf | (np.abs(a - b) <= atol)
    if floor is not None:
        ok |= (b < floor) & (np.isneginf(a) | (np.abs(a - b) <= 1e-3))
    return ok


def compare_mappings(reads, gpu_arrays, orc_arrays, min_logp=-20.0, tol=1e-6, top_k=0, ratio=30.0, edge=1e-6, deep=True):
    """Mapping lists of the GPU against the oracle's, position by position (CSR triples over the same reads).
    Entries above e^min_logp agree in order, node and value (`tol`).  In ratio mode every entry that is not within
    `edge` of the cut (best - ratio; rounding may put such an entry on either side) must be on BOTH lists with the
    same value -- so the list lengths agree up to entries at the cut (deep = False leaves that out: where the 400-slot
    vector overflows, the unpinned regime of DESIGN.md section 2, the two restatements drop different tails).  Equal probabilities may swap places (the two
    haplotype copies of a k-mer; sparsevec's tie order is unpinned)."""
    gpo, gnd, glp = gpu_arrays
    opo, ond, olp = orc_arrays
    assert gpo.shape == opo.shape
    g = 0
    for r in reads:
        for i in range(len(r)):
            a0, a1 = int(gpo[g]), int(gpo[g + 1])
            b0, b1 = int(opo[g]), int(opo[g + 1])
            gn, gl = gnd[a0:a1], glp[a0:a1]
            on, ol = ond[b0:b1], olp[b0:b1]
            # entries that matter (prob > e^min_logp) must agree in order, node and value
            ka, kb = int((gl > min_logp).sum()), int((ol > min_logp).sum())
            assert ka == kb, (i, gn, gl, on, ol)
            # (nodes with equal probability -- e.g. the two haplotype copies of a k-mer -- may swap)
            assert np.max(np.abs(gl[:ka] - ol[:kb]), initial=0.0) < tol, (i, gl, ol)
            capped = a1 - a0 == 400 or b1 - b0 == 400 or (top_k and a1 - a0 == top_k)  # (a full fixed-size list too)
            if capped:
                # list capped at MAX_ACTIVE_NODES: which of the nodes that TIE with the 400th value are
                # kept is arbitrary (sparsevec tie order is unpinned); compare strictly above the cut
                # (a fixed-size list goes on below any ratio: values are compared down to e^-30 of the best)
                ka = kb = int((gl > max(gl[-1] + 1e-9, gl[0] - 30.0)).sum())
            assert sorted(gn[:ka].tolist()) == sorted(on[:kb].tolist()), (i, gn, on)
            od = dict(zip(on[:kb].tolist(), ol[:kb].tolist()))
            assert all(abs(od[n] - l) < tol for n, l in zip(gn[:ka].tolist(), gl[:ka].tolist())), (i, gn, gl, on, ol)
            # lists are sorted descending and respect the ratio / the fixed size
            assert np.all(np.diff(gl[np.isfinite(gl)]) <= 1e-12)
            if top_k:
                assert a1 - a0 == b1 - b0 <= top_k
            elif a1 > a0:
                assert gl[0] - gl[-1] < ratio + 1e-9
                if deep and not capped and b1 > b0 and np.isfinite(gl[0]):
                    # the whole list, down to the cut: same members, same values
                    cut = max(gl[0], ol[0]) - ratio + edge
                    ga = {int(n): float(l) for n, l in zip(gn, gl) if l > cut}
                    oa = {int(n): float(l) for n, l in zip(on, ol) if l > cut}
                    assert set(ga) == set(oa), (i, sorted(set(ga) ^ set(oa)), gl, ol)
                    assert all(abs(ga[n] - oa[n]) < max(tol, 1e-5) for n in ga), (i, gl, ol)
                    assert abs((a1 - a0) - (b1 - b0)) <= sum(1 for l in list(gl) + list(ol) if l <= cut), (i, gl, ol)
            g += 1


def same_mappings(a, b, near=1e-9):
    """Two mapping CSR triples made by the GPU path for the same reads (another grouping of the reads, another model
    handle, another chunking): offsets and values bit-equal; nodes equal except that entries of one list whose values
    are within `near` of each other (the two haplotype copies of a k-mer: their linear values may differ in the last
    bit, and which of the two gets the larger one depends on the order of additions) may trade places."""
    po1, nd1, lp1 = a
    po2, nd2, lp2 = b
    if not (np.array_equal(po1, po2) and np.array_equal(lp1, lp2)):
        return False
    bad = np.flatnonzero(nd1 != nd2)
    if bad.size == 0:
        return True
    # the differing entries of each list must be a permutation among near-equal values
    owner = np.searchsorted(po1, bad, side="right") - 1
    for p in np.unique(owner):
        ix = bad[owner == p]
        if sorted(nd1[ix].tolist()) != sorted(nd2[ix].tolist()):
            return False
        # every swapped entry sits in a run of near-equal values that covers its partner
        for lo in ix:
            partner = ix[nd2[ix] == nd1[lo]]
            if partner.size != 1 or abs(float(lp1[lo]) - float(lp1[partner[0]])) > near:
                return False
    return True


def subset_csr(offsets, arrays, idx):
    """CSR triple of the reads `idx` cut out of a mapping triple over all reads (offsets: base offsets of the reads)."""
    po, nd, lp = arrays
    out_off, nodes, logp = [0], [], []
    for r in idx:
        p0, p1 = int(offsets[r]), int(offsets[r + 1])
        e0, e1 = int(po[p0]), int(po[p1])
        nodes.append(nd[e0:e1])
        logp.append(lp[e0:e1])
        out_off.extend((po[p0 + 1:p1 + 1].astype(np.int64) - e0 + out_off[-1]).tolist())
    return (np.array(out_off, dtype=np.uint64), np.concatenate(nodes) if nodes else np.zeros(0, np.uint32),
            np.concatenate(logp) if logp else np.zeros(0))


# (bucket width, mode): mode & 1 reverses the order inside a bucket, mode & 2 shifts the buckets by half a width
TIE_RULES = ((1e-9, 0), (1e-9, 1), (1e-9, 2), (1e-9, 3), (1e-6, 0), (1e-6, 1), (1e-6, 2), (1e-6, 3), (0.0, 1))


def compare_mappings_tie_aware(oracle_mod, om, reads, gpu_arrays, orc_arrays, use_max_ratio=True, **kw):
    """compare_mappings read by read; a read that fails is run again through the oracle with its value sorts
    breaking near-ties the other ways (orc_set_tie_rule: values in one bucket of width 1e-9 / 1e-6 count as tied,
    in iteration order or reversed) and passes when the GPU lists equal the oracle's under one of them.  The
    reference's top-k cuts (table.rs:117-149) sort log values whose last bits are rounding noise, so which of two
    near-equal nodes stays is not a property of the algorithm; what the cut leaves out then moves deep entries
    (below ~1e-8 of the best) by up to a gap probability.  -> (reads that needed another tie order, reads in the overflow regime)."""
    import ctypes as C
    L = oracle_mod.lib()
    L.orc_set_tie_rule.argtypes = [C.c_double, C.c_int]
    off = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    retried = overflow = 0
    for ri, r in enumerate(reads):
        g1 = subset_csr(off, gpu_arrays, [ri])
        try:
            compare_mappings([r], g1, subset_csr(off, orc_arrays, [ri]), **kw)
            continue
        except AssertionError as e:
            first = e
        retried += 1
        for eps, rev in TIE_RULES:
            L.orc_set_tie_rule(eps, rev)
            try:
                o1, _ = om.generate_mappings([r], None, use_max_ratio, n_threads=1)
                compare_mappings([r], g1, o1, **kw)
                break
            except AssertionError:
                continue
            finally:
                L.orc_set_tie_rule(0.0, 0)
        else:
            # a read that fills a 400-slot sparse vector of the reference (the overflow regime of DESIGN.md section 2:
            # the two restatements do not drop the same inserts) is outside the parity domain
            f = om.forward(r, oracle_mod.FWD_SPARSE_RATIO if use_max_ratio else oracle_mod.FWD_SPARSE_TOPK)
            if max([len(f.nodes(i, w)) for i in range(len(r)) if not f.is_dense(i) for w in (0, 1, 2)], default=0) < 400:
                raise first
            retried -= 1
            overflow += 1
    return retried, overflow


def scores_tie_aware(oracle_mod, got, want, recompute, tol=1e-6):
    """Per-read scores `got` against the oracle's `want`; a read off by more than tol is recomputed by
    `recompute(read_index)` under the other tie rules (see compare_mappings_tie_aware) and must match one of them.
    -> number of reads that needed another tie order."""
    import ctypes as C
    L = oracle_mod.lib()
    L.orc_set_tie_rule.argtypes = [C.c_double, C.c_int]
    bad = np.flatnonzero(~((np.abs(got - want) < tol) | ((got == want))))
    for b in bad:
        seen = [float(want[b])]
        for eps, rev in TIE_RULES:
            L.orc_set_tie_rule(eps, rev)
            try:
                v = recompute(int(b))
            finally:
                L.orc_set_tie_rule(0.0, 0)
            seen.append(float(v))
            if abs(v - got[b]) < tol or v == got[b]:
                break
        else:
            raise AssertionError((int(b), float(got[b]), seen))
    return int(bad.size)


def wide_class_vs_generic(arrays, reads):
    """generate_mappings with the 448-thread 400-slot kernels (wide_fwd_kernel.h / wide_bwd_kernel.h) and with the
    one-wave generic ones (PHMM_NO_WIDE_CLASS=1: the kernels every parity test of rounds 1-2 ran) on the same reads.
    The two are the same algorithm statement by statement; what may differ is the last bit of a sum (fused multiply-adds
    are placed per kernel).  -> dict: max |d ln P|, positions whose lists differ as SETS, max |d list ln p| over the
    rest, max |d node usage|, flags of the call, seconds of the second call of each."""
    import os
    import time
    import dbgphmm_amd as D

    def run(no_wide):
        os.environ["PHMM_NO_WIDE_CLASS"] = "1" if no_wide else "0"
        try:
            gm, rc = D.PHMMModel(arrays), D.ReadCollection(reads)
            gm.generate_mappings(rc, None, True)
            t0 = time.perf_counter()
            mp, nf = gm.generate_mappings(rc, None, True)
            dt = time.perf_counter() - t0
            _, flags = rc.last_call_info()
            return mp.read_logp()[1].copy(), [x.copy() for x in mp.arrays()], nf.copy(), flags.copy(), dt
        finally:
            os.environ.pop("PHMM_NO_WIDE_CLASS", None)

    a, b = run(True), run(False)
    (pa, na, la), (pb, nb, lb) = a[1], b[1]
    bad, dmax = 0, 0.0
    if not np.array_equal(pa, pb):
        bad = int((np.diff(pa.astype(np.int64)) != np.diff(pb.astype(np.int64))).sum()) if pa.shape == pb.shape else len(pa)
    elif np.array_equal(na, nb):
        dmax = float(np.max(np.abs(la - lb), initial=0.0))
    else:
        for i in np.unique(np.searchsorted(pa, np.flatnonzero(na != nb), side="right") - 1):
            s0, s1 = int(pa[i]), int(pa[i + 1])
            if sorted(na[s0:s1].tolist()) != sorted(nb[s0:s1].tolist()):
                bad += 1
            else:
                dmax = max(dmax, float(np.max(np.abs(np.sort(la[s0:s1]) - np.sort(lb[s0:s1])))))
    return dict(d_logp=float(np.max(np.abs(a[0] - b[0]), initial=0.0)), list_positions_differing=bad, positions=len(pa) - 1,
                d_list_logp=dmax, d_node_freq=float(np.max(np.abs(a[2] - b[2]), initial=0.0)), flags=a[3],
                generic_s=a[4], wide_s=b[4])
