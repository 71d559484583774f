"""CPU: the C-ABI library loads and exports every symbol include/phmm_amd.h declares, host
logic (graph -> PHMM builders) follows the reference, and the product path fails loudly
without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import dbgphmm_amd as D
from dbgphmm_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "phmm_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phmm_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _ffi.lib()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libphmm_amd.so does not export {n}"
    assert sorted(_ffi.DECLARED_SYMBOLS) == names
    assert lib.phmm_version().startswith(b"dbgphmm_amd")


def test_params_abi_matches_host():
    lib = _ffi.lib()
    from dbgphmm_amd.params import CPHMMParams
    import ctypes as C
    for p in (0.0, 0.001, 0.01, 0.1):
        c = CPHMMParams()
        assert lib.phmm_params_uniform(p, C.byref(c)) == 0
        h = D.PHMMParams.uniform(p).to_c()
        for name, _ in CPHMMParams._fields_:
            a, b = getattr(c, name), getattr(h, name)
            assert a == b or abs(a - b) < 1e-15, (p, name, a, b)
    c = CPHMMParams()
    assert lib.phmm_params_new(0.1, 0.1, 0.1, 1e-5, 400, 50, C.byref(c)) == _ffi.PHMM_EINVAL  # params.rs:83
    assert b"n_active_nodes" in lib.phmm_last_error()


@pytest.mark.skipif(_ffi.lib().phmm_device_count() > 0, reason="GPU present")
def test_product_path_fails_loudly_without_gpu():
    with pytest.raises(D.PhmmError) as e:
        D.PHMMModel(D.mock_linear().to_phmm(D.PHMMParams.default()))
    assert e.value.code == _ffi.PHMM_ENODEVICE


def test_workspace_and_read_info_entry_points_without_a_call():
    import ctypes as C
    lib = _ffi.lib()
    rc = D.ReadCollection([b"ACGT", b"GG"])
    cols, flags = np.zeros(2, np.uint16), np.zeros(2, np.uint32)
    # nothing has run on these reads yet
    assert lib.phmm_reads_last_call_info(rc._h, cols.ctypes.data_as(C.c_void_p), flags.ctypes.data_as(C.c_void_p)) == _ffi.PHMM_EINVAL
    assert lib.phmm_reads_last_call_info(None, None, None) == _ffi.PHMM_EINVAL
    if lib.phmm_device_count() == 0:
        assert lib.phmm_workspace_bytes() == 0
        assert lib.phmm_release_workspace() == _ffi.PHMM_ENODEVICE  # no CPU stand-in for the device pool either


def test_reads_validation_is_host_side():
    rc = D.ReadCollection([b"ACGT", b"GG"])
    assert len(rc) == 2 and rc.total_bases() == 6
    with pytest.raises(D.PhmmError):
        D.ReadCollection([b""])  # the reference panics on an empty read (table.rs:388)


def test_seq_graph_to_phmm_linear():
    # graph/mocks.rs:8-12 -> seq_graph.rs:160-223: init = ln 1 - ln 10, trans = ln 1
    a = D.mock_linear().to_phmm(D.PHMMParams.default())
    assert a.n_nodes == 10 and a.n_edges == 9
    assert bytes(a.emission) == b"ATTCGATCGT"
    assert np.allclose(a.init_logp, np.log(0.1)) and np.allclose(a.trans_logp, 0.0)


def test_toy_repeat_graph():
    # multi_dbg/toy.rs:260-303 through to_node_centric_graph (multi_dbg.rs:1551-1604)
    sg, k = D.toy_repeat()
    assert k == 4 and sg.base.shape[0] == 15 and sg.edge_src.shape[0] == 16
    edges = set(zip(sg.edge_src.tolist(), sg.edge_dst.tolist()))
    assert {(8, 9), (8, 6), (5, 9), (5, 6), (0, 1), (13, 14)} <= edges and (14, 0) not in edges
    a = sg.to_phmm(D.PHMMParams.default())
    # trans = cn(child) / sum of emittable child cn: node 5 (ccag) -> 6 (cagc, x3) or 9 (cagg, x1)
    t = {(int(s), int(d)): float(np.exp(w)) for s, d, w in zip(a.edge_src, a.edge_dst, a.trans_logp)}
    assert abs(t[(5, 6)] - 0.75) < 1e-15 and abs(t[(5, 9)] - 0.25) < 1e-15
    assert t[(11, 12)] == 0.0  # into a non-emittable 'n' node
    assert np.isneginf(a.init_logp[12]) and abs(np.exp(a.init_logp[6]) - 3 / 18) < 1e-15
    nz = sg.to_non_zero_phmm(D.PHMMParams.default())
    assert np.array_equal(nz.init_logp, a.init_logp)  # all copy numbers already >= 1


def test_dbg_from_haplotypes_and_vectorised_builder():
    hap = D.random_genome(300, 1)
    k = 12
    sg = D.dbg_from_haplotypes([hap, D.diverge(hap, 0.02, 2)], k)
    n = sg.base.shape[0]
    assert n >= 300 + k - 1
    assert (sg.base == ord("n")).sum() >= k - 1  # trailing padded k-mers are non-emittable
    # every emittable k-mer but the very first has a parent; unitig numbering is contiguous
    indeg = np.bincount(sg.edge_dst, minlength=n)
    assert (indeg == 0).sum() == 1
    assert np.mean(sg.edge_dst.astype(int) - sg.edge_src.astype(int) == 1) > 0.9
    p = D.PHMMParams.uniform(0.001)
    a, b = sg.to_phmm(p), D.vectorised_to_phmm(sg, p, 0)
    assert np.array_equal(a.init_logp, b.init_logp)
    assert np.allclose(np.exp(a.trans_logp), np.exp(b.trans_logp), atol=1e-15)
    # sum of init probs = 1; transition probs out of every node with an emittable child sum to 1
    assert abs(np.exp(a.init_logp).sum() - 1.0) < 1e-12
    out = np.zeros(n)
    np.add.at(out, a.edge_src, np.exp(a.trans_logp))
    assert np.all((np.abs(out - 1.0) < 1e-12) | (out == 0.0))


def test_sample_reads_follow_the_model():
    hap = D.random_genome(2000, 5)
    sg = D.dbg_from_haplotypes([hap], 16)
    a = D.vectorised_to_phmm(sg, D.PHMMParams.uniform(0.001), 0)
    reads = D.sample_reads(a, 20000, 200, seed=1)
    assert sum(map(len, reads)) >= 20000
    g = hap.tobytes()
    exact = sum(1 for r in reads if g.find(r) >= 0)
    assert exact > 0.4 * len(reads)  # 3 error kinds x p=0.001 x L=200: P(error-free) = e^-0.6
    assert all(set(r) <= set(b"ACGT") for r in reads)
