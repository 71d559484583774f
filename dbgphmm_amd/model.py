"""Host-side mirror of the reference's `PHMMModel` call surface over the C ABI.

Method names, argument meaning and error behaviour follow the reference
(`impl PHMMModel`, /root/reference/src/hmmv2/{forward,backward,freq,hint}.rs) so that the
parity tests read like the reference's own tests.  All compute happens in
libphmm_amd.so (HIP, gfx950); this module only marshals arrays.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from .graph import PHMMArrays
from .params import PHMMParams


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    # torch tensor (device or host output buffer)
    return C.c_void_p(a.data_ptr())


class ReadCollection:
    """ReadCollection<S> (src/common/collection.rs:38-83): concatenated bases + offsets."""

    def __init__(self, reads: Sequence[bytes]):
        self.reads = [bytes(r) for r in reads]
        self.offsets = np.zeros(len(self.reads) + 1, dtype=np.uint64)
        self.offsets[1:] = np.cumsum([len(r) for r in self.reads])
        self.bases = np.frombuffer(b"".join(self.reads), dtype=np.uint8)
        if self.bases.shape[0] == 0:
            self.bases = np.zeros(1, dtype=np.uint8)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().phmm_reads_create(_ptr(self.bases), _ptr(self.offsets), len(self.reads), C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.lib().phmm_reads_destroy(self._h)
            self._h = None

    def __len__(self) -> int:
        return len(self.reads)

    def total_bases(self) -> int:
        return int(self.offsets[-1])

    def last_call_info(self) -> Tuple[np.ndarray, np.ndarray]:
        """(dense warm-up columns[R], PHMM_READ_* flags[R]) of the most recent adaptive-sparse call."""
        cols = np.empty(len(self.reads), dtype=np.uint16)
        flags = np.empty(len(self.reads), dtype=np.uint32)
        _ffi.check(_ffi.lib().phmm_reads_last_call_info(self._h, _ptr(cols), _ptr(flags)))
        return cols, flags


class Mappings:
    """Mappings (src/hmmv2/hint.rs:150-152) over a ReadCollection: 3-level CSR."""

    def __init__(self, handle, reads: ReadCollection):
        self._h = handle
        self.reads = reads
        self._cache = None

    @staticmethod
    def from_arrays(reads: ReadCollection, pos_off: np.ndarray, nodes: np.ndarray,
                    logp: Optional[np.ndarray] = None) -> "Mappings":
        po = np.ascontiguousarray(pos_off, dtype=np.uint64)
        nd = np.ascontiguousarray(nodes, dtype=np.uint32)
        lp = None if logp is None else np.ascontiguousarray(logp, dtype=np.float64)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().phmm_mappings_create(reads._h, _ptr(po), _ptr(nd), _ptr(lp), C.byref(h)))
        return Mappings(h, reads)

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.lib().phmm_mappings_destroy(self._h)
            self._h = None

    def arrays(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        if self._cache is None:
            L = _ffi.lib()
            tp, te = L.phmm_mappings_total_positions(self._h), L.phmm_mappings_total_entries(self._h)
            po = np.empty(tp + 1, dtype=np.uint64)
            nd = np.empty(max(te, 1), dtype=np.uint32)
            lp = np.empty(max(te, 1), dtype=np.float64)
            _ffi.check(L.phmm_mappings_export(self._h, _ptr(po), _ptr(nd), _ptr(lp)))
            self._cache = (po, nd[:te], lp[:te])
        return self._cache

    def nodes(self, read: int, pos: int) -> List[int]:
        po, nd, _ = self.arrays()
        g = int(self.reads.offsets[read]) + pos
        return nd[int(po[g]):int(po[g + 1])].tolist()

    def probs(self, read: int, pos: int) -> np.ndarray:
        po, _, lp = self.arrays()
        g = int(self.reads.offsets[read]) + pos
        return lp[int(po[g]):int(po[g + 1])]

    def read_logp(self, out_logp=None):
        """ln P(read) of the forward pass that produced these mappings -> (total, per read)."""
        lp = np.empty(len(self.reads)) if out_logp is None else out_logp
        tot = np.empty(1)
        _ffi.check(_ffi.lib().phmm_mappings_read_logp(self._h, _ptr(lp), _ptr(tot)))
        return float(tot[0]), lp

    def map_nodes(self, model_after: "PHMMModel", map_off: np.ndarray, map_nodes: np.ndarray) -> "Mappings":
        """Mapping::map_nodes (hint.rs:60-88) for every read: carry the lists over to another graph through a
        node map given as CSR over this mapping's graph (MultiDbg::hint_kp1_from_hint_k,
        PurgeEdgeMap::update_mapping)."""
        mo = np.ascontiguousarray(map_off, dtype=np.uint32)
        mn = np.ascontiguousarray(map_nodes, dtype=np.uint32)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().phmm_mappings_map_nodes(model_after._h, self.reads._h, self._h, _ptr(mo), _ptr(mn),
                                                      mo.shape[0] - 1, C.byref(h)))
        return Mappings(h, self.reads)

    def to_node_freqs(self, n_nodes: int) -> np.ndarray:
        """Mappings::to_node_freqs (hint.rs:161-171)"""
        out = np.empty(n_nodes)
        _ffi.check(_ffi.lib().phmm_mappings_node_freqs(self._h, n_nodes, _ptr(out)))
        return out


class DenseTables:
    """PHMMTables of one read (natural-log values), tables[i] <-> x[i] (src/hmmv2.rs:7-29)."""

    def __init__(self, m, i, d, scal):
        self.m, self.i, self.d, self.scal = m, i, d, scal

    def __len__(self):
        return self.m.shape[0]

    @property
    def mb(self):
        return self.scal[:, 0]

    @property
    def ib(self):
        return self.scal[:, 1]

    @property
    def e(self):
        return self.scal[:, 2]


class PHMMModel:
    """PHMMModel<N, E> (src/hmmv2/common.rs:61-64) resident on one MI355X."""

    def __init__(self, arrays: PHMMArrays):
        self.arrays = arrays
        self.param = arrays.param
        self._cp = arrays.param.to_c()
        em = np.ascontiguousarray(arrays.emission, dtype=np.uint8)
        init = np.ascontiguousarray(arrays.init_logp, dtype=np.float64)
        src = np.ascontiguousarray(arrays.edge_src, dtype=np.uint32)
        dst = np.ascontiguousarray(arrays.edge_dst, dtype=np.uint32)
        tr = np.ascontiguousarray(arrays.trans_logp, dtype=np.float64)
        self.n_nodes, self.n_edges = em.shape[0], src.shape[0]
        h = C.c_void_p()
        _ffi.check(_ffi.lib().phmm_model_create(self.n_nodes, self.n_edges, _ptr(em), _ptr(init), _ptr(src),
                                                _ptr(dst), _ptr(tr), C.byref(self._cp), C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.lib().phmm_model_destroy(self._h)
            self._h = None

    def set_probs(self, init_logp: np.ndarray, trans_logp: np.ndarray) -> None:
        """next candidate copy-number vector on the same topology (posterior.rs:483-501)"""
        a = np.ascontiguousarray(init_logp, dtype=np.float64)
        b = np.ascontiguousarray(trans_logp, dtype=np.float64)
        assert a.shape[0] == self.n_nodes and b.shape[0] == self.n_edges
        _ffi.check(_ffi.lib().phmm_model_set_probs(self._h, _ptr(a), _ptr(b)))

    # ---- dense: forward / backward / run (forward.rs:24-45, backward.rs:24-53, freq.rs:42-46)
    def _tables(self, read: bytes, want_f: bool, want_b: bool):
        r = np.frombuffer(bytes(read), dtype=np.uint8)
        L, N = r.shape[0], self.n_nodes
        if L == 0:
            raise _ffi.PhmmError(_ffi.PHMM_EINVAL, "empty read")
        f = [np.empty((L, N)) for _ in range(3)] + [np.empty((L, 3))] if want_f else [None] * 4
        b = [np.empty((L, N)) for _ in range(3)] + [np.empty((L, 3))] if want_b else [None] * 4
        _ffi.check(_ffi.lib().phmm_dense_tables(self._h, _ptr(r), L, *[_ptr(x) for x in f], *[_ptr(x) for x in b]))
        return (DenseTables(*f) if want_f else None, DenseTables(*b) if want_b else None)

    def forward(self, read: bytes) -> DenseTables:
        return self._tables(read, True, False)[0]

    def backward(self, read: bytes) -> DenseTables:
        return self._tables(read, False, True)[1]

    def run(self, read: bytes) -> "PHMMOutput":
        f, b = self._tables(read, True, True)
        return PHMMOutput(self, bytes(read), f, b)

    def backward_sparse(self, read: bytes):
        """PHMMModel::backward_sparse (backward.rs:146-185) of one read -> (DenseTables with -inf where the
        reference's sparse table holds no element, is_dense[L])."""
        r = np.frombuffer(bytes(read), dtype=np.uint8)
        L, N = r.shape[0], self.n_nodes
        if L == 0:
            raise _ffi.PhmmError(_ffi.PHMM_EINVAL, "empty read")
        b = [np.empty((L, N)) for _ in range(3)] + [np.empty((L, 3))]
        dense = np.zeros(L, dtype=np.uint8)
        _ffi.check(_ffi.lib().phmm_backward_sparse_tables(self._h, _ptr(r), L, *[_ptr(x) for x in b], _ptr(dense)))
        return DenseTables(*b), dense.astype(bool)

    def run_sparse(self, reads: ReadCollection):
        """PHMMModel::run_sparse (freq.rs:51-55) over a read set -> (ln P forward[R], ln P backward[R],
        node_freq[N] summed over the reads)."""
        lf, lb, nf = np.empty(len(reads)), np.empty(len(reads)), np.empty(self.n_nodes)
        _ffi.check(_ffi.lib().phmm_run_sparse(self._h, reads._h, _ptr(lf), _ptr(lb), _ptr(nf)))
        return lf, lb, nf

    def to_full_prob_sparse_backward(self, reads: ReadCollection):
        """PHMMModel::to_full_prob_sparse_backward (freq.rs:153-163) -> (total ln P, per-read ln P)."""
        lp = np.empty(len(reads))
        tot = np.empty(1)
        _ffi.check(_ffi.lib().phmm_full_prob_sparse_backward(self._h, reads._h, _ptr(lp), _ptr(tot)))
        return float(tot[0]), lp

    # ---- read-set drivers
    def run_dense(self, reads: ReadCollection, want_backward: bool = True, want_freq: bool = True,
                  out_logp=None, out_logp_backward=None, out_node_freq=None):
        """`run` over a read set + to_full_prob_forward/backward + summed to_node_freqs
        (freq.rs:89-119, 245-255).  Output buffers may be numpy arrays or torch tensors
        (host or device); they are allocated as numpy when omitted."""
        R, N = len(reads), self.n_nodes
        lf = np.empty(R) if out_logp is None else out_logp
        lb = (np.empty(R) if out_logp_backward is None else out_logp_backward) if want_backward else None
        nf = (np.empty(N) if out_node_freq is None else out_node_freq) if want_freq else None
        _ffi.check(_ffi.lib().phmm_run_dense(self._h, reads._h, _ptr(lf), _ptr(lb), _ptr(nf)))
        return lf, lb, nf

    def to_full_prob_reads(self, reads: ReadCollection, mappings: Optional[Mappings] = None,
                           use_max_ratio: bool = True, out_logp=None):
        """PHMMModel::to_full_prob_reads (freq.rs:175-192) -> (total ln P(R|G), per-read ln P)."""
        lp = np.empty(len(reads)) if out_logp is None else out_logp
        tot = np.empty(1)
        _ffi.check(_ffi.lib().phmm_full_prob_reads(self._h, reads._h, mappings._h if mappings else None,
                                                   int(use_max_ratio), _ptr(lp), _ptr(tot)))
        return float(tot[0]), lp

    def to_full_prob_reads_candidates(self, reads: ReadCollection, mappings: Mappings, init_logp: np.ndarray,
                                      trans_logp: np.ndarray):
        """candidate-batched likelihood (posterior.rs:483-515): init [C,N], trans [C,E]."""
        a = np.ascontiguousarray(init_logp, dtype=np.float64)
        b = np.ascontiguousarray(trans_logp, dtype=np.float64)
        Cn = a.shape[0]
        lp = np.empty((Cn, len(reads)))
        tot = np.empty(Cn)
        _ffi.check(_ffi.lib().phmm_full_prob_reads_candidates(self._h, reads._h, mappings._h, Cn, _ptr(a), _ptr(b),
                                                              _ptr(lp), _ptr(tot)))
        return tot, lp

    def run_dense_edge_freqs(self, reads: ReadCollection):
        """PHMMOutput::to_edge_and_init_freqs (freq.rs:276-298) summed over the reads
        -> (per-read ln P, edge_freq[E], init_freq[N])."""
        lf, ef, nf = np.empty(len(reads)), np.empty(max(self.n_edges, 1)), np.empty(self.n_nodes)
        _ffi.check(_ffi.lib().phmm_run_dense_edges(self._h, reads._h, _ptr(lf), _ptr(ef), _ptr(nf)))
        return lf, ef[:self.n_edges], nf

    def q_score_exact(self, edge_freqs: np.ndarray, init_freqs: np.ndarray):
        """q_score_exact (src/hmmv2/q.rs:66-96) -> (init, trans, prior); QScore::total() is their sum."""
        ef = np.ascontiguousarray(edge_freqs, dtype=np.float64)
        nf = np.ascontiguousarray(init_freqs, dtype=np.float64)
        if ef.size < self.n_edges or nf.size < self.n_nodes:
            raise ValueError("edge_freqs / init_freqs shorter than the model")
        q = np.empty(3)
        _ffi.check(_ffi.lib().phmm_q_score_exact(self._h, _ptr(ef) if self.n_edges else None, _ptr(nf), _ptr(q)))
        return float(q[0]), float(q[1]), float(q[2])

    def to_full_prob_reads_copy_nums(self, reads: ReadCollection, mappings: Mappings, copy_nums: np.ndarray,
                                     min_copy_num: int = 0):
        """The same loop with candidates given as copy-number vectors [C,N] (what the sampler varies,
        posterior.rs:483-515); init / trans are derived on the device as SeqGraph::to_phmm does
        (seq_graph.rs:160-209)."""
        cn = np.ascontiguousarray(copy_nums, dtype=np.uint32)
        Cn = cn.shape[0]
        lp = np.empty((Cn, len(reads)))
        tot = np.empty(Cn)
        _ffi.check(_ffi.lib().phmm_full_prob_reads_copy_nums(self._h, reads._h, mappings._h, Cn, _ptr(cn),
                                                             int(min_copy_num), _ptr(lp), _ptr(tot)))
        return tot, lp

    def generate_mappings(self, reads: ReadCollection, mappings: Optional[Mappings] = None,
                          use_max_ratio: bool = True, out_node_freq=None):
        """PHMMModel::generate_mappings (hint.rs:193-220) -> (Mappings, node_freq[N])."""
        h = C.c_void_p()
        nf = np.empty(self.n_nodes) if out_node_freq is None else out_node_freq
        _ffi.check(_ffi.lib().phmm_generate_mappings(self._h, reads._h, mappings._h if mappings else None,
                                                     int(use_max_ratio), C.byref(h), _ptr(nf)))
        return Mappings(h, reads), nf


class PHMMOutput:
    """PHMMOutput (src/hmmv2/table.rs:450-517) of one dense run."""

    def __init__(self, model: PHMMModel, read: bytes, forward: DenseTables, backward: DenseTables):
        self.model, self.read, self.forward, self.backward = model, read, forward, backward

    def to_full_prob_forward(self) -> float:
        return float(self.forward.e[-1])

    def to_full_prob_backward(self) -> float:
        return float(self.backward.mb[0])

    def to_emit_probs(self, merged_index: int):
        """table.rs:500-505 with the merged indexing of table.rs:414-434 (log values)."""
        L = len(self.forward)
        p = self.to_full_prob_forward()
        n = self.model.n_nodes
        ninf = np.full(n, -np.inf)
        if merged_index == 0:
            fm, fi, fd = ninf, ninf, ninf
        else:
            fm, fi, fd = (self.forward.m[merged_index - 1], self.forward.i[merged_index - 1],
                          self.forward.d[merged_index - 1])
        if merged_index >= L:
            pe = np.full(n, self.model.param.p_end)
            bm, bi, bd = pe, pe, pe
        else:
            bm, bi, bd = self.backward.m[merged_index], self.backward.i[merged_index], self.backward.d[merged_index]
        return fm + bm - p, fi + bi - p, fd + bd - p

    def to_node_freqs(self) -> np.ndarray:
        """freq.rs:245-255 through the device path (single-read run_dense)."""
        rc = ReadCollection([self.read])
        _, _, nf = self.model.run_dense(rc, True, True)
        return nf
