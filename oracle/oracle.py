"""ctypes binding of the CPU oracle (oracle/libphmm_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package ``dbgphmm_amd``.
See oracle/phmm_oracle.h for the citations and the parity-pin statement.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libphmm_oracle.so")

MAX_ACTIVE_NODES = 400

FWD_DENSE, FWD_MAPPING, FWD_SPARSE_TOPK, FWD_SPARSE_RATIO, FWD_SPARSE_V0_TOPK, FWD_SPARSE_V0_RATIO = range(6)
BWD_DENSE, BWD_MAPPING, BWD_SPARSE, BWD_BY_FORWARD = range(4)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "phmm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "phmm_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


class _MappingView(C.Structure):
    _fields_ = [("pos_off", C.c_void_p), ("nodes", C.c_void_p), ("logp", C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i64, u64, u32, dbl, i32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_uint32, C.c_double, C.c_int
        L.orc_last_error.restype = C.c_char_p
        L.orc_logadd.restype = dbl
        L.orc_logadd.argtypes = [dbl, dbl]
        L.orc_params_uniform.argtypes = [dbl, vp]
        L.orc_params_new.argtypes = [dbl, dbl, dbl, dbl, i64, i64, vp]
        L.orc_model_create.restype = vp
        L.orc_model_create.argtypes = [u32, u32, vp, vp, vp, vp, vp]
        L.orc_model_destroy.argtypes = [vp]
        L.orc_forward.restype = vp
        L.orc_forward.argtypes = [vp, vp, vp, u64, i32, vp]
        L.orc_backward.restype = vp
        L.orc_backward.argtypes = [vp, vp, vp, u64, i32, vp, vp]
        L.orc_tables_destroy.argtypes = [vp]
        L.orc_forward_score_only.argtypes = [vp, vp, vp, u64, vp, i32, vp]
        L.orc_tables_len.restype = i64
        L.orc_tables_len.argtypes = [vp]
        L.orc_tables_is_dense.argtypes = [vp, i64]
        L.orc_tables_get.argtypes = [vp, i64, vp, vp, vp, vp]
        L.orc_tables_nodes.restype = i64
        L.orc_tables_nodes.argtypes = [vp, i64, i32, vp]
        L.orc_tables_full_prob.restype = dbl
        L.orc_tables_full_prob.argtypes = [vp]
        L.orc_emit_probs.argtypes = [vp, vp, i64, vp, vp, vp, vp]
        L.orc_node_freqs.argtypes = [vp, vp, vp]
        L.orc_output_mapping.argtypes = [vp, vp, i32, i64, dbl, vp, vp, vp]
        L.orc_trans_and_init_probs.argtypes = [vp, vp, vp, vp, vp, u64, u64, vp, vp]
        L.orc_edge_and_init_freqs.argtypes = [vp, vp, vp, vp, vp, u64, vp, vp]
        L.orc_full_prob_reads.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp, i32, i32, vp]
        L.orc_generate_mappings.restype = vp
        L.orc_generate_mappings.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp, i32, i32]
        L.orc_mappings_total_positions.restype = u64
        L.orc_mappings_total_positions.argtypes = [vp]
        L.orc_mappings_total_entries.restype = u64
        L.orc_mappings_total_entries.argtypes = [vp]
        L.orc_mappings_export.argtypes = [vp, vp, vp, vp]
        L.orc_mappings_node_freqs.argtypes = [vp, u32, vp]
        L.orc_mappings_destroy.argtypes = [vp]
        L.orc_run_dense_reads.argtypes = [vp, vp, vp, vp, u64, i32, vp, vp, vp]
        _lib = L
    return _lib


def _err() -> str:
    return lib().orc_last_error().decode()


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def logadd(x: float, y: float) -> float:
    return lib().orc_logadd(x, y)


class Mapping:
    """One read's Mapping (hint.rs:27-30) as CSR arrays."""

    def __init__(self, pos_off: np.ndarray, nodes: np.ndarray, logp: np.ndarray):
        self.pos_off = np.ascontiguousarray(pos_off, dtype=np.uint64)
        self.nodes_flat = np.ascontiguousarray(nodes, dtype=np.uint32)
        self.logp_flat = np.ascontiguousarray(logp, dtype=np.float64)

    @staticmethod
    def from_lists(nodes: Sequence[Sequence[int]], logp: Optional[Sequence[Sequence[float]]] = None) -> "Mapping":
        off = np.zeros(len(nodes) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in nodes])
        flat = np.array([v for x in nodes for v in x], dtype=np.uint32)
        lp = np.zeros(flat.shape[0]) if logp is None else np.array([v for x in logp for v in x], dtype=np.float64)
        return Mapping(off, flat, lp)

    def __len__(self) -> int:
        return self.pos_off.shape[0] - 1

    def nodes(self, i: int) -> List[int]:
        return self.nodes_flat[int(self.pos_off[i]):int(self.pos_off[i + 1])].tolist()

    def probs(self, i: int) -> np.ndarray:
        return self.logp_flat[int(self.pos_off[i]):int(self.pos_off[i + 1])]

    def _view(self) -> _MappingView:
        return _MappingView(self.pos_off.ctypes.data, self.nodes_flat.ctypes.data, self.logp_flat.ctypes.data)


class Tables:
    """PHMMTables (table.rs:363-435)."""

    def __init__(self, handle: int, n_nodes: int):
        self._h = handle
        self.n_nodes = n_nodes

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_tables_destroy(self._h)
            self._h = None

    def __len__(self) -> int:
        return lib().orc_tables_len(self._h)

    def is_dense(self, i: int) -> bool:
        return bool(lib().orc_tables_is_dense(self._h, i))

    def table(self, i: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """(m, i, d, [mb, ib, e]) of tables[i] (i == -1: init_table), log values."""
        n = self.n_nodes
        m, ins, d, s = np.empty(n), np.empty(n), np.empty(n), np.empty(3)
        if lib().orc_tables_get(self._h, i, _ptr(m), _ptr(ins), _ptr(d), _ptr(s)):
            raise IndexError(_err())
        return m, ins, d, s

    def nodes(self, i: int, which: int = 0) -> List[int]:
        n = lib().orc_tables_nodes(self._h, i, which, None)
        idx = np.empty(max(n, 1), dtype=np.uint32)
        lib().orc_tables_nodes(self._h, i, which, _ptr(idx))
        return idx[:n].tolist()

    def full_prob(self) -> float:
        return lib().orc_tables_full_prob(self._h)


class Model:
    """Oracle-side PHMMModel; built from the same flat arrays the C ABI takes."""

    def __init__(self, arrays):
        self.arrays = arrays
        self.param = arrays.param
        self._cparam = arrays.param.to_c()
        self._em = np.ascontiguousarray(arrays.emission, dtype=np.uint8)
        self._init = np.ascontiguousarray(arrays.init_logp, dtype=np.float64)
        self._src = np.ascontiguousarray(arrays.edge_src, dtype=np.uint32)
        self._dst = np.ascontiguousarray(arrays.edge_dst, dtype=np.uint32)
        self._tr = np.ascontiguousarray(arrays.trans_logp, dtype=np.float64)
        self.n_nodes = self._em.shape[0]
        self.n_edges = self._src.shape[0]
        self._h = lib().orc_model_create(self.n_nodes, self.n_edges, _ptr(self._em), _ptr(self._init),
                                         _ptr(self._src), _ptr(self._dst), _ptr(self._tr))
        if not self._h:
            raise ValueError(_err())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_model_destroy(self._h)
            self._h = None

    def _p(self):
        return C.byref(self._cparam)

    @staticmethod
    def _read(read) -> np.ndarray:
        return np.frombuffer(bytes(read), dtype=np.uint8)

    # -- single read drivers -------------------------------------------------
    def forward(self, read, mode: int = FWD_DENSE, mapping: Optional[Mapping] = None) -> Tables:
        r = self._read(read)
        mv = mapping._view() if mapping is not None else None
        h = lib().orc_forward(self._h, self._p(), _ptr(r), r.shape[0], mode, C.byref(mv) if mv else None)
        if not h:
            raise RuntimeError(_err())
        return Tables(h, self.n_nodes)

    def backward(self, read, mode: int = BWD_DENSE, mapping: Optional[Mapping] = None,
                 forward: Optional[Tables] = None) -> Tables:
        r = self._read(read)
        mv = mapping._view() if mapping is not None else None
        h = lib().orc_backward(self._h, self._p(), _ptr(r), r.shape[0], mode, C.byref(mv) if mv else None,
                               forward._h if forward is not None else None)
        if not h:
            raise RuntimeError(_err())
        return Tables(h, self.n_nodes)

    def forward_score_only(self, read, mapping: Optional[Mapping] = None, use_max_ratio: bool = True) -> float:
        r = self._read(read)
        out = C.c_double()
        mv = mapping._view() if mapping is not None else None
        if lib().orc_forward_score_only(self._h, self._p(), _ptr(r), r.shape[0], C.byref(mv) if mv else None,
                                        int(use_max_ratio), C.byref(out)):
            raise RuntimeError(_err())
        return out.value

    def run(self, read) -> "Output":
        return Output(self, read, self.forward(read), self.backward(read))

    def run_sparse(self, read) -> "Output":
        return Output(self, read, self.forward(read, FWD_SPARSE_TOPK), self.backward(read, BWD_SPARSE))

    def run_sparse_adaptive(self, read, use_max_ratio: bool) -> "Output":
        f = self.forward(read, FWD_SPARSE_RATIO if use_max_ratio else FWD_SPARSE_TOPK)
        return Output(self, read, f, self.backward(read, BWD_BY_FORWARD, forward=f))

    def run_with_mapping(self, read, mapping: Mapping) -> "Output":
        return Output(self, read, self.forward(read, FWD_MAPPING, mapping),
                      self.backward(read, BWD_MAPPING, mapping))

    # -- read-set drivers ------------------------------------------------------
    @staticmethod
    def _pack(reads: Sequence[bytes]):
        off = np.zeros(len(reads) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in reads])
        bases = np.frombuffer(b"".join(bytes(r) for r in reads), dtype=np.uint8)
        if bases.shape[0] == 0:
            bases = np.zeros(1, dtype=np.uint8)
        return bases, off

    def full_prob_reads(self, reads: Sequence[bytes], mappings=None, use_max_ratio: bool = True,
                        n_threads: int = 0) -> np.ndarray:
        """to_full_prob_reads (freq.rs:175-192) -> per-read log P."""
        bases, off = self._pack(reads)
        out = np.empty(len(reads))
        mp = mappings
        rc = lib().orc_full_prob_reads(self._h, self._p(), _ptr(bases), _ptr(off), len(reads),
                                       _ptr(mp[0]) if mp else None, _ptr(mp[1]) if mp else None,
                                       _ptr(mp[2]) if mp else None, int(use_max_ratio), n_threads, _ptr(out))
        if rc:
            raise RuntimeError(_err())
        return out

    def generate_mappings(self, reads: Sequence[bytes], mappings=None, use_max_ratio: bool = True,
                          n_threads: int = 0):
        """generate_mappings (hint.rs:193-220) -> (pos_off[total+1], nodes, logp) flat CSR."""
        bases, off = self._pack(reads)
        mp = mappings
        h = lib().orc_generate_mappings(self._h, self._p(), _ptr(bases), _ptr(off), len(reads),
                                        _ptr(mp[0]) if mp else None, _ptr(mp[1]) if mp else None,
                                        _ptr(mp[2]) if mp else None, int(use_max_ratio), n_threads)
        if not h:
            raise RuntimeError(_err())
        try:
            tp = lib().orc_mappings_total_positions(h)
            te = lib().orc_mappings_total_entries(h)
            po = np.empty(tp + 1, dtype=np.uint64)
            nd = np.empty(max(te, 1), dtype=np.uint32)
            lp = np.empty(max(te, 1), dtype=np.float64)
            lib().orc_mappings_export(h, _ptr(po), _ptr(nd), _ptr(lp))
            nf = np.empty(self.n_nodes)
            lib().orc_mappings_node_freqs(h, self.n_nodes, _ptr(nf))
        finally:
            lib().orc_mappings_destroy(h)
        return (po, nd[:te], lp[:te]), nf

    def run_dense_reads(self, reads: Sequence[bytes], n_threads: int = 0):
        bases, off = self._pack(reads)
        lf, lb, nf = np.empty(len(reads)), np.empty(len(reads)), np.empty(self.n_nodes)
        if lib().orc_run_dense_reads(self._h, self._p(), _ptr(bases), _ptr(off), len(reads), n_threads,
                                     _ptr(lf), _ptr(lb), _ptr(nf)):
            raise RuntimeError(_err())
        return lf, lb, nf


class Output:
    """PHMMOutput (table.rs:450-517)."""

    def __init__(self, model: Model, read, forward: Tables, backward: Tables):
        self.model, self.read, self.forward, self.backward = model, bytes(read), forward, backward

    def to_full_prob_forward(self) -> float:
        return self.forward.full_prob()

    def to_full_prob_backward(self) -> float:
        return self.backward.full_prob()

    def to_emit_probs(self, merged_index: int):
        n = self.model.n_nodes
        m, ins, d, s = np.empty(n), np.empty(n), np.empty(n), np.empty(3)
        if lib().orc_emit_probs(self.forward._h, self.backward._h, merged_index, _ptr(m), _ptr(ins), _ptr(d), _ptr(s)):
            raise RuntimeError(_err())
        return m, ins, d, s

    def to_node_freqs(self) -> np.ndarray:
        out = np.empty(self.model.n_nodes)
        if lib().orc_node_freqs(self.forward._h, self.backward._h, _ptr(out)):
            raise RuntimeError(_err())
        return out

    def _mapping(self, by_ratio: int, n_active: int, max_ratio: float) -> Mapping:
        L = len(self.forward)
        po = np.empty(L + 1, dtype=np.uint64)
        nd = np.empty(L * MAX_ACTIVE_NODES, dtype=np.uint32)
        lp = np.empty(L * MAX_ACTIVE_NODES, dtype=np.float64)
        if lib().orc_output_mapping(self.forward._h, self.backward._h, by_ratio, n_active, max_ratio,
                                    _ptr(po), _ptr(nd), _ptr(lp)):
            raise RuntimeError(_err())
        t = int(po[L])
        return Mapping(po, nd[:t].copy(), lp[:t].copy())

    def to_mapping(self, n_active_nodes: int) -> Mapping:
        return self._mapping(0, n_active_nodes, 0.0)

    def to_mapping_by_score_ratio(self, max_ratio: float) -> Mapping:
        return self._mapping(1, 0, max_ratio)

    def to_trans_and_init_probs(self, i: int):
        """-> (tp[E,6], ip[N,6]) log values, columns mm,im,dm,md,id,dd (freq.rs:332-389)."""
        r = np.frombuffer(self.read, dtype=np.uint8)
        tp = np.empty((max(self.model.n_edges, 1), 6))
        ip = np.empty((self.model.n_nodes, 6))
        if lib().orc_trans_and_init_probs(self.model._h, self.model._p(), self.forward._h, self.backward._h,
                                          _ptr(r), r.shape[0], i, _ptr(tp), _ptr(ip)):
            raise RuntimeError(_err())
        return tp[:self.model.n_edges], ip

    def to_edge_and_init_freqs(self):
        r = np.frombuffer(self.read, dtype=np.uint8)
        ef = np.empty(max(self.model.n_edges, 1))
        nf = np.empty(self.model.n_nodes)
        if lib().orc_edge_and_init_freqs(self.model._h, self.model._p(), self.forward._h, self.backward._h,
                                         _ptr(r), r.shape[0], _ptr(ef), _ptr(nf)):
            raise RuntimeError(_err())
        return ef[:self.model.n_edges], nf


def map_nodes(mapping_arrays, map_off, map_nodes, max_active_nodes: int = 400):
    """Mapping::map_nodes (src/hmmv2/hint.rs:60-88) on flat CSR mappings (pure Python: small cases only).
    Per position: m[node_after] += prob / |node_map(node)| (Prob `+=` = log-add, prob.rs:181-197), then the
    `max_active_nodes` most probable in descending order (ties: by node id here; a HashMap's order there)."""
    po, nd, lp = mapping_arrays
    out_off, out_nodes, out_lp = [0], [], []
    for p in range(len(po) - 1):
        acc = {}
        for j in range(int(po[p]), int(po[p + 1])):
            a0, a1 = int(map_off[nd[j]]), int(map_off[nd[j] + 1])
            for q in range(a0, a1):
                v = float(lp[j]) - math.log(a1 - a0)
                k = int(map_nodes[q])
                acc[k] = v if k not in acc else float(np.logaddexp(acc[k], v))
        items = sorted(acc.items(), key=lambda kv: (-kv[1], kv[0]))[:max_active_nodes]
        out_nodes.extend(k for k, _ in items)
        out_lp.extend(v for _, v in items)
        out_off.append(len(out_nodes))
    return (np.array(out_off, dtype=np.uint64), np.array(out_nodes, dtype=np.uint32), np.array(out_lp, dtype=np.float64))
