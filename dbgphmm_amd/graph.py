"""Host-side graph -> PHMM parameter builders (numpy, no device work).

Mirrors the reference's caller-side adapters so that tests read like the reference's:

* ``SeqGraph.to_phmm / to_uniform_phmm / to_non_zero_phmm``
  -- /root/reference/src/graph/seq_graph.rs:160-273
* ``GenomeGraph.to_seq_graph`` -- src/graph/genome_graph.rs:252-397 (forward strand)
* mocks -- src/graph/mocks.rs:8-62, src/hmmv2/mocks.rs:27-54, src/multi_dbg/toy.rs:260-303
* ``dbg_from_haplotypes`` -- the k-mer graph -> node-centric graph of
  src/multi_dbg.rs:1370-1409, 1551-1604 (PHMM node = k-mer, emission = last base,
  no edges through the all-``n`` terminal), built from true genome k-mers with true
  copy numbers.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .params import PHMMParams

NULL_BASE = ord("n")  # src/common.rs NULL_BASE


def _ln(x: float) -> float:
    return math.log(x) if x > 0 else -math.inf


@dataclass
class PHMMArrays:
    """Flat PHMMModel (src/hmmv2/common.rs:61-64): the arrays that cross the C ABI."""

    param: PHMMParams
    emission: np.ndarray  # u8[N]
    init_logp: np.ndarray  # f64[N]
    edge_src: np.ndarray  # u32[E]
    edge_dst: np.ndarray  # u32[E]
    trans_logp: np.ndarray  # f64[E]
    is_emittable: Optional[np.ndarray] = None  # bool[N]

    @property
    def n_nodes(self) -> int:
        return int(self.emission.shape[0])

    @property
    def n_edges(self) -> int:
        return int(self.edge_src.shape[0])


@dataclass
class SeqGraph:
    """DiGraph<N: SeqNode, E: SeqEdge> (src/graph/seq_graph.rs): one node per base."""

    copy_num: np.ndarray  # i64[N]
    base: np.ndarray  # u8[N]
    edge_src: np.ndarray  # u32[E]
    edge_dst: np.ndarray  # u32[E]
    edge_copy_num: Optional[np.ndarray] = None  # i64[E], -1 = None

    def is_emittable(self) -> np.ndarray:
        return self.base != NULL_BASE

    # seq_graph.rs:160-223
    def _to_phmm(self, param: PHMMParams, min_copy_num: int) -> PHMMArrays:
        n = self.base.shape[0]
        emit = self.is_emittable()
        cn = np.maximum(self.copy_num, min_copy_num)
        total = int(np.where(emit, cn, 0).sum())
        init = np.full(n, -np.inf)
        with np.errstate(divide="ignore"):
            init[emit] = np.log(cn[emit].astype(np.float64)) - math.log(total) if total > 0 else -np.inf
        e = self.edge_src.shape[0]
        trans = np.full(e, -np.inf)
        # total emittable child copy number per parent (seq_graph.rs:124-135)
        child_cn = np.where(emit[self.edge_dst], cn[self.edge_dst], 0)
        tot_child = np.zeros(n, dtype=np.int64)
        np.add.at(tot_child, self.edge_src, child_cn)
        for j in range(e):
            s, d = int(self.edge_src[j]), int(self.edge_dst[j])
            ecn = -1 if self.edge_copy_num is None else int(self.edge_copy_num[j])
            if ecn >= 0:  # seq_graph.rs:185-197
                if emit[d] and ecn > 0:
                    assert self.copy_num[s] > 0
                    trans[j] = _ln(ecn / int(self.copy_num[s]))
            else:  # seq_graph.rs:198-209
                if emit[d] and tot_child[s] > 0:
                    trans[j] = _ln(int(cn[d]) / int(tot_child[s]))
        return PHMMArrays(param, self.base.astype(np.uint8), init, self.edge_src.astype(np.uint32),
                          self.edge_dst.astype(np.uint32), trans, emit.copy())

    def to_phmm(self, param: PHMMParams) -> PHMMArrays:
        return self._to_phmm(param, 0)

    def to_non_zero_phmm(self, param: PHMMParams) -> PHMMArrays:
        """seq_graph.rs:263-273 (copy numbers clamped to >= 1; used for mapping)."""
        return self._to_phmm(param, 1)

    def to_uniform_phmm(self, param: PHMMParams) -> PHMMArrays:
        """seq_graph.rs:224-262"""
        n = self.base.shape[0]
        emit = self.is_emittable()
        n_emit = int(emit.sum())
        init = np.where(emit, -math.log(n_emit) if n_emit else -np.inf, -np.inf).astype(np.float64)
        n_child = np.zeros(n, dtype=np.int64)
        np.add.at(n_child, self.edge_src, emit[self.edge_dst].astype(np.int64))
        with np.errstate(divide="ignore"):
            trans = np.where(emit[self.edge_dst], -np.log(n_child[self.edge_src].astype(np.float64)), -np.inf)
        return PHMMArrays(param, self.base.astype(np.uint8), init, self.edge_src.astype(np.uint32),
                          self.edge_dst.astype(np.uint32), trans.astype(np.float64), emit.copy())


def vectorised_to_phmm(sg: SeqGraph, param: PHMMParams, min_copy_num: int = 0) -> PHMMArrays:
    """Same result as ``SeqGraph._to_phmm`` for graphs without edge copy numbers,
    without the per-edge Python loop (used for 1e5..1e6-node synthetic DBGs)."""
    assert sg.edge_copy_num is None
    n = sg.base.shape[0]
    emit = sg.is_emittable()
    cn = np.maximum(sg.copy_num, min_copy_num)
    total = int(np.where(emit, cn, 0).sum())
    init = np.full(n, -np.inf)
    init[emit] = np.log(cn[emit].astype(np.float64)) - math.log(total)
    child_cn = np.where(emit[sg.edge_dst], cn[sg.edge_dst], 0)
    tot_child = np.zeros(n, dtype=np.int64)
    np.add.at(tot_child, sg.edge_src, child_cn)
    with np.errstate(divide="ignore", invalid="ignore"):
        trans = np.where((child_cn > 0) & (tot_child[sg.edge_src] > 0),
                         np.log(child_cn.astype(np.float64) / tot_child[sg.edge_src].astype(np.float64)),
                         -np.inf)
    return PHMMArrays(param, sg.base.astype(np.uint8), init, sg.edge_src.astype(np.uint32),
                      sg.edge_dst.astype(np.uint32), trans, emit.copy())


# ---------------------------------------------------------------- genome graph / mocks

def genome_graph_to_seq_graph(seqs: Sequence[bytes], copy_nums: Sequence[int],
                              edges: Sequence[Tuple[int, int, Optional[int]]] = ()) -> SeqGraph:
    """genome_graph.rs:252-397: each GenomeNode (sequence, copy number) becomes a chain of
    one-base nodes whose internal edges carry Some(copy_num); GenomeEdges connect tail->head."""
    base: List[int] = []
    cn: List[int] = []
    es: List[int] = []
    ed: List[int] = []
    ec: List[int] = []
    heads, tails = [], []
    for seq, c in zip(seqs, copy_nums):
        start = len(base)
        for b in seq:
            base.append(b)
            cn.append(c)
        for j in range(len(seq) - 1):
            es.append(start + j)
            ed.append(start + j + 1)
            ec.append(c)
        heads.append(start)
        tails.append(start + len(seq) - 1)
    for (s, t, c) in edges:
        es.append(tails[s])
        ed.append(heads[t])
        ec.append(-1 if c is None else c)
    return SeqGraph(np.array(cn, dtype=np.int64), np.array(base, dtype=np.uint8),
                    np.array(es, dtype=np.uint32), np.array(ed, dtype=np.uint32),
                    np.array(ec, dtype=np.int64))


def mock_linear() -> SeqGraph:
    """graph/mocks.rs:8-12"""
    return genome_graph_to_seq_graph([b"ATTCGATCGT"], [1])


def mock_linear_from(seq: bytes) -> SeqGraph:
    return genome_graph_to_seq_graph([seq], [1])


def mock_crossing(has_edge_copy_number: bool) -> SeqGraph:
    """graph/mocks.rs:44-62 with the sequences its own test pins (mocks.rs:81-84)."""
    seqs = [b"TGCTCTGGCG", b"ATTAGGAGCA", b"GCTGATAGGG", b"CGAAGATGAG"]
    if has_edge_copy_number:
        edges = [(0, 2, 2), (1, 2, 0), (0, 3, 0), (1, 3, 2)]
    else:
        edges = [(0, 2, None), (1, 2, None), (0, 3, None), (1, 3, None)]
    return genome_graph_to_seq_graph(seqs, [2, 2, 2, 2], edges)


def node_centric_from_dbg(node_is_terminal: Sequence[bool],
                          edges: Sequence[Tuple[int, int, int, int]]) -> SeqGraph:
    """MultiDbg::to_node_centric_graph with add_terminal=false (multi_dbg.rs:1551-1604):
    DBG edge (k-mer; src node, dst node, base, copy_num) -> PHMM node; for each
    non-terminal DBG node, parents x childs (petgraph order: newest edge first)."""
    n_nodes = len(node_is_terminal)
    incoming: List[List[int]] = [[] for _ in range(n_nodes)]
    outgoing: List[List[int]] = [[] for _ in range(n_nodes)]
    for e, (s, t, _b, _c) in enumerate(edges):
        outgoing[s].append(e)
        incoming[t].append(e)
    es, ed = [], []
    for v in range(n_nodes):
        if node_is_terminal[v]:
            continue
        for e1 in reversed(incoming[v]):
            for e2 in reversed(outgoing[v]):
                es.append(e1)
                ed.append(e2)
    return SeqGraph(np.array([c for (_, _, _, c) in edges], dtype=np.int64),
                    np.array([b for (_, _, b, _) in edges], dtype=np.uint8),
                    np.array(es, dtype=np.uint32), np.array(ed, dtype=np.uint32), None)


def toy_repeat() -> Tuple[SeqGraph, int]:
    """multi_dbg/toy.rs:260-303 (k=4): returns (node-centric graph, k)."""
    term = [True] + [False] * 13
    (nnn, nnt, ntc, tcc, ccc, cca, cag, agc, gca, agg, gga, gaa, aan, ann) = range(14)
    o = ord
    edges = [
        (nnn, nnt, o("T"), 1), (nnt, ntc, o("C"), 1), (ntc, tcc, o("C"), 1), (tcc, ccc, o("C"), 1),
        (ccc, cca, o("A"), 1), (cca, cag, o("G"), 1),
        (cag, agc, o("C"), 3), (agc, gca, o("A"), 3), (gca, cag, o("G"), 3),
        (cag, agg, o("G"), 1), (agg, gga, o("A"), 1), (gga, gaa, o("A"), 1),
        (gaa, aan, o("n"), 1), (aan, ann, o("n"), 1), (ann, nnn, o("n"), 1),
    ]
    return node_centric_from_dbg(term, edges), 4


# ---------------------------------------------------------------- synthetic genomes / DBG

_B = np.uint64(0x9E3779B97F4A7C15)  # odd multiplier for the rolling hash
_ENC = np.full(256, 4, dtype=np.uint64)
for _i, _c in enumerate(b"ACGT"):
    _ENC[_c] = _i


def random_genome(length: int, seed: int) -> np.ndarray:
    """i.i.d. uniform ACGT (random_seq.rs:9-18 semantics; our own PRNG stream)."""
    rng = np.random.default_rng(seed)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=length)]


def diverge(hap: np.ndarray, rate: float, seed: int) -> np.ndarray:
    """second haplotype: per-base divergence split evenly sub/ins/del
    (random_seq.rs:131-139; genome.rs:177-185)."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    u = rng.random(hap.shape[0])
    kind = rng.integers(0, 3, size=hap.shape[0])
    rnd = rng.integers(0, 4, size=hap.shape[0])
    for i, b in enumerate(hap.tolist()):
        if u[i] >= rate:
            out.append(b)
        elif kind[i] == 0:  # substitution to a different base
            c = int(acgt[rnd[i]])
            if c == b:
                c = int(acgt[(rnd[i] + 1) % 4])
            out.append(c)
        elif kind[i] == 1:  # insertion after
            out.append(b)
            out.append(int(acgt[rnd[i]]))
        # deletion: skip
    return np.array(out, dtype=np.uint8)


def mutate_exact(seq: np.ndarray, rate: float, rng: np.random.Generator) -> np.ndarray:
    """random_mutation_with_rng (random_seq.rs:163-181): exactly round(len * rate) edit operations, each at a
    uniformly drawn index of the ORIGINAL sequence and of a uniformly drawn kind (substitution to a different
    base / insertion of a random base left of x[i] / deletion of x[i], MutationProfile::uniform), sorted,
    de-duplicated and applied left to right with a running offset (random_seq.rs:97-120).  Our own PRNG stream:
    the property, not the reference's data."""
    seq = np.asarray(seq, dtype=np.uint8)
    n_mut = max(int(round(seq.shape[0] * rate)), 0)
    if n_mut == 0 or seq.shape[0] == 0:
        return seq.copy()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    ops = set()
    for _ in range(n_mut):
        ix = int(rng.integers(0, seq.shape[0]))
        other = [int(b) for b in acgt if int(b) != int(seq[ix])]
        b_mut = other[int(rng.integers(0, len(other)))]
        b_ins = int(acgt[int(rng.integers(0, 4))])
        kind = int(rng.integers(0, 3))
        # (index, kind order Mut < Ins < Del as in the reference's derive(Ord), base)
        ops.add((ix, kind, b_mut if kind == 0 else (b_ins if kind == 1 else 0)))
    out = seq.tolist()
    offset = 0
    for ix, kind, b in sorted(ops):
        j = ix + offset
        if j < 0 or j >= len(out):
            continue
        if kind == 0:
            out[j] = b
        elif kind == 1:
            out.insert(j, b)
            offset += 1
        else:
            del out[j]
            offset -= 1
    return np.array(out, dtype=np.uint8)


def tandem_repeat_polyploid_with_unique_homo_ends(unit_size: int, n_unit: int, unit_seed: int,
                                                  divergence_init: float, div_init_seed: int, end_length: int,
                                                  n_haplotypes: int, divergence_between_haplotypes: float,
                                                  div_seed: int) -> List[np.ndarray]:
    """genome.rs:294-340: a random unit repeated n_unit times, mutated once by `divergence_init` (the repeat's own
    divergence, H0), flanked by a unique prefix / suffix of `end_length` bases that every haplotype shares;
    haplotype 0 carries the repeat as it is, every further haplotype a copy of it mutated by
    `divergence_between_haplotypes` (H) -- all drawn from ONE generator seeded by `div_seed`.
    -> the haplotypes (the reference's `Genome`).  Argument order as in the reference."""
    unit = random_genome(unit_size, seed=unit_seed * 3 + 17)
    tandem = np.tile(unit, n_unit)
    tandem = mutate_exact(tandem, divergence_init, np.random.default_rng([div_init_seed, 0xD1]))
    prefix = random_genome(end_length, seed=(unit_seed + 1) * 3 + 18)
    suffix = random_genome(end_length, seed=(unit_seed - 1) * 3 + 1_000_019)
    haps = [np.concatenate([prefix, tandem, suffix])]
    rng = np.random.default_rng([div_seed, 0xD2])
    for _ in range(1, n_haplotypes):
        haps.append(np.concatenate([prefix, mutate_exact(tandem, divergence_between_haplotypes, rng), suffix]))
    return haps


def genome_phmm(haps: Sequence[np.ndarray], param: PHMMParams) -> PHMMArrays:
    """The PHMM of the genome itself: GenomeGraph::from_styled_seqs -> to_seq_graph -> to_phmm
    (genome_graph.rs:252-323, seq_graph.rs:160-211) for linear haplotypes of copy number 1 -- one node per base,
    chain edges of probability 1, uniform init.  The reference samples its datasets from this model
    (e2e.rs:163-232: generate_dataset)."""
    lens = [int(len(h)) for h in haps]
    n = sum(lens)
    base = np.concatenate([np.asarray(h, dtype=np.uint8) for h in haps])
    src = np.arange(n - 1, dtype=np.uint32)
    ends = np.cumsum(lens)[:-1] - 1  # no edge from the last base of a haplotype to the next haplotype
    src = np.delete(src, ends) if len(lens) > 1 else src
    init = np.full(n, -math.log(n))
    return PHMMArrays(param, base, init, src.astype(np.uint32), (src + 1).astype(np.uint32),
                      np.zeros(src.shape[0]), np.ones(n, dtype=bool))


def sample_genome_reads(haps: Sequence[np.ndarray], param: PHMMParams, coverage: int, read_len: int, seed: int,
                        max_reads: Optional[int] = None) -> List[bytes]:
    """generate_dataset (e2e.rs:163-232) on the forward strand: fragments of `read_len` states drawn from the
    genome's own PHMM at uniformly random start points until genome_size * coverage bases (a fragment that
    reaches the end of its haplotype stops there)."""
    gp = genome_phmm(haps, param)
    return sample_reads(gp, coverage * gp.n_nodes, read_len, seed, max_reads)


def _window_hashes(enc: np.ndarray, w: int) -> np.ndarray:
    """polynomial hash (mod 2^64) of every length-w window of enc (uint64 codes):
    sum_t (enc[i+t]+1) * B^(w-1-t), evaluated blockwise over a sliding-window view."""
    n = enc.shape[0]
    if n < w:
        return np.zeros(0, dtype=np.uint64)
    with np.errstate(over="ignore"):
        pw = np.ones(w, dtype=np.uint64)
        for j in range(1, w):
            pw[j] = pw[j - 1] * _B
        coef = pw[::-1].copy()
        win = np.lib.stride_tricks.sliding_window_view(enc + np.uint64(1), w)
        out = np.empty(win.shape[0], dtype=np.uint64)
        step = max(1, (1 << 24) // max(w, 1))
        for s0 in range(0, win.shape[0], step):
            blk = win[s0:s0 + step]
            out[s0:s0 + step] = (blk * coef).sum(axis=1, dtype=np.uint64)
        return out


def dbg_from_haplotypes(haps: Sequence[np.ndarray], k: int, with_occurrences: bool = False):
    """k-mer graph of the haplotypes with k-1 leading/trailing ``n`` pads
    (kmer/kmer.rs:60-75) as a node-centric SeqGraph: node = distinct k-mer,
    copy number = multiplicity, base = last base, edge u->v iff suffix_{k-1}(u) ==
    prefix_{k-1}(v) and that (k-1)-mer is not the all-``n`` terminal
    (multi_dbg.rs:1388, 1580-1591).  Nodes are numbered by first occurrence along the
    haplotypes, so unitigs are contiguous runs of ids."""
    pads = np.full(k - 1, NULL_BASE, dtype=np.uint8)
    km_hash, pre_hash, suf_hash, last_base, windows = [], [], [], [], []
    for h in haps:
        a = np.concatenate([pads, np.asarray(h, dtype=np.uint8), pads])
        enc = _ENC[a]
        hk = _window_hashes(enc, k)
        hk1 = _window_hashes(enc, k - 1)
        nk = hk.shape[0]
        km_hash.append(hk)
        pre_hash.append(hk1[:nk])
        suf_hash.append(hk1[1:nk + 1])
        last_base.append(a[k - 1:k - 1 + nk])
        windows.append(np.lib.stride_tricks.sliding_window_view(a, k))
    km = np.concatenate(km_hash)
    pre = np.concatenate(pre_hash)
    suf = np.concatenate(suf_hash)
    lb = np.concatenate(last_base)
    win = np.concatenate(windows) if len(windows) > 1 else windows[0]
    uniq, first, inv, counts = np.unique(km, return_index=True, return_inverse=True, return_counts=True)
    order = np.argsort(first, kind="stable")  # unique-id -> rank by first occurrence
    rank = np.empty_like(order)
    rank[order] = np.arange(order.shape[0])
    node_of_occ = rank[inv]
    n = uniq.shape[0]
    first_occ = first[order]
    # verify there was no hash collision: every occurrence equals its representative
    rep = win[first_occ]
    step = 1 << 18
    for s in range(0, win.shape[0], step):
        blk = slice(s, min(s + step, win.shape[0]))
        if not np.array_equal(win[blk], rep[node_of_occ[blk]]):
            raise RuntimeError("k-mer hash collision; change the hash multiplier")
    copy_num = counts[order].astype(np.int64)
    base = lb[first_occ]
    npre = pre[first_occ]
    nsuf = suf[first_occ]
    with np.errstate(over="ignore"):
        term = _window_hashes(np.full(k - 1, 4, dtype=np.uint64), k - 1)[0]
    # edges: join suffix(u) == prefix(v)
    porder = np.argsort(npre, kind="stable")
    ps = npre[porder]
    lo = np.searchsorted(ps, nsuf, side="left")
    hi = np.searchsorted(ps, nsuf, side="right")
    cnt = np.where(nsuf == term, 0, hi - lo)
    src = np.repeat(np.arange(n, dtype=np.int64), cnt)
    offs = np.concatenate([[0], np.cumsum(cnt)])
    within = np.arange(src.shape[0]) - offs[src]
    dst = porder[lo[src] + within]
    sg = SeqGraph(copy_num, base.astype(np.uint8), src.astype(np.uint32), dst.astype(np.uint32), None)
    if not with_occurrences:
        return sg
    # node of every k-mer occurrence, per haplotype (window j of the padded haplotype)
    cuts = np.cumsum([h.shape[0] for h in km_hash])[:-1]
    return sg, np.split(node_of_occ, cuts)


def kp1_node_map(haps: Sequence[np.ndarray], k: int):
    """The node map of MultiDbg::hint_kp1_from_hint_k (multi_dbg.rs:1325-1335): a node of the k-HMM (a k-mer) goes to
    the nodes of the (k+1)-HMM whose (k+1)-mer ends with it (the edges of the k+1 graph into the node that IS the
    k-mer).  -> (SeqGraph of k+1, map_off[N_k + 1], map_nodes[], SeqGraph of k) for `Mappings.map_nodes`."""
    sg_k, occ_k = dbg_from_haplotypes(haps, k, True)
    sg_k1, occ_k1 = dbg_from_haplotypes(haps, k + 1, True)
    pairs = []
    for a, b in zip(occ_k, occ_k1):
        # (k+1)-window j of the haplotype padded with k n's ends with k-window j of the one padded with k-1
        m = min(a.shape[0], b.shape[0])
        pairs.append(np.stack([a[:m], b[:m]], axis=1))
    pr = np.unique(np.concatenate(pairs), axis=0)
    n_k = sg_k.base.shape[0]
    cnt = np.bincount(pr[:, 0], minlength=n_k)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
    return sg_k1, off, pr[:, 1].astype(np.uint32), sg_k


# ---------------------------------------------------------------- read sampling

def sample_reads(model: PHMMArrays, n_bases_total: int, state_count: int, seed: int,
                 max_reads: Optional[int] = None) -> List[bytes]:
    """Reads drawn from the PHMM generative model itself (src/hmmv2/sample.rs:270-419):
    Begin -> pick_init_node by init_prob; {Match,Ins,Del} transitions by param; Match emits
    the node base with p_match else one of the other three, Ins emits uniform ACGT; stop after
    ``state_count`` transitions (ReadLength::StateCount) or when no child has p>0.  Reads are
    sampled until ``n_bases_total`` bases (ReadAmount::TotalBases, sample.rs:223-232).
    Our own PRNG stream (numpy PCG64); all reads of a batch are walked in lockstep."""
    p = model.param
    rng = np.random.default_rng(seed)
    n = model.n_nodes
    # padded child table with cumulative linear trans probs
    order = np.argsort(model.edge_src, kind="stable")
    csrc = model.edge_src[order].astype(np.int64)
    cdst = model.edge_dst[order].astype(np.int64)
    cpr = np.exp(model.trans_logp[order])
    deg = np.bincount(csrc, minlength=n)
    maxdeg = int(deg.max()) if deg.size else 0
    off = np.concatenate([[0], np.cumsum(deg)])
    slot = np.arange(csrc.shape[0]) - off[csrc]
    child = np.zeros((n, max(maxdeg, 1)), dtype=np.int64)
    cum = np.zeros((n, max(maxdeg, 1)))
    child[csrc, slot] = cdst
    cum[csrc, slot] = cpr
    cum = np.cumsum(cum, axis=1)
    tot = cum[:, -1]
    init = np.exp(model.init_logp)
    init_cdf = np.cumsum(init / init.sum())
    e = math.exp
    # rows: from-state M, I, D ; cols: to M, to I, to D  (sample.rs:345-386)
    tr = np.array([[e(p.p_MM), e(p.p_MI), e(p.p_MD)],
                   [e(p.p_IM), e(p.p_II), e(p.p_ID)],
                   [e(p.p_DM), e(p.p_DI), e(p.p_DD)]])
    pm = e(p.p_match)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    B, IB, M, I, D, DEAD = 0, 1, 2, 3, 4, 5
    reads: List[bytes] = []
    total = 0
    batch = 64
    while total < n_bases_total and (max_reads is None or len(reads) < max_reads):
        if max_reads is not None:
            batch = min(batch, max_reads - len(reads))
        R = batch
        node = np.searchsorted(init_cdf, rng.random(R), side="right").clip(0, n - 1)
        state = np.full(R, B)
        out = np.full((R, state_count), 255, dtype=np.uint8)
        for step in range(state_count):
            alive = state != DEAD
            if not alive.any():
                break
            u = rng.random(R)
            u2 = rng.random(R)
            u3 = rng.random(R)
            frm = np.where(state == B, 0, np.where(state == IB, 1, state - 2)).clip(0, 2)
            w = tr[frm]
            uu = u * w.sum(axis=1)
            to_m = uu < w[:, 0]
            to_i = (~to_m) & (uu < w[:, 0] + w[:, 1])
            to_d = ~(to_m | to_i)
            begin = (state == B) | (state == IB)
            # normal states need a child with p > 0 (sample.rs:430-447)
            has_child = tot[node] > 0
            dead_now = alive & ~begin & ~has_child
            ch_slot = (cum[node] <= (u2 * tot[node])[:, None]).sum(axis=1).clip(0, child.shape[1] - 1)
            ch = child[node, ch_slot]
            go = alive & ~dead_now
            # next node: begin -> the picked init node itself; normal M/D moves to the child
            new_node = np.where(begin, node, np.where(to_i, node, ch))
            new_state = np.where(to_m, M, np.where(to_i, np.where(begin, IB, I), D))
            # emission (picker.rs:21-44)
            nb = model.emission[new_node]
            mis = (np.searchsorted(acgt, nb).clip(0, 3) + 1 + (u3 * 3).astype(np.int64).clip(0, 2)) % 4
            rnd4 = (u3 * 4).astype(np.int64).clip(0, 3)
            em_match = np.where(rng.random(R) < pm, nb, acgt[mis])
            base = np.where(to_m, em_match, np.where(to_i, acgt[rnd4], 255)).astype(np.uint8)
            out[go, step] = base[go]
            # InsBegin re-picks the init node for its next transition (sample.rs:402-414)
            repick = np.searchsorted(init_cdf, rng.random(R), side="right").clip(0, n - 1)
            node = np.where(go, np.where(begin & to_i, repick, new_node), node)
            state = np.where(go, new_state, DEAD)
        for r in range(R):
            row = out[r]
            seq = row[row != 255].tobytes()
            if not seq:
                continue
            reads.append(seq)
            total += len(seq)
            if total >= n_bases_total or (max_reads is not None and len(reads) >= max_reads):
                break
        batch = min(batch * 2, 4096)
    return reads
