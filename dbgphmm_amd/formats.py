"""On-disk formats either side of the hot path: host-side readers / writers.

* DBG text format of ``MultiDbg`` (src/multi_dbg/output.rs:157-199 grammar, 203-336 reader): ``K``, ``N``,
  ``E`` lines; the PHMM is derived from it the way ``MultiDbg::to_phmm`` does
  (multi_dbg.rs:1370-1409 via to_node_centric_graph, multi_dbg.rs:1551-1604): one PHMM node per FULL edge
  (k-mer) with the edge's base and copy number, one PHMM edge per (incoming, outgoing) pair of full edges at
  every non-terminal full node.
* MAP text format of ``Mappings`` (output.rs:490-527 writer, 531-573 reader): ``read  pos  base
  node:logp,node:logp,...``; ``.gz`` / ``.mpz`` are gzip (output.rs:476-480).
* FASTA reads (common/collection.rs:225-270): ``>r{i}`` records, bases sanitised to ``ACGT``.

Nothing here touches the device; these are the files a ``draft`` / ``infer`` run of the reference leaves behind
and resumes from (bin/infer.rs:47-48, 90-94).
"""
from __future__ import annotations

import gzip
import io
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .graph import NULL_BASE, SeqGraph


def _open(path: str, mode: str):
    """text-mode open; gzip for .gz / .mpz / .dbz (output.rs:134-138, 476-480)"""
    if str(path).endswith((".gz", ".mpz", ".dbz")):
        return io.TextIOWrapper(gzip.open(path, mode + "b"), encoding="ascii")
    return open(path, mode, encoding="ascii")


# ------------------------------------------------------------------------------------------------ FASTA
def sanitize_bases(seq: bytes) -> bytes:
    """common/collection.rs:236-249: upper-case ACGT, anything else is an error (the reference panics)."""
    out = seq.upper()
    bad = set(out) - set(b"ACGT")
    if bad:
        raise ValueError(f"Non DNA base {chr(min(bad))!r} appeared in seq")
    return out


def read_fasta(path: str) -> List[bytes]:
    reads, cur = [], None
    with _open(path, "r") as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                if cur is not None:
                    reads.append(sanitize_bases("".join(cur).encode()))
                cur = []
            elif cur is not None:
                cur.append(line)
    if cur is not None:
        reads.append(sanitize_bases("".join(cur).encode()))
    return reads


def write_fasta(path: str, reads: Sequence[bytes]) -> None:
    """ReadCollection::to_fasta (collection.rs:225-234): records r0, r1, ..."""
    with _open(path, "w") as fh:
        for i, r in enumerate(reads):
            fh.write(f">r{i}\n{r.decode()}\n")


# ------------------------------------------------------------------------------------------------ MAP
def write_map(path_or_file, reads: Sequence[bytes], mapping_arrays, k: int = 0, n_edges_full: int = 0,
              n_edges_compact: int = 0) -> None:
    """MultiDbg::to_map_writer (output.rs:490-527).  mapping_arrays = (pos_off, nodes, logp) flat CSR."""
    po, nd, lp = mapping_arrays
    fh = _open(path_or_file, "w") if isinstance(path_or_file, str) else path_or_file
    try:
        fh.write("# dbgphmm_amd\n")
        fh.write(f"# k={k} n_edges_full={n_edges_full} n_edges_compact={n_edges_compact}\n")
        fh.write("# read\tpos\tbase\tnodes_and_probs\n")
        g = 0
        for i, read in enumerate(reads):
            fh.write(f"# i={i}\n")
            for j, base in enumerate(read):
                a0, a1 = int(po[g]), int(po[g + 1])
                items = ",".join(f"{int(nd[a])}:{float(lp[a])!r}" for a in range(a0, a1))
                fh.write(f"{i}\t{j}\t{chr(base)}\t{items}\n")
                g += 1
    finally:
        if isinstance(path_or_file, str):
            fh.close()


def read_map(path_or_file) -> Tuple[List[bytes], Tuple[np.ndarray, np.ndarray, np.ndarray]]:
    """MultiDbg::from_map_reader_raw (output.rs:531-573) -> (reads as written, (pos_off, nodes, logp)).
    Lines must come in (read, pos) order as the writer emits them (the reference asserts the same)."""
    fh = _open(path_or_file, "r") if isinstance(path_or_file, str) else path_or_file
    reads: List[bytearray] = []
    pos_off, nodes, logp = [0], [], []
    try:
        for line in fh:
            if not line.strip() or line.startswith("#"):
                continue
            parts = line.split()
            i, j, base = int(parts[0]), int(parts[1]), parts[2]
            if i == len(reads):
                reads.append(bytearray())
            if i != len(reads) - 1 or j != len(reads[i]):
                raise ValueError("MAP lines out of order")
            reads[i].append(ord(base))
            if len(parts) > 3:
                for item in parts[3].split(","):
                    n, p = item.split(":")
                    nodes.append(int(n))
                    logp.append(float(p))
            pos_off.append(len(nodes))
    finally:
        if isinstance(path_or_file, str):
            fh.close()
    return [bytes(r) for r in reads], (np.array(pos_off, dtype=np.uint64), np.array(nodes, dtype=np.uint32),
                                        np.array(logp, dtype=np.float64))


# ------------------------------------------------------------------------------------------------ DBG
@dataclass
class DbgFile:
    """The compact de Bruijn graph of a DBG file: nodes = (k-1)-mers, edges = unitigs carrying their full edges."""

    k: int
    km1mers: List[bytes] = field(default_factory=list)                 # N lines, by id
    edges: List[Tuple[int, int, bytes, int, List[int]]] = field(default_factory=list)  # (s, t, kmer field, copy_num, full ids)

    @property
    def n_edges_full(self) -> int:
        return sum(len(e[4]) for e in self.edges)

    def to_seq_graph(self) -> SeqGraph:
        """MultiDbg::to_seq_graph (multi_dbg.rs:1370-1392): PHMM node v = full edge v (base, copy number);
        PHMM edges = parents x children of every non-terminal full node (multi_dbg.rs:1580-1591)."""
        n_full = self.n_edges_full
        base = np.zeros(n_full, dtype=np.uint8)
        copy_num = np.zeros(n_full, dtype=np.int64)
        seen = np.zeros(n_full, dtype=bool)
        src, dst = [], []
        n_compact = len(self.km1mers)
        ins: List[List[int]] = [[] for _ in range(n_compact)]   # last full edge of the unitigs entering a node
        outs: List[List[int]] = [[] for _ in range(n_compact)]  # first full edge of the unitigs leaving it
        for (s, t, kmer, cn, full) in self.edges:
            seq = kmer[self.k - 1:]
            if len(seq) != len(full):
                raise ValueError("length of seq and edges_in_full is different")
            for i, e in enumerate(full):
                if e >= n_full or seen[e]:
                    raise ValueError("index of edge in full is wrong")
                seen[e] = True
                base[e] = seq[i]
                copy_num[e] = cn
                if i > 0:  # the simple full node between two full edges of one unitig
                    src.append(full[i - 1])
                    dst.append(e)
            outs[s].append(full[0])
            ins[t].append(full[-1])
        for v in range(n_compact):
            if all(b == NULL_BASE for b in self.km1mers[v]):
                continue  # no PHMM edges through the terminal node (add_terminal = false)
            for e1 in ins[v]:
                for e2 in outs[v]:
                    src.append(e1)
                    dst.append(e2)
        return SeqGraph(copy_num, base, np.array(src, dtype=np.uint32), np.array(dst, dtype=np.uint32), None)


def read_dbg(path_or_text: str, is_text: bool = False) -> DbgFile:
    """MultiDbg::from_dbg_reader (output.rs:203-336): K / N / E lines, '#' comments, anything else ignored."""
    fh = io.StringIO(path_or_text) if is_text else _open(path_or_text, "r")
    dbg: Optional[DbgFile] = None
    nodes: List[bytes] = []
    edges = []
    k = None
    try:
        for line in fh:
            if not line.strip():
                continue
            c = line[0]
            it = line.split()
            if c == "K":
                k = int(it[1])
            elif c == "N":
                if int(it[1]) != len(nodes):
                    raise ValueError("node is not sorted")
                nodes.append(it[2].encode())
            elif c == "E":
                if k is None:
                    raise ValueError("E line before K")
                if int(it[1]) != len(edges):
                    raise ValueError("edge is not sorted")
                edges.append((int(it[2]), int(it[3]), it[4].encode(), int(it[5]), [int(x) for x in it[6].split(",")]))
    finally:
        fh.close()
    if k is None:
        raise ValueError("no K line")
    dbg = DbgFile(k, nodes, edges)
    return dbg


def write_dbg(path_or_file, dbg: DbgFile) -> None:
    """MultiDbg::to_dbg_writer (output.rs:179-199)"""
    fh = _open(path_or_file, "w") if isinstance(path_or_file, str) else path_or_file
    try:
        fh.write("# dbgphmm_amd\n")
        fh.write(f"K\t{dbg.k}\n")
        for i, km in enumerate(dbg.km1mers):
            fh.write(f"N\t{i}\t{km.decode()}\n")
        for i, (s, t, kmer, cn, full) in enumerate(dbg.edges):
            fh.write(f"E\t{i}\t{s}\t{t}\t{kmer.decode()}\t{cn}\t{','.join(map(str, full))}\n")
    finally:
        if isinstance(path_or_file, str):
            fh.close()


def dbg_from_seq_graph_kmers(kmers: Sequence[bytes], copy_nums: Sequence[int], k: int) -> DbgFile:
    """A (non-compacted but grammatical) DBG file from a k-mer multiset: every (k-1)-mer is a node, every k-mer an
    edge holding one full edge with its own index -- what the reader turns back into exactly this k-mer graph.
    (The reference writes unitigs; its reader accepts any such decomposition.)"""
    ids = {}
    km1 = []

    def node_of(x: bytes) -> int:
        if x not in ids:
            ids[x] = len(km1)
            km1.append(x)
        return ids[x]

    edges = []
    for e, (km, cn) in enumerate(zip(kmers, copy_nums)):
        edges.append((node_of(km[:-1]), node_of(km[1:]), km, int(cn), [e]))
    return DbgFile(k, km1, edges)
