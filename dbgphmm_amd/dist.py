"""Multi-GPU: one process per GPU, reads sharded by bases, one all-reduce per evaluation.

The path shards over independent units (reads are independent given the model:
/root/reference/src/hmmv2/freq.rs:181-191, hint.rs:199-219).  Every rank holds the whole
model and its own contiguous shard of reads; the only exchange is ONE sum all-reduce of
``[sum ln P(R|X), node_freq[N]]`` per call (RCCL over xGMI on MI355X, gloo on CPU for the
tests) -- the rayon ``.product()`` / ``Mappings::to_node_freqs`` reductions of the reference.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_reads(lengths: Sequence[int], world_size: int) -> List[Tuple[int, int]]:
    """Contiguous read ranges [lo, hi) per rank, balanced on the number of bases."""
    n = len(lengths)
    total = int(np.sum(lengths)) if n else 0
    csum = np.concatenate([[0], np.cumsum(lengths)])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        k = int(np.searchsorted(csum, target, side="left"))
        bounds.append(min(max(k, bounds[-1]), n))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def shard_grid(n_candidates: int, lengths: Sequence[int], world_size: int):
    """2-D split of a candidate batch (multi_dbg/posterior.rs:504-515: candidates x reads are all independent):
    the ranks form a pc x pr grid, pc = the largest divisor of world_size that is <= n_candidates -- splitting
    candidates needs no reduction at all -- and the reads are cut pr ways only when there are fewer candidates than
    ranks.  -> per rank ((cand_lo, cand_hi), (read_lo, read_hi)); rank = cand_shard * pr + read_shard.
    The per-candidate totals are completed by ONE sum all-reduce of a [n_candidates] vector in which every rank
    fills the candidates it owns with the sum over its reads."""
    pc = max(d for d in range(1, world_size + 1) if world_size % d == 0 and d <= max(n_candidates, 1))
    pr = world_size // pc
    reads = shard_reads(lengths, pr)
    out = []
    for rank in range(world_size):
        c, r = divmod(rank, pr)
        lo, hi = n_candidates * c // pc, n_candidates * (c + 1) // pc
        out.append(((lo, hi), reads[r]))
    return out


def pack_partial(total_logp: float, node_freq: np.ndarray) -> np.ndarray:
    buf = np.empty(1 + node_freq.shape[0], dtype=np.float64)
    buf[0] = total_logp
    buf[1:] = node_freq
    return buf


def all_reduce_partial(buf, dist=None):
    """Sum [sum lnP, node_freq[N]] over ranks.  ``buf`` is a torch tensor (device tensor under
    nccl/RCCL, CPU tensor under gloo) or a numpy array (wrapped, gloo)."""
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: F811
    if isinstance(buf, np.ndarray):
        t = torch.from_numpy(buf)
        dist.all_reduce(t)
        return buf
    if buf.is_cuda and dist.get_backend() == "gloo":
        # rehearsal on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device)
        t = buf.cpu()
        dist.all_reduce(t)
        buf.copy_(t)
        return buf
    dist.all_reduce(buf)
    return buf
