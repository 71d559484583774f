"""CPU, world_size 2, gloo: the N>1 path = read sharding + one all-reduce of
[sum ln P, node_freq[N]].  Per-rank compute is stood in by the oracle (test infrastructure);
the GPU product path plugs into the same dist helpers in bench.py."""
import os
import socket

import numpy as np
import pytest

import dbgphmm_amd as D
from dbgphmm_amd import dist as PD
from helpers import small_dbg_model


def test_shard_reads_balanced_and_complete():
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 1000, size=101).tolist()
    for ws in (1, 2, 3, 8):
        sh = PD.shard_reads(lens, ws)
        assert sh[0][0] == 0 and sh[-1][1] == len(lens)
        assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
        per = [sum(lens[lo:hi]) for lo, hi in sh]
        assert max(per) - min(per) <= 2 * max(lens)
    assert PD.shard_reads([], 4) == [(0, 0)] * 4
    assert PD.shard_reads([5], 2)[0][1] - PD.shard_reads([5], 2)[0][0] + PD.shard_reads([5], 2)[1][1] - PD.shard_reads([5], 2)[1][0] == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    arrays, _ = small_dbg_model(300, 12, 0.001, seed=4)
    reads = D.sample_reads(arrays, 10 ** 9, 60, seed=2, max_reads=9)
    lo, hi = PD.shard_reads([len(r) for r in reads], world)[rank]
    om = O.Model(arrays)
    if hi > lo:
        lf, _, nf = om.run_dense_reads(reads[lo:hi], n_threads=1)
        buf = PD.pack_partial(float(lf.sum()), nf)
    else:
        buf = PD.pack_partial(0.0, np.zeros(arrays.n_nodes))
    PD.all_reduce_partial(buf, dist)
    q.put((rank, buf))
    dist.destroy_process_group()


def test_two_rank_all_reduce_matches_single_process(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    arrays, _ = small_dbg_model(300, 12, 0.001, seed=4)
    reads = D.sample_reads(arrays, 10 ** 9, 60, seed=2, max_reads=9)
    lf, _, nf = oracle.Model(arrays).run_dense_reads(reads, n_threads=2)
    for r in (0, 1):
        assert abs(got[r][0] - lf.sum()) < 1e-9
        assert np.max(np.abs(got[r][1:] - nf)) < 1e-9
    assert np.array_equal(got[0], got[1])
