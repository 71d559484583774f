"""CPU, world_size 2, gloo: the N>1 path = read sharding + one all-reduce of
[sum ln P, node_freq[N]].  The second test drives bench.py's own N > 1 logic (build_workload's strong-scaling
shards of ONE read set, pack, all-reduce); with no GPU here the per-rank compute is the oracle's (test
infrastructure) -- tests/test_gpu_dist.py runs the same path through the HIP library on the GPU box."""
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


def test_shard_grid_covers_every_candidate_read_pair_once():
    rng = np.random.default_rng(1)
    lens = rng.integers(50, 1000, size=57).tolist()
    for C, ws in ((64, 8), (3, 8), (1, 4), (5, 6), (7, 1), (256, 2)):
        grid = PD.shard_grid(C, lens, ws)
        assert len(grid) == ws
        cover = np.zeros((C, len(lens)), dtype=int)
        for (c0, c1), (r0, r1) in grid:
            cover[c0:c1, r0:r1] += 1
        assert np.all(cover == 1)
        # candidates are split first: reads are only cut when there are fewer candidates than ranks
        n_read_shards = len({rr for _, rr in grid})
        assert n_read_shards == 1 or C < ws


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


def _bench_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle import oracle as O
    arrays, reads, w = bench.build_workload("cfg1", rank, world, "strong")
    lf, _, nf = O.Model(arrays).run_dense_reads(reads, n_threads=1)
    buf = PD.pack_partial(float(lf.sum()), nf)
    PD.all_reduce_partial(buf, dist)
    q.put((rank, len(reads), sum(map(len, reads)), buf))
    dist.destroy_process_group()


def test_bench_strong_scaling_shards_reduce_to_the_unsharded_result(oracle):
    """bench.py at N = 2 (BASELINE.json configs[3]: the SAME workload sharded): the ranks' read sets partition the
    one read set of the configuration, and the all-reduced [sum ln P, node_freq] equals the single-process result."""
    import bench
    import torch.multiprocessing as mp
    arrays, reads, w = bench.build_workload("cfg1", 0, 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] + got[1][1] == len(reads) and got[0][2] + got[1][2] == sum(map(len, reads))
    assert abs(got[0][2] - got[1][2]) <= 2 * max(map(len, reads))  # balanced on bases
    a0, r0, _ = bench.build_workload("cfg1", 0, 2)
    a1, r1, _ = bench.build_workload("cfg1", 1, 2)
    assert r0 + r1 == reads
    lf, _, nf = oracle.Model(arrays).run_dense_reads(reads, n_threads=2)
    for g in got:
        assert abs(g[3][0] - lf.sum()) < 1e-9 and np.max(np.abs(g[3][1:] - nf)) < 1e-9
    assert np.array_equal(got[0][3], got[1][3])
    # weak scaling: every rank its own read set
    _, w0, _ = bench.build_workload("cfg1", 0, 2, "weak")
    _, w1, _ = bench.build_workload("cfg1", 1, 2, "weak")
    assert w0 != w1 and abs(sum(map(len, w0)) - sum(map(len, reads))) < 2 * max(map(len, reads))


def test_bench_launcher_starts_n_ranks(tmp_path, monkeypatch):
    """`bench.py --gpus N` without RANK in the environment spawns N children with RANK / WORLD_SIZE / MASTER_* set
    and returns their worst exit code (checked here with a stand-in script: no GPU in this container)."""
    import bench
    import sys
    stub = tmp_path / "stub.py"
    stub.write_text("import os, sys\n"
                    "open(os.path.join(os.path.dirname(__file__), 'rank%s' % os.environ['RANK']), 'w').write("
                    "' '.join([os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'], os.environ['MASTER_ADDR'], os.environ['MASTER_PORT']] + sys.argv[1:]))\n"
                    "sys.exit(3 if os.environ['RANK'] == '1' and '--fail' in sys.argv else 0)\n")
    monkeypatch.setattr(bench, "__file__", str(stub))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "2"])
    monkeypatch.delenv("RANK", raising=False)
    assert bench.spawn_ranks(3) == 0
    seen = [(tmp_path / f"rank{r}").read_text().split() for r in range(3)]
    assert all(s[0] == "3" and s[1] == str(r) and s[2] == "127.0.0.1" for r, s in enumerate(seen))
    assert len({s[3] for s in seen}) == 1 and seen[0][4:] == ["--gpus", "3", "--steps", "2"]
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--fail"])
    assert bench.spawn_ranks(2) == 3
