"""GPU side of the multi-GPU path (one MI355X here; the 8-GPU run is the driver's): the sharded sum over
dist.shard_reads ranges equals the unsharded result through the HIP library, and `bench.py --gpus 2` really
starts two ranks (on one device they rendezvous over gloo: RCCL refuses two ranks per device)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import dbgphmm_amd as D
from dbgphmm_amd import dist as PD
from helpers import small_dbg_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_sum_equals_unsharded(gpu_lib):
    """freq.rs:181-191 / hint.rs:199-219 sharded: per-read results are bit-equal whatever the shard, the reduced
    [sum ln P, node_freq[N]] equal up to the order of the additions."""
    arrays, sg = small_dbg_model(2000, 16, 0.003, seed=12, min_copy_num=1)
    reads = D.sample_reads(arrays, 10 ** 9, 200, seed=5, max_reads=150)
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    mp, nf = gm.generate_mappings(rc, None, True)
    tot, lp = mp.read_logp()
    lf, lb, nfd = gm.run_dense(rc)
    for world in (2, 4, 8):
        red, red_d = np.zeros(1 + arrays.n_nodes), np.zeros(1 + arrays.n_nodes)
        for lo, hi in PD.shard_reads([len(r) for r in reads], world):
            src = D.ReadCollection(reads[lo:hi])
            smp, snf = gm.generate_mappings(src, None, True)
            stot, slp = smp.read_logp()
            assert np.array_equal(slp, lp[lo:hi])
            red += PD.pack_partial(stot, snf)
            slf, _, snfd = gm.run_dense(src)
            assert np.max(np.abs(slf - lf[lo:hi])) < 1e-10
            red_d += PD.pack_partial(float(slf.sum()), snfd)
        assert abs(red[0] - tot) < 1e-9 and np.max(np.abs(red[1:] - nf)) < 1e-9
        assert abs(red_d[0] - lf.sum()) < 1e-8 and np.max(np.abs(red_d[1:] - nfd)) < 1e-8


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_gpus_2_starts_two_ranks(gpu_lib, scaling):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg1", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg1", "--steps", "2",
                          "--warmup", "1", "--scaling", scaling], env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    lines = [l for l in two.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    j2 = json.loads(lines[0])
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2 and j2["scaling"] == scaling
    assert len(j2["config"]["per_rank_ms"]) == 2 and j2["config"]["backend"] in ("gloo", "nccl")
    if scaling == "strong":  # the same read set, sharded
        assert j2["config"]["total_bases"] == j1["config"]["total_bases"]
        assert j2["config"]["bases_rank0"] < j1["config"]["bases_rank0"]
    else:
        assert j2["config"]["total_bases"] > 1.8 * j1["config"]["total_bases"]


def test_bench_candidates_two_ranks_split_the_batch(gpu_lib):
    """--mode candidates --gpus 2: the candidates are dealt to the ranks (dist.shard_grid), one all-reduce of the
    [C] totals; candidate 0's total equals the one-GPU run's."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg1m", "--mode", "candidates", "--candidates", "6",
            "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = subprocess.run(base, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run(base + ["--gpus", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    j2 = json.loads([l for l in two.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j2["config"]["candidates_rank0"] == 3 and j1["config"]["candidates_rank0"] == 6
    assert abs(j2["config"]["sum_lnP_candidate0"] - j1["config"]["sum_lnP_candidate0"]) < 1e-9
