#!/usr/bin/env python3
"""Headline benchmark: read-bases/s through forward+backward P(R|X) on MI355X.

One "step" = one pass of the hot path over the synthetic read set:

  cfg3 (default, BASELINE.json configs[2] -- the configuration the target is quoted on):
      100 kb diploid genome (1 % divergence), 20x HiFi reads (p = 0.001, L = 1000), k = 40 DBG.
      step = PHMMModel::generate_mappings(reads, None, true) = run_sparse_adaptive:
      forward_sparse (dense warm-up, then the <= 400-node frontier) + backward_by_forward +
      per-position node posteriors (Mappings) + per-node usage sums, exactly what `infer`
      runs once per k; per-read ln P(R|X) comes out of the same pass.
  cfg2 (BASELINE.json configs[1]): 10 kb haploid, dense forward + backward + node posteriors.
  --mode candidates (cfg3): the inner loop of `infer` (multi_dbg/posterior.rs:483-515): C candidate
      copy-number vectors x all reads, hinted forward score on the mappings of the mapping step.

N > 1 (BASELINE.json configs[3]): ONE read set, sharded over the ranks by contiguous ranges balanced on bases
(dbgphmm_amd/dist.py) -- strong scaling; every rank holds the whole graph; the only collective is ONE RCCL
all-reduce of [sum ln P, node_freq[N]] per step.  `--scaling weak` gives every rank its own
20x read set instead.  `python bench.py --gpus N` without RANK in the environment starts the N ranks itself
(fresh child processes, created before this process touches a GPU); under torch.distributed.run it is a rank.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "cfg3": dict(genome=100_000, haplotypes=2, k=40, coverage=20, read_len=1000, p=0.001, mode="sparse",
                 desc="cfg3: synthetic 100 kb diploid genome (1% divergence), 20x HiFi reads (p=0.001, L=1000), "
                      "k=40 DBG; sparse-adaptive forward+backward+posteriors (generate_mappings)"),
    "cfg2": dict(genome=10_000, haplotypes=1, k=40, coverage=20, read_len=1000, p=0.001, mode="dense",
                 desc="cfg2: synthetic 10 kb haploid genome, 20x HiFi reads (p=0.001, L=1000), k=40 DBG, "
                      "dense forward+backward+node posteriors"),
    # BASELINE.json configs[4]'s graph and read set (the `infer` loop itself is the reference's control plane: per k one
    # generate_mappings on a new graph, then MAX_ITER candidate batches on its mappings, then the mappings carried to
    # k+1 -- multi_dbg/posterior.rs:698-826, 314-417, 483-515): --mode mapping / candidates / map_nodes are its steps
    "cfg5": dict(genome=1_000_000, haplotypes=2, k=40, coverage=20, read_len=1000, p=0.001, mode="sparse",
                 desc="cfg5: synthetic 1 Mb diploid genome (1% divergence), 20x HiFi reads (p=0.001, L=1000), k=40 DBG "
                      "(N = 1.3e6); sparse-adaptive forward+backward+posteriors (generate_mappings), chunked"),
    "cfg1m": dict(genome=5_000, haplotypes=2, k=16, coverage=10, read_len=200, p=0.001, mode="sparse",
                  desc="cfg1m: 5 kb diploid, 10x, L=200, k=16, sparse-adaptive flow (small: tests and rehearsals)"),
    "cfg1": dict(genome=1_000, haplotypes=1, k=16, coverage=10, read_len=200, p=0.001, mode="dense",
                 desc="cfg1: 1 kb haploid, 10x, L=200, k=16, dense forward+backward"),
    # the reference's own simulation (scripts/sim.sh:184-214, run_n4): draft -k 40 -C 10 -L 10000 -p 0.0003
    # -U 10000 -N 4 -E 2000 -H 0.01 --H0 0.0002 -P 2 -- tandem repeat of four 10 kb units, 10 kb HiFi reads
    "rep": dict(tandem=(10000, 4, 0, 0.0002, 1, 2000, 2, 0.01, 0), haplotypes=2, k=40, coverage=10, read_len=10000,
                p=0.0003, mode="sparse",
                desc="rep: sim.sh run_n4 shape -- tandem repeat 10 kb unit x 4 (H0=0.0002) + 2 kb unique ends, diploid "
                     "(H=0.01), 10x of 10 000-base reads (p=0.0003) drawn from the genome, k=40 DBG; generate_mappings"),
    # the widest frontier among the reference's own test datasets (hmmv2/tests/dbg.rs:73-75: unit 20 bp x 200,
    # 2 % divergence inside the repeat), scaled to fill a GPU: 200 such loci worth of reads (coverage 4000x)
    "rep20": dict(tandem=(20, 200, 0, 0.02, 0, 300, 2, 0.02, 0), haplotypes=2, k=40, coverage=400, read_len=1000,
                  p=0.001, mode="sparse",
                  desc="rep20: hmmv2/tests/dbg.rs u20n200 -- tandem repeat 20 bp unit x 200 (2 % divergence) + 300 bp "
                       "ends, diploid (2 %), 400x of 1000-base reads (p=0.001) drawn from the genome, k=40 DBG; "
                       "generate_mappings"),
}


def cfg_haplotypes(name: str):
    import dbgphmm_amd as D
    w = WORKLOADS[name]
    if "tandem" in w:  # genome.rs:294-340
        return D.tandem_repeat_polyploid_with_unique_homo_ends(*w["tandem"])
    hap = D.random_genome(w["genome"], seed=3)
    return [hap] if w["haplotypes"] == 1 else [hap, D.diverge(hap, 0.01, seed=4)]


def cfg_seq_graph(name: str):
    import dbgphmm_amd as D
    return D.dbg_from_haplotypes(cfg_haplotypes(name), WORKLOADS[name]["k"])


def build_workload(name: str, rank: int = 0, world: int = 1, scaling: str = "strong", read_len: int = 0):
    """-> (arrays, reads of THIS rank, workload dict).  strong: the one read set of the configuration
    (seed 1000) cut into `world` contiguous shards balanced on bases; weak: a full read set per rank.
    read_len > 0 overrides the configuration's read length (same coverage: fewer, longer reads)."""
    import dbgphmm_amd as D
    from dbgphmm_amd import dist as PD
    w = dict(WORKLOADS[name])
    if read_len > 0 and read_len != w["read_len"]:
        w["desc"] = w["desc"].replace(f"L={w['read_len']}", f"L={read_len}") + f" [--read-len {read_len}]"
        w["read_len"] = read_len
    sg = cfg_seq_graph(name)
    param = D.PHMMParams.uniform(w["p"]).with_(n_warmup=w["k"])
    # mapping generation uses the non-zero PHMM (multi_dbg/posterior.rs:616-619); with true copy
    # numbers >= 1 everywhere it equals to_phmm
    arrays = D.vectorised_to_phmm(sg, param, 1 if w["mode"] == "sparse" else 0)
    if "tandem" in w:
        # reads are fragments of the GENOME (generate_dataset, e2e.rs:163-232), not walks of the collapsed graph
        haps = cfg_haplotypes(name)

        def draw(seed):
            return D.sample_genome_reads(haps, param, w["coverage"], w["read_len"], seed)
    else:
        total = w["coverage"] * w["genome"] * w["haplotypes"]

        def draw(seed):
            return D.sample_reads(arrays, total, w["read_len"], seed=seed)
    if scaling == "weak":
        reads = draw(1000 + rank)
    else:
        reads = draw(1000)
        lo, hi = PD.shard_reads([len(r) for r in reads], world)[rank]
        reads = reads[lo:hi]
    return arrays, reads, w


def frontier_profile(rc, mp):
    """What the last generate_mappings call did on this read set (for the bench line): list lengths, reads that took
    the rarer routes, dense columns."""
    from dbgphmm_amd import _ffi
    cols, flags = rc.last_call_info()
    cnt = np.diff(mp.arrays()[0].astype(np.int64))
    n = max(cnt.shape[0], 1)
    return {"mean_list": float(cnt.mean()) if cnt.size else 0.0, "max_list": int(cnt.max()) if cnt.size else 0,
            "share_list_gt8": float((cnt > 8).sum()) / n, "share_list_gt16": float((cnt > 16).sum()) / n,
            "share_list_gt64": float((cnt > 64).sum()) / n,
            "forced_switch_reads": int(((flags & _ffi.PHMM_READ_FORCED_SWITCH) != 0).sum()),
            "wide_frontier_reads": int(((flags & _ffi.PHMM_READ_WIDE_FRONTIER) != 0).sum()),
            "deferred_reads": int(((flags & _ffi.PHMM_READ_DEFERRED) != 0).sum()),
            "mean_dense_cols": float(cols.mean()) if cols.size else 0.0, "max_dense_cols": int(cols.max()) if cols.size else 0}


def cpu_baseline(arrays, reads, mode: str, gpu_logp, gpu_check, budget_s: float = 20.0):
    """The oracle (C restatement of the reference; OpenMP over reads = the rayon stand-in) timed on
    this box's host cores on a bounded sample of the same workload (about `budget_s` seconds).  Its outputs
    are not thrown away: per-read ln P of the sample is compared with the GPU's (`max_abs_dlogp`)."""
    from oracle import oracle as O
    O.build()
    om = O.Model(arrays)
    cores = min(os.cpu_count() or 1, 16)
    lens = np.array([len(r) for r in reads])
    full_ix = np.flatnonzero(lens >= 0.9 * lens.max())
    if full_ix.size == 0:
        full_ix = np.arange(len(reads))

    def run(sample, threads):
        if mode == "dense":
            return om.run_dense_reads(sample, n_threads=threads)
        return om.generate_mappings(sample, None, True, n_threads=threads)

    if mode == "dense":
        # cost is proportional to bases: time a short prefix, then cut every read to fit the budget
        probe = reads[full_ix[0]][:60]
        t0 = time.time()
        run([probe], 1)
        t_base = (time.time() - t0) / len(probe)
        per_read = max(20, min(int(lens[full_ix[0]]), int(budget_s / max(t_base, 1e-9))))
        ix = full_ix[:cores]
        sample = [reads[r][:per_read] for r in ix]
    else:
        # cost is dominated by the dense warm-up columns of every read: keep whole reads, time one
        # batch of `cores` reads and size the sample from it
        t0 = time.time()
        run([reads[r] for r in full_ix[:cores]], cores)
        t_batch = time.time() - t0
        n = int(cores * max(1.0, min(budget_s / max(t_batch, 1e-3), full_ix.size / cores)))
        ix = full_ix[:n]
        sample = [reads[r] for r in ix]
    t0 = time.time()
    res = run(sample, cores)
    dt = time.time() - t0
    nb = sum(len(r) for r in sample)
    # parity of the sample: the oracle's per-read ln P against the GPU's
    if mode == "dense":
        glf = gpu_check(sample)  # the cut reads are different reads: the GPU runs the same bytes
        dlogp = float(np.max(np.abs(glf - res[0])))
    else:
        olp = om.full_prob_reads(sample, None, True, n_threads=cores)  # (forward only: a fraction of the timed work)
        dlogp = float(np.max(np.abs(gpu_logp[ix] - olp)))
    what = "dense forward+backward+node posteriors" if mode == "dense" else \
        "generate_mappings (sparse-adaptive forward + backward_by_forward + posteriors)"
    return {"value": nb / dt, "unit": "bases/s", "cores": cores, "kind": "port",
            "sample": f"{len(sample)} reads, {nb} bases of the same workload, {what}, {cores} OpenMP threads, {dt:.1f} s",
            "max_abs_dlogp": dlogp, "parity_reads": len(sample)}


def cpu_baseline_candidates(arrays, reads, mp_arrays, offsets, budget_s: float = 15.0):
    """oracle to_full_prob_reads(reads, Some(mappings)) (freq.rs:175-192) on a sample of the reads, one candidate."""
    from oracle import oracle as O
    O.build()
    om = O.Model(arrays)
    cores = min(os.cpu_count() or 1, 16)
    po, nd, lp = mp_arrays

    def cut(n):
        p1 = int(offsets[n])
        e1 = int(po[p1])
        return reads[:n], (po[:p1 + 1].copy(), nd[:e1].copy(), lp[:e1].copy())

    n = min(len(reads), 4 * cores)
    sub, mpa = cut(n)
    t0 = time.time()
    om.full_prob_reads(sub, mpa, True, n_threads=cores)
    t1 = time.time() - t0
    n = int(min(len(reads), max(n, n * budget_s / max(t1, 1e-3))))
    sub, mpa = cut(n)
    t0 = time.time()
    olp = om.full_prob_reads(sub, mpa, True, n_threads=cores)
    dt = time.time() - t0
    nb = sum(len(r) for r in sub)
    return {"value": nb / dt, "unit": "candidate-bases/s", "cores": cores, "kind": "port",
            "sample": f"{n} reads ({nb} bases) x 1 candidate, forward_with_mapping_score_only, {cores} OpenMP threads, {dt:.1f} s"}, olp


def pmc_traffic(kernel: str, workload: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same command
    (profiles/, made by tools/profile_r2.sh + tools/pmc_traffic.py: FETCH_SIZE / WRITE_SIZE in separate
    passes, corrected with the calibration kernels of tools/pmc_calib.hip).  bench.py cannot collect
    hardware counters itself; None when the summary is absent."""
    # (only a summary made with THIS round's kernels: an older one would pair other kernels' bytes with these times)
    for name in (os.environ.get("PHMM_PMC_SUMMARY"), f"profiles/r3_{workload}_pmc_traffic.json"):
        if not name:
            continue
        path = name if os.path.isabs(name) else os.path.join(ROOT, name)
        try:
            doc = json.load(open(path))
            k = next(v for nm, v in doc["kernels"].items() if nm.startswith(kernel))
            return float(k["traffic_bytes_per_launch"]), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def spawn_ranks(n: int) -> int:
    """`bench.py --gpus N` from a plain shell: start the N ranks as fresh child processes (this process has not
    touched a GPU and never will), one per device, rendezvous on 127.0.0.1.  Rank 0 prints the JSON line.
    The port is picked by binding port 0 and closing the socket: if somebody takes it before rank 0 binds (exit code
    PORT_TAKEN from init_process_group's short timeout), the launch is repeated once with a fresh port."""
    for attempt in range(2):
        rc = _spawn_ranks_once(n)
        if rc != PORT_TAKEN:
            return rc
    return rc


PORT_TAKEN = 75


def _spawn_ranks_once(n: int) -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:  # a rank died: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--mode", choices=("mapping", "candidates", "map_nodes"), default="mapping")
    ap.add_argument("--candidates", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spin-up", type=float, default=2.0,
                    help="seconds of extra UNTIMED steps behind the warm-up (not part of --warmup or --steps): the first "
                         "GPU process on a fresh box runs its first seconds of steps 10-15 %% slower than a warm one")
    ap.add_argument("--read-len", type=int, default=0, help="override the workload's read length (e.g. 10000: HiFi)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    # one rank per GPU; on a box with fewer GPUs than ranks (rehearsals) ranks share devices
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    backend = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.device_count() >= int(os.environ.get("LOCAL_WORLD_SIZE", world)):
            backend = "nccl"
        else:
            backend = "gloo"  # rehearsal only: RCCL refuses two ranks on one device
        import datetime
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),  # RCCL over xGMI
                                        timeout=datetime.timedelta(seconds=120))
            else:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
        except Exception as e:  # a failed rendezvous ends quickly and visibly (spawn_ranks retries a taken port once)
            print(f"rank {rank}: rendezvous failed: {e}", file=sys.stderr, flush=True)
            sys.exit(PORT_TAKEN if "address already in use" in str(e).lower() or "EADDRINUSE" in str(e) else 1)
    dev = torch.device("cuda", local_rank)

    import dbgphmm_amd as D
    from dbgphmm_amd import _ffi
    from dbgphmm_amd import dist as PD
    L = _ffi.lib()
    _ffi.check(L.phmm_set_device(local_rank))
    _ffi.check(L.phmm_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    L.phmm_enable_timing(1)

    cand_range = None
    if args.mode == "candidates" and world > 1:
        # 2-D split: candidates first (no reduction), reads only when there are fewer candidates than ranks
        arrays, all_reads, w = build_workload(args.workload, 0, 1, "strong", args.read_len)
        grid = PD.shard_grid(args.candidates, [len(r) for r in all_reads], world)
        cand_range, (rlo, rhi) = grid[rank]
        reads = all_reads[rlo:rhi]
    else:
        arrays, reads, w = build_workload(args.workload, rank, world, args.scaling, args.read_len)
    model = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    n_bases = rc.total_bases()
    N = model.n_nodes
    out_logp = torch.empty(len(rc), dtype=torch.float64, device=dev)
    # [sum ln P, node_freq[N]]: the one buffer that is all-reduced
    red = torch.zeros(1 + N, dtype=torch.float64, device=dev)
    state = {"ar_ms": 0.0}

    def reduce_partial():
        if dist is None:
            return
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        PD.all_reduce_partial(red, dist)
        torch.cuda.synchronize()
        state["ar_ms"] += (time.perf_counter() - t0) * 1e3

    cand = None
    if args.mode == "map_nodes":
        # Mapping::map_nodes over the read set (hint.rs:60-88) with the node map of hint_kp1_from_hint_k
        # (multi_dbg.rs:1325-1335): the mappings of the k graph carried to the k+1 graph
        if world > 1:
            raise SystemExit("--mode map_nodes is a one-GPU measurement")
        sg1, map_off, map_nodes, _ = D.kp1_node_map(cfg_haplotypes(args.workload), w["k"])
        model1 = D.PHMMModel(D.vectorised_to_phmm(sg1, arrays.param.with_(n_warmup=w["k"] + 1), 1))
        mp0, _ = model.generate_mappings(rc, None, True)
        for _ in range(max(args.warmup, 1)):
            mp1 = mp0.map_nodes(model1, map_off, map_nodes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            mp1 = mp0.map_nodes(model1, map_off, map_nodes)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        e0, e1 = int(mp0.arrays()[1].shape[0]), int(mp1.arrays()[1].shape[0])
        # what the carried lists are for: the hinted likelihood on the k+1 graph (to_full_prob_reads with mappings)
        tot1, lp1 = model1.to_full_prob_reads(rc, mp1)
        print(json.dumps({
            "metric": "read-bases/sec through Mapping::map_nodes (mappings of k carried to k+1)",
            "value": n_bases * args.steps / dt, "unit": "bases/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["desc"] + "; phmm_mappings_map_nodes to the k+1 graph", "n_nodes": N,
                       "n_nodes_kp1": model1.n_nodes, "reads": len(rc), "bases": n_bases, "entries_k": e0, "entries_kp1": e1,
                       "map_fan_out_max": int(np.diff(map_off.astype(np.int64)).max()),
                       "sum_lnP_hinted_kp1": tot1, "finite": bool(np.all(np.isfinite(lp1)))},
            "roofline": {"bound": "hbm", "achieved": 12.0 * (e0 + e1) * args.steps / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": 12.0 * (e0 + e1) * args.steps / dt / 1e9 / 8000.0, "traffic": None,
                         "kernel": "map_nodes_kernel", "algorithmic_bytes_per_launch": 12.0 * (e0 + e1),
                         "note": "12 B per list entry read + 12 B written; one wave per read position (latency-bound merge in LDS)"}}),
              flush=True)
        return
    if args.mode == "candidates":
        if w["mode"] != "sparse":
            raise SystemExit("--mode candidates runs on a sparse-flow workload (cfg3, cfg1m)")
        # candidates the way the sampler makes them: the current copy numbers with a few k-mers moved by +-1
        # (neighbour cycles of multi_dbg/neighbors.rs change a handful of edges); to_phmm semantics (min 0)
        sg = cfg_seq_graph(args.workload)
        rng = np.random.default_rng(5)
        cn = np.repeat(sg.copy_num.astype(np.uint32)[None, :], args.candidates, axis=0)
        for c in range(1, args.candidates):
            ix = rng.integers(0, N, size=16)
            cn[c, ix] = np.maximum(cn[c, ix].astype(np.int64) + rng.choice([-1, 1], size=16), 0).astype(np.uint32)
        c_lo, c_hi = cand_range if cand_range else (0, args.candidates)
        cand = np.ascontiguousarray(cn[c_lo:c_hi])
        mp0, _ = model.generate_mappings(rc, None, True)
        state["mappings"] = mp0
        cand_tot = torch.zeros(args.candidates, dtype=torch.float64, device=dev)  # the one vector that is all-reduced

    def step():
        if args.mode == "candidates":
            tot, _ = model.to_full_prob_reads_copy_nums(rc, state["mappings"], cand, 0)
            cand_tot.zero_()
            cand_tot[c_lo:c_hi] = torch.from_numpy(tot).to(dev)
            if dist is not None:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                PD.all_reduce_partial(cand_tot, dist)  # sum over the read shards of every candidate
                torch.cuda.synchronize()
                state["ar_ms"] += (time.perf_counter() - t0) * 1e3
            return
        if w["mode"] == "dense":
            model.run_dense(rc, True, True, out_logp=out_logp, out_node_freq=red[1:])
            red[0] = out_logp.sum()
        else:
            mp, _ = model.generate_mappings(rc, None, True, out_node_freq=red[1:])
            state["mappings"] = mp
            tot, _ = mp.read_logp(out_logp)  # per-read ln P(R|X) of the same forward pass
            red[0] = tot
        reduce_partial()

    def stats(which):
        ms, n, c = C.c_double(), C.c_uint64(), C.c_uint64()
        L.phmm_last_call_stats(which, C.byref(ms), C.byref(n), C.byref(c))
        return ms.value, n.value, c.value

    # the very first call: workspace allocation (the pool grows to its working size) and no grouping hints
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first_ms = None
    for i in range(args.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            first_ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    # spin-up (untimed): measured on cfg3 with the same build on one fresh box -- first process with --warmup 1
    # 252-256 ms per step, with --warmup 6 215 ms, every later process 220 ms whatever its warm-up.  A step of a big
    # workload (cfg5: 28 s) warms the device by itself: nothing is added when the warm-up took that long already.
    warm_s = time.perf_counter() - t0
    spin_steps = 0

    def agree(x):
        """the ranks' mean of x: the same number on every rank (a step holds an all-reduce: the count must agree)"""
        if dist is None:
            return float(x)
        v = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        PD.all_reduce_partial(v, dist)
        return float(v.item()) / world

    if agree(1.0 if (args.spin_up > 0 and warm_s < 12.0) else 0.0) == 1.0:
        t_spin = time.perf_counter()
        step()
        torch.cuda.synchronize()
        dt1 = max(time.perf_counter() - t_spin, 1e-3)
        more = int(agree(min(11.0, max(0.0, np.ceil(args.spin_up / dt1) - 1.0))))
        for _ in range(more):
            step()
        torch.cuda.synchronize()
        spin_steps = 1 + more
    if dist is not None:
        dist.barrier()
    state["ar_ms"] = 0.0
    acc = {0: [0.0, 0, 0], 1: [0.0, 0, 0], 2: [0.0, 0, 0]}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k in acc:
            ms, n, c = stats(k)
            acc[k][0] += ms
            acc[k][1] += n
            acc[k][2] += c
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    my_ms = dt / max(args.steps, 1) * 1e3
    t = torch.tensor([dt, float(n_bases)], dtype=torch.float64, device=dev)
    per_rank_ms = [my_ms]
    if dist is not None:
        if backend == "gloo":
            t = t.cpu()
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum)
        dt, total_bases = float(tmax[0]), float(tsum[1])
        gathered = [None] * world
        dist.all_gather_object(gathered, my_ms)
        per_rank_ms = [float(x) for x in gathered]
    else:
        total_bases = float(n_bases)
    if args.mode == "candidates" and world > 1:
        # every (candidate, base) pair is computed once: candidates x bases of the one read set
        cand_work = float(args.candidates) * float(sum(len(r) for r in all_reads))
    else:
        cand_work = float(args.candidates) * total_bases

    extra = {}
    if w["mode"] == "sparse" and rank == 0 and args.mode == "mapping":
        # a call without grouping hints on a warm pool: what `infer` sees at a new k (new graph, same reads)
        rc_cold = D.ReadCollection(reads)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model.generate_mappings(rc_cold, None, True)
        torch.cuda.synchronize()
        extra["cold_hint_ms"] = (time.perf_counter() - t1) * 1e3
        del rc_cold
        # the inner loop of `infer` on the mappings just produced: hinted forward score
        # (to_full_prob_reads with mappings, freq.rs:175-192), not part of the timed step
        mp = state["mappings"]
        extra["frontier"] = frontier_profile(rc, mp)
        model.to_full_prob_reads(rc, mp)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            tot, _lp = model.to_full_prob_reads(rc, mp)
        torch.cuda.synchronize()
        dth = (time.perf_counter() - t1) / reps
        extra.update({"hinted_forward_bases_per_s": n_bases / dth, "hinted_forward_ms": dth * 1e3,
                      "sum_lnP_hinted": tot, "mean_mapping_list": mp.arrays()[1].shape[0] / max(n_bases, 1)})

    if rank == 0:
        def roof(k, bytes_per_cell):
            ms, n, cells = acc[k]
            if n == 0 or ms <= 0:
                return None
            avg_s = ms / n * 1e-3
            cells_per_launch = cells / n
            ach = bytes_per_cell * cells_per_launch / avg_s / 1e9
            return {"avg_launch_us": avg_s * 1e6, "launches_per_step": n // max(args.steps, 1),
                    "cells_per_launch": cells_per_launch, "algorithmic_bytes_per_cell": bytes_per_cell,
                    "achieved": ach}
        common = {
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
        }
        if args.mode == "candidates":
            # hinted forward: one wave per (read, candidate); bytes that MUST move per cell = the 4-byte node id of
            # the list (the column itself lives in registers / LDS), so HBM is not what bounds it -- reported as
            # instruction-issue / latency bound with the SQ counters of profiles/ 
            mp = state["mappings"]
            cells = int(mp.arrays()[1].shape[0]) * int(cand.shape[0])
            ms, n, _c = acc[2]
            out = {"metric": "candidate-read-bases/sec through the hinted forward P(R|X') (inner loop of infer)",
                   "value": cand_work * args.steps / dt, "unit": "candidate-bases/s", **common,
                   "config": {"workload": w["desc"] + f"; {args.candidates} candidate copy-number vectors x all reads, "
                              "init/trans built on the device (phmm_full_prob_reads_copy_nums)",
                              "candidates": args.candidates, "candidates_rank0": int(cand.shape[0]), "n_nodes": N, "reads_rank0": len(rc), "bases_rank0": n_bases,
                              "list_cells_per_step": cells, "upload_bytes_per_step": int(cand.nbytes),
                              "sum_lnP_candidate0": float(cand_tot[0]), "per_rank_ms": per_rank_ms,
                              "all_reduce_ms_per_step": state["ar_ms"] / max(args.steps, 1), "backend": backend,
                              "grid": "candidates x read shards (dist.shard_grid)" if world > 1 else "one GPU"},
                   # (the contract's bounds are "hbm" | "mfma": this kernel is bound by neither -- VALU issue, see the note --
                   # so the HBM line is filled in with its 4 B / list cell and frac says how little that is)
                   "roofline": {"bound": "hbm", "achieved": 4.0 * cells * args.steps / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                                "frac": 4.0 * cells * args.steps / dt / 1e9 / 8000.0, "traffic": None, "limiter": "valu-issue",
                                "kernel": "hinted_lean_kernel", "algorithmic_bytes_per_launch": 4.0 * cells,
                                "kernel_ms_per_step": ms / max(args.steps, 1), "launches_per_step": n // max(args.steps, 1),
                                "note": "latency / instruction-issue bound (one wave walks one read for one candidate); "
                                        "HBM fraction reported for the contract only"}}
            if not args.no_cpu_baseline and world == 1:
                cb, olp = cpu_baseline_candidates(arrays, reads, mp.arrays(), rc.offsets)
                _, glp = model.to_full_prob_reads_copy_nums(rc, mp, cand[:1], 1)  # candidate 0 == the mapping model's copy numbers
                cb["max_abs_dlogp"] = float(np.max(np.abs(glp[0][:olp.shape[0]] - olp)))
                out["cpu_baseline"] = cb
            print(json.dumps(out), flush=True)
        else:
            # dominant kernel: bwd_step (backward column + fused F(.)B posterior): B write 24 + B prev read 24 +
            # F re-read 24 = 72 algorithmic bytes per cell; fwd_step: 24 + 24 = 48.
            rb, rf = roof(1, 72.0), roof(0, 48.0)
            kname = "bwd_step<64>" if w["mode"] == "sparse" else "bwd_step"
            traffic, traffic_src = pmc_traffic("phmm::bwd_step<64" if w["mode"] == "sparse" else "phmm::bwd_", args.workload)
            out = {
                "metric": "read-bases/sec through forward+backward P(R|X)",
                "value": total_bases * args.steps / dt, "unit": "bases/s", **common,
                "config": {"workload": w["desc"], "n_nodes": N, "n_edges": model.n_edges, "reads_rank0": len(rc),
                           "bases_rank0": n_bases, "total_bases": int(total_bases),
                           "dense_cells_per_step_rank0": int(acc[1][2] // max(args.steps, 1)),
                           "parallelism": f"one read set sharded over {world} GPU(s) by bases; one all-reduce of [sum lnP, node_freq[N]]"
                           if args.scaling == "strong" or world == 1 else
                           f"{world} GPU(s), each with its own 20x read set; one all-reduce of [sum lnP, node_freq[N]]",
                           "first_call_ms": first_ms, "spin_up_steps": spin_steps, "per_rank_ms": per_rank_ms,
                           "all_reduce_ms_per_step": state["ar_ms"] / max(args.steps, 1), "backend": backend, **extra},
                "roofline": {"bound": "hbm", "achieved": rb["achieved"] if rb else 0.0, "peak": 8000.0, "unit": "GB/s",
                             "frac": (rb["achieved"] / 8000.0) if rb else 0.0, "traffic": traffic, "traffic_unit": "bytes/launch",
                             "traffic_source": traffic_src, "kernel": kname,
                             "algorithmic_bytes_per_launch": (72.0 * rb["cells_per_launch"]) if rb else None,
                             **({k: v for k, v in rb.items() if k != "achieved"} if rb else {}),
                             "fwd_step": rf},
            }
            fr = extra.get("frontier") or {}
            if fr.get("share_list_gt64", 0.0) > 0.05:
                # a wide-frontier workload (tandem repeat with a unit shorter than k): the step is the 400-slot frontier
                # kernels', not the dense columns' -- say so next to the dense kernel's HBM line
                out["roofline"]["note"] = ("most of this workload's step is wide_forward_kernel / wide_backward_kernel "
                                           "(frontiers of 65-400 nodes: instruction-issue / barrier bound, SQ counters in "
                                           "profiles/r3_rep20_sq_counters.txt); this object describes the dense warm-up's "
                                           "bwd_step launches only")
            if not args.no_cpu_baseline and world == 1:
                def gpu_check(sample):
                    return model.run_dense(D.ReadCollection(sample), False, False)[0]
                out["cpu_baseline"] = cpu_baseline(arrays, reads, w["mode"], out_logp.cpu().numpy(), gpu_check)
            print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
