#!/usr/bin/env python3
"""Headline benchmark: read-bases/s through forward+backward P(R|X) on MI355X.

One "step" = one pass of the hot path over the whole synthetic read set of this rank:

  cfg3 (default, BASELINE.json configs[2] -- the configuration the target is quoted on):
      100 kb diploid genome (1 % divergence), 20x HiFi reads (p = 0.001, L = 1000), k = 40 DBG.
      step = PHMMModel::generate_mappings(reads, None, true) = run_sparse_adaptive:
      forward_sparse (dense warm-up, then the <= 400-node frontier) + backward_by_forward +
      per-position node posteriors (Mappings) + per-node usage sums, exactly what `infer`
      runs once per k; per-read ln P(R|X) comes out of the same pass.
  cfg2 (BASELINE.json configs[1]): 10 kb haploid, dense forward + backward + node posteriors.

At N > 1 every rank holds the same graph and its own 20x read shard (weak scaling); the only
collective is ONE RCCL all-reduce of [sum ln P, node_freq[N]] per step.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
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
    "cfg1": dict(genome=1_000, haplotypes=1, k=16, coverage=10, read_len=200, p=0.001, mode="dense",
                 desc="cfg1: 1 kb haploid, 10x, L=200, k=16, dense forward+backward"),
}


def build_workload(name: str, rank: int):
    import dbgphmm_amd as D
    w = WORKLOADS[name]
    hap = D.random_genome(w["genome"], seed=3)
    haps = [hap] if w["haplotypes"] == 1 else [hap, D.diverge(hap, 0.01, seed=4)]
    sg = D.dbg_from_haplotypes(haps, w["k"])
    param = D.PHMMParams.uniform(w["p"]).with_(n_warmup=w["k"])
    # mapping generation uses the non-zero PHMM (multi_dbg/posterior.rs:616-619); with true copy
    # numbers >= 1 everywhere it equals to_phmm
    arrays = D.vectorised_to_phmm(sg, param, 1 if w["mode"] == "sparse" else 0)
    total = w["coverage"] * sum(len(h) for h in haps)
    reads = D.sample_reads(arrays, total, w["read_len"], seed=1000 + rank)
    return arrays, reads, w


def cpu_baseline(arrays, reads, mode: str, budget_s: float = 20.0):
    """The oracle (C restatement of the reference; OpenMP over reads = the rayon stand-in) timed on
    this box's host cores on a bounded sample of the same workload (about `budget_s` seconds)."""
    from oracle import oracle as O
    O.build()
    om = O.Model(arrays)
    cores = min(os.cpu_count() or 1, 16)

    def run(sample, threads):
        if mode == "dense":
            om.run_dense_reads(sample, n_threads=threads)
        else:
            om.generate_mappings(sample, None, True, n_threads=threads)

    full = [r for r in reads if len(r) >= 0.9 * max(map(len, reads))] or list(reads)
    if mode == "dense":
        # cost is proportional to bases: time a short prefix, then cut every read to fit the budget
        probe = full[0][:60]
        t0 = time.time()
        run([probe], 1)
        t_base = (time.time() - t0) / len(probe)
        per_read = max(20, min(len(full[0]), int(budget_s / max(t_base, 1e-9))))
        sample = [r[:per_read] for r in full[:cores]]
    else:
        # cost is dominated by the dense warm-up columns of every read: keep whole reads, time one
        # batch of `cores` reads and size the sample from it
        t0 = time.time()
        run(full[:cores], cores)
        t_batch = time.time() - t0
        n = int(cores * max(1.0, min(budget_s / max(t_batch, 1e-3), len(full) / cores)))
        sample = full[:n]
    t0 = time.time()
    run(sample, cores)
    dt = time.time() - t0
    nb = sum(len(r) for r in sample)
    what = "dense forward+backward+node posteriors" if mode == "dense" else \
        "generate_mappings (sparse-adaptive forward + backward_by_forward + posteriors)"
    return {"value": nb / dt, "unit": "bases/s", "cores": cores, "kind": "port",
            "sample": f"{len(sample)} reads, {nb} bases of the same workload, {what}, {cores} OpenMP threads, {dt:.1f} s"}


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same command
    (profiles/, made by tools/profile_r1.sh + tools/pmc_traffic.py: FETCH_SIZE / WRITE_SIZE in separate
    passes, corrected with the calibration kernels of tools/pmc_calib.hip).  bench.py cannot collect
    hardware counters itself; None when the summary is absent."""
    path = os.environ.get("PHMM_PMC_SUMMARY", os.path.join(ROOT, "profiles", "r1_cfg3_pmc_traffic.json"))
    try:
        doc = json.load(open(path))
        k = next(v for name, v in doc["kernels"].items() if name.startswith(kernel))
        return float(k["traffic_bytes_per_launch"]), os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # one rank per GPU; on a box with fewer GPUs than ranks (rehearsals) ranks share devices
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.device_count() >= int(os.environ.get("LOCAL_WORLD_SIZE", world)):
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
        else:
            dist.init_process_group("gloo")  # rehearsal only: RCCL refuses two ranks on one device
    dev = torch.device("cuda", local_rank)

    import dbgphmm_amd as D
    from dbgphmm_amd import _ffi
    from dbgphmm_amd import dist as PD
    L = _ffi.lib()
    _ffi.check(L.phmm_set_device(local_rank))
    _ffi.check(L.phmm_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    L.phmm_enable_timing(1)

    arrays, reads, w = build_workload(args.workload, rank)
    model = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    n_bases = rc.total_bases()
    N = model.n_nodes
    out_logp = torch.empty(len(rc), dtype=torch.float64, device=dev)
    # [sum ln P, node_freq[N]]: the one buffer that is all-reduced
    red = torch.zeros(1 + N, dtype=torch.float64, device=dev)
    state = {}

    def step():
        if w["mode"] == "dense":
            model.run_dense(rc, True, True, out_logp=out_logp, out_node_freq=red[1:])
            red[0] = out_logp.sum()
        else:
            mp, _ = model.generate_mappings(rc, None, True, out_node_freq=red[1:])
            state["mappings"] = mp
            tot, _ = mp.read_logp(out_logp)  # per-read ln P(R|X) of the same forward pass
            red[0] = tot
        if dist is not None:
            PD.all_reduce_partial(red, dist)

    def stats(which):
        ms, n, c = C.c_double(), C.c_uint64(), C.c_uint64()
        L.phmm_last_call_stats(which, C.byref(ms), C.byref(n), C.byref(c))
        return ms.value, n.value, c.value

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    acc = {0: [0.0, 0, 0], 1: [0.0, 0, 0]}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k in (0, 1):
            ms, n, c = stats(k)
            acc[k][0] += ms
            acc[k][1] += n
            acc[k][2] += c
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, float(n_bases)], dtype=torch.float64, device=dev)
    if dist is not None and dist.get_backend() == "gloo":
        t = t.cpu()
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum)
        dt, total_bases = float(tmax[0]), float(tsum[1])
    else:
        total_bases = float(n_bases)

    extra = {}
    if w["mode"] == "sparse" and rank == 0:
        # the inner loop of `infer` on the mappings just produced: hinted forward score
        # (to_full_prob_reads with mappings, freq.rs:175-192), not part of the timed step
        mp = state["mappings"]
        model.to_full_prob_reads(rc, mp)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            tot, _lp = model.to_full_prob_reads(rc, mp)
        torch.cuda.synchronize()
        dth = (time.perf_counter() - t1) / reps
        extra = {"hinted_forward_bases_per_s": n_bases / dth, "hinted_forward_ms": dth * 1e3,
                 "sum_lnP_hinted": tot, "mean_mapping_list": mp.arrays()[1].shape[0] / max(n_bases, 1)}

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
        # dominant kernel: bwd_step (backward column + fused F(.)B posterior): B write 24 + B prev read 24 +
        # F re-read 24 = 72 algorithmic bytes per cell; fwd_step: 24 + 24 = 48.
        rb, rf = roof(1, 72.0), roof(0, 48.0)
        traffic, traffic_src = pmc_traffic("phmm::bwd_step<64") if args.workload == "cfg3" else (None, None)
        out = {
            "metric": "read-bases/sec through forward+backward P(R|X)",
            "value": total_bases * args.steps / dt,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": w["desc"], "n_nodes": N, "n_edges": model.n_edges, "reads_per_gpu": len(rc),
                       "bases_per_gpu": n_bases, "dense_cells_per_step_per_gpu": int(acc[1][2] // max(args.steps, 1)),
                       "parallelism": f"reads sharded over {world} GPU(s); one all-reduce of [sum lnP, node_freq[N]]",
                       **extra},
            "roofline": {"bound": "hbm", "achieved": rb["achieved"] if rb else 0.0, "peak": 8000.0, "unit": "GB/s",
                         "frac": (rb["achieved"] / 8000.0) if rb else 0.0, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "kernel": "bwd_step<64>",
                         "algorithmic_bytes_per_launch": (72.0 * rb["cells_per_launch"]) if rb else None,
                         **({k: v for k, v in rb.items() if k != "achieved"} if rb else {}),
                         "fwd_step": rf},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(arrays, reads, w["mode"])
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
