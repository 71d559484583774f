#!/usr/bin/env python3
"""Headline benchmark: read-bases/s through forward+backward P(R|X) on MI355X.

One "step" = one pass of the hot path over the whole synthetic read set resident on
the GPU: per-read ln P(R|X) (forward), the backward pass and the per-node usage
posteriors.  At N>1 every rank holds the same graph and its own 20x read shard
(weak scaling); the only collective is one RCCL all-reduce of
[sum ln P, node_freq[N]] per step.

Prints ONE JSON line on rank 0 (contract in the task statement).
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
    # BASELINE.json configs[1]
    "cfg2": dict(genome=10_000, haplotypes=1, k=40, coverage=20, read_len=1000, p=0.001, mode="dense",
                 desc="cfg2: synthetic 10 kb haploid genome, 20x HiFi reads (p=0.001, L=1000), k=40 DBG, "
                      "dense forward+backward+node posteriors"),
    # BASELINE.json configs[0] shape (plumbing)
    "cfg1": dict(genome=1_000, haplotypes=1, k=16, coverage=10, read_len=200, p=0.001, mode="dense",
                 desc="cfg1: 1 kb haploid, 10x, L=200, k=16, dense forward+backward"),
}


def build_workload(name: str, rank: int):
    import dbgphmm_amd as D
    w = WORKLOADS[name]
    hap = D.random_genome(w["genome"], seed=2)
    haps = [hap] if w["haplotypes"] == 1 else [hap, D.diverge(hap, 0.01, seed=3)]
    sg = D.dbg_from_haplotypes(haps, w["k"])
    param = D.PHMMParams.uniform(w["p"]).with_(n_warmup=w["k"])
    arrays = D.vectorised_to_phmm(sg, param, 0)
    total = w["coverage"] * w["genome"] * w["haplotypes"]
    reads = D.sample_reads(arrays, total, w["read_len"], seed=1000 + rank)
    return arrays, reads, w


def cpu_baseline(arrays, reads, budget_s: float = 20.0):
    """The oracle (C restatement of the reference, OpenMP over reads = the rayon stand-in)
    timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import oracle as O
    O.build()
    om = O.Model(arrays)
    cores = min(os.cpu_count() or 1, 16)
    sample = list(reads[:cores])
    # one short calibration read, then as many full reads as fit the budget
    t0 = time.time()
    om.run_dense_reads([sample[0][:50]], n_threads=1)
    per_base = (time.time() - t0) / 50.0
    bases_budget = max(1, int(budget_s / max(per_base, 1e-9)))
    per_read = max(20, min(len(sample[0]), bases_budget))
    sample = [r[:per_read] for r in sample]
    t0 = time.time()
    om.run_dense_reads(sample, n_threads=cores)
    dt = time.time() - t0
    nb = sum(len(r) for r in sample)
    return {"value": nb / dt, "unit": "bases/s", "cores": cores, "kind": "port",
            "sample": f"{len(sample)} reads x {per_read} bases of the same workload, dense forward+backward+"
                      f"node posteriors, {cores} OpenMP threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import dbgphmm_amd as D
    from dbgphmm_amd import _ffi
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
    # [sum ln P, node_freq[N]] : the one buffer that is all-reduced
    red = torch.zeros(1 + N, dtype=torch.float64, device=dev)

    def step():
        model.run_dense(rc, True, True, out_logp=out_logp, out_node_freq=red[1:])
        red[0] = out_logp.sum()
        if dist is not None:
            dist.all_reduce(red)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    fwd_ms = bwd_ms = 0.0
    fwd_n = bwd_n = 0
    cells = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        ms, n, c = C.c_double(), C.c_uint64(), C.c_uint64()
        L.phmm_last_call_stats(0, C.byref(ms), C.byref(n), C.byref(c))
        fwd_ms += ms.value
        fwd_n += n.value
        cells = c.value
        L.phmm_last_call_stats(1, C.byref(ms), C.byref(n), C.byref(c))
        bwd_ms += ms.value
        bwd_n += n.value
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, float(n_bases)], dtype=torch.float64, device=dev)
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum)
        dt, total_bases = float(tmax[0]), float(tsum[1])
    else:
        total_bases = float(n_bases)

    if rank == 0:
        # roofline of the dominant kernel: bwd_step (backward column + fused posterior).
        # Algorithmic bytes per cell: B write 24 + B prev read 24 + F re-read 24 = 72
        # (forward step: F prev read 24 + F write 24 = 48); cells per launch = N x active reads.
        cells_per_launch = cells / max(bwd_n // max(args.steps, 1), 1)
        avg_s = (bwd_ms / max(bwd_n, 1)) * 1e-3
        achieved = 72.0 * cells_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
        fwd_avg_s = (fwd_ms / max(fwd_n, 1)) * 1e-3
        fwd_cells_per_launch = cells / max(fwd_n // max(args.steps, 1), 1)
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
                       "bases_per_gpu": n_bases, "cells_per_step_per_gpu": int(cells),
                       "parallelism": f"reads sharded over {world} GPU(s); one all-reduce of [sum lnP, node_freq]"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": None,
                         "kernel": "bwd_step", "avg_launch_us": avg_s * 1e6, "launches_per_step": bwd_n // max(args.steps, 1),
                         "algorithmic_bytes_per_cell": 72,
                         "fwd_step": {"avg_launch_us": fwd_avg_s * 1e6,
                                      "achieved": 48.0 * fwd_cells_per_launch / fwd_avg_s / 1e9 if fwd_avg_s > 0 else 0.0,
                                      "algorithmic_bytes_per_cell": 48}},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(arrays, reads)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
