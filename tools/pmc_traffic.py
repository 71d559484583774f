#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into per-launch HBM traffic of the dense kernels.

Inputs (directories written by rocprofv3 --pmc <X> --kernel-trace -d <dir>):
  --calib-fetch / --calib-write : tools/pmc_calib under FETCH_SIZE / WRITE_SIZE
  --bench-fetch / --bench-write : bench.py --steps 1 --warmup 0 --no-cpu-baseline under the same

FETCH_SIZE / WRITE_SIZE are reported in KB (1024 B).  The calibration run moves a known byte
count in the dense kernels' own pattern (8 B per lane, 512 B per wave instruction); its
bytes/counter ratio is the correction applied to the bench kernels (MI355X_MICROARCH.md "HBM":
FETCH_SIZE reads half the bytes of wide streaming reads on gfx950; other widths must be
calibrated).  Output: one JSON document (stdout or --out).
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def load(d, counter, last_fraction=1.0):
    """per kernel: counter values in dispatch order; last_fraction < 1 keeps only the tail (the timed step
    of a `--warmup 1 --steps 1` run: the first step has no warm-up hints yet and groups the reads differently)"""
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    rows[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out = {}
    for k, v in rows.items():
        v.sort()
        vals = [x for _, x in v]
        if last_fraction < 1.0 and len(vals) >= 2:
            vals = vals[len(vals) - max(1, int(round(len(vals) * last_fraction))):]
        out[k] = vals
    return out


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    for k in ("calib-fetch", "calib-write", "bench-fetch", "bench-write"):
        ap.add_argument("--" + k, required=True)
    ap.add_argument("--calib-bytes", type=float, default=float(2 << 30))
    ap.add_argument("--last-fraction", type=float, default=1.0)
    ap.add_argument("--out")
    a = ap.parse_args()
    KB = 1024.0
    cf, cw = load(a.calib_fetch, "FETCH_SIZE"), load(a.calib_write, "WRITE_SIZE")

    def mean_of(rows, key):
        for k, v in rows.items():
            if key in k:
                return sum(v) / len(v)
        raise SystemExit(f"kernel {key} not found")

    n3 = (a.calib_bytes // 8 // 3) * 24
    calib = {
        "read8": {"bytes": a.calib_bytes, "FETCH_SIZE_KB": mean_of(cf, "calib_read8")},
        "read16": {"bytes": a.calib_bytes, "FETCH_SIZE_KB": mean_of(cf, "calib_read16")},
        "write8": {"bytes": a.calib_bytes, "WRITE_SIZE_KB": mean_of(cw, "calib_write8")},
        "write16": {"bytes": a.calib_bytes, "WRITE_SIZE_KB": mean_of(cw, "calib_write16")},
        "copy3x8": {"bytes_read": n3, "bytes_written": n3, "FETCH_SIZE_KB": mean_of(cf, "calib_copy3x8"),
                    "WRITE_SIZE_KB": mean_of(cw, "calib_copy3x8")},
    }
    calib["read8_factor"] = a.calib_bytes / (calib["read8"]["FETCH_SIZE_KB"] * KB)
    calib["read16_factor"] = a.calib_bytes / (calib["read16"]["FETCH_SIZE_KB"] * KB)
    calib["write8_factor"] = a.calib_bytes / (calib["write8"]["WRITE_SIZE_KB"] * KB)
    calib["write16_factor"] = a.calib_bytes / (calib["write16"]["WRITE_SIZE_KB"] * KB)
    # the dense kernels move one f64 per lane per access: the single-stream 8 B/lane factors apply.
    # copy3x8 (three such streams in, three out, interleaved) is the cross-check: with these factors it
    # reads back as copy3x8_fetch_ratio / copy3x8_write_ratio x its true byte count.
    fetch_corr = calib["read8_factor"]
    write_corr = calib["write8_factor"]
    calib["fetch_bytes_per_counted_byte"] = fetch_corr
    calib["write_bytes_per_counted_byte"] = write_corr
    calib["copy3x8_fetch_ratio"] = calib["copy3x8"]["FETCH_SIZE_KB"] * KB * fetch_corr / n3
    calib["copy3x8_write_ratio"] = calib["copy3x8"]["WRITE_SIZE_KB"] * KB * write_corr / n3

    bf, bw = load(a.bench_fetch, "FETCH_SIZE", a.last_fraction), load(a.bench_write, "WRITE_SIZE", a.last_fraction)
    kernels = {}
    for name in sorted(set(bf) | set(bw)):
        if "phmm::" not in name:
            continue
        f, w = bf.get(name, []), bw.get(name, [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        kernels[short(name)] = {
            "launches": max(len(f), len(w)),
            "FETCH_SIZE_KB_per_launch": fm, "WRITE_SIZE_KB_per_launch": wm,
            "read_bytes_per_launch": fm * KB * fetch_corr, "write_bytes_per_launch": wm * KB * write_corr,
            "traffic_bytes_per_launch": fm * KB * fetch_corr + wm * KB * write_corr,
        }
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (kernel-trace only), "
                     "bench.py --steps 1 --warmup 1 --no-cpu-baseline (timed step only) and tools/pmc_calib",
           "correction": "bytes = counter[KB] * 1024 * factor; factors from the 8 B/lane streaming kernels of "
                         "tools/pmc_calib.hip (FETCH_SIZE x2, WRITE_SIZE x1 on gfx950)",
           "calibration": calib, "kernels": kernels}
    s = json.dumps(doc, indent=1)
    if a.out:
        open(a.out, "w").write(s + "\n")
    else:
        print(s)


if __name__ == "__main__":
    main()
