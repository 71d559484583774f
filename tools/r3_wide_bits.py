"""The 448-thread 400-slot class against the one-wave generic kernels (tests/helpers.py: wide_class_vs_generic) on
more datasets than the GPU test runs, with the wall time of generate_mappings under each."""
import os, sys
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, ".."), os.path.join(here, "..", "tests")]
from helpers import wide_class_vs_generic
from repeat_cases import dataset
from fuzz_cases import make_case

cases = [(f"{n} k=40", *dataset(n, 40, coverage=c)[:2]) for n, c in (("u20n200", 20), ("u20", 20), ("u100", 20), ("u100n100", 5))]
rng = np.random.default_rng(77)
for n in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    c = make_case(rng, n)
    cases.append((c["tag"], c["arrays"], c["reads"]))
ok = True
for tag, arrays, reads in cases:
    r = wide_class_vs_generic(arrays, reads)
    good = (r["d_logp"] < 1e-9 and r["d_list_logp"] < 1e-9 and r["d_node_freq"] < 1e-6
            and r["list_positions_differing"] <= 1e-4 * r["positions"])
    ok &= good
    print(f"{tag}: reads={len(reads)} N={arrays.n_nodes} flagged={int((r['flags'] != 0).sum())} forced={int(((r['flags'] & 4) != 0).sum())} "
          f"d lnP {r['d_logp']:.3g} lists differing {r['list_positions_differing']}/{r['positions']} d list {r['d_list_logp']:.3g} "
          f"d node_freq {r['d_node_freq']:.3g} generic {r['generic_s']*1e3:.1f} ms wide {r['wide_s']*1e3:.1f} ms {'ok' if good else 'DIFFERENT'}", flush=True)
print("ALL WITHIN 1e-9" if ok else "DIFFERENCES")
sys.exit(0 if ok else 1)
