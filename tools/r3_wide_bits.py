"""The 448-thread 400-slot class against the one-wave generic kernels: same bits (ln P per read, mapping lists) on
datasets that use the class heavily; wall time of generate_mappings with each."""
import os, sys, time
import numpy as np
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
import dbgphmm_amd as D
from repeat_cases import dataset
from fuzz_cases import make_case


def run(arrays, reads, no_wide):
    os.environ["PHMM_NO_WIDE_CLASS"] = "1" if no_wide else "0"
    gm = D.PHMMModel(arrays)
    rc = D.ReadCollection(reads)
    t0 = time.time()
    mp, nf = gm.generate_mappings(rc, None, True)
    dt = time.time() - t0
    t0 = time.time()
    mp, nf = gm.generate_mappings(rc, None, True)
    dt2 = time.time() - t0
    cols, flags = rc.last_call_info()
    return mp.read_logp()[1].copy(), [a.copy() for a in mp.arrays()], nf.copy(), flags.copy(), dt2


def cmp(tag, arrays, reads):
    a = run(arrays, reads, True)
    b = run(arrays, reads, False)
    dl = np.max(np.abs(a[0] - b[0])) if len(a[0]) else 0.0
    (pa, na, la), (pb, nb, lb) = a[1], b[1]
    same_cnt = np.array_equal(pa, pb)
    bad = 0
    dmax = 0.0
    if same_cnt:
        if np.array_equal(na, nb):
            dmax = float(np.max(np.abs(la - lb), initial=0.0))
        else:
            # positions whose lists differ as SETS (a different order of near-equal entries is not a difference)
            diff = np.flatnonzero(na != nb)
            pos = np.unique(np.searchsorted(pa, diff, side="right") - 1)
            for i in pos:
                s0, s1 = int(pa[i]), int(pa[i + 1])
                if sorted(na[s0:s1].tolist()) != sorted(nb[s0:s1].tolist()):
                    bad += 1
                else:
                    dmax = max(dmax, float(np.max(np.abs(np.sort(la[s0:s1]) - np.sort(lb[s0:s1])))))
    else:
        bad = int((np.diff(pa.astype(np.int64)) != np.diff(pb.astype(np.int64))).sum()) if pa.shape == pb.shape else -1
    dnf = float(np.max(np.abs(a[2] - b[2]), initial=0.0))
    # (a forced switch's nearly flat columns: a last-bit difference may move an entry across the ratio cut)
    good = dl < 1e-9 and 0 <= bad <= 1e-4 * max(1, len(pa) - 1) and dmax < 1e-9 and dnf < 1e-6
    print(f"{tag}: reads={len(reads)} N={arrays.n_nodes} flags!=0: {int((a[3] != 0).sum())} forced={int(((a[3] & 4) != 0).sum())} "
          f"max d lnP {dl:.3g}  list positions differing {bad}  max d list logp {dmax:.3g}  max d node_freq {dnf:.3g}  "
          f"generic {a[4]*1e3:.1f} ms  wide {b[4]*1e3:.1f} ms  {'ok' if good else 'DIFFERENT'}", flush=True)
    return good


ok = True
for name, k, cov in (("u20n200", 40, 20), ("u20", 40, 20), ("u100", 40, 20), ("u100n100", 40, 5)):
    arrays, reads, sg, haps = dataset(name, k, coverage=cov)
    ok &= cmp(f"{name} k={k}", arrays, reads)
rng = np.random.default_rng(77)
for n in range(30):
    case = make_case(rng, n)
    ok &= cmp(case["tag"], case["arrays"], case["reads"])
print("ALL WITHIN 1e-9" if ok else "DIFFERENCES")
sys.exit(0 if ok else 1)
