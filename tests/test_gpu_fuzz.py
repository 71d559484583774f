"""Randomised GPU-vs-oracle parity (tests/fuzz_cases.py) on a fixed seed: 40 random graphs (k 8..20, haploid and
diploid, error rates 0.1 %..2 %), read sets of 3..90 reads, parameter overrides (warm-up threshold, score ratio,
Del-chain length).  tools/fuzz_parity.py runs the same cases in bulk (round 2: 1 680 cases of five seeds passed)."""
import numpy as np
import pytest

from fuzz_cases import check_case, check_case_medium, make_case, make_case_medium

pytestmark = pytest.mark.gpu


def test_random_cases_match_oracle(gpu_lib, oracle):
    import re
    rng = np.random.default_rng(20261005)
    seen = {"cases_with_forced": 0, "plain": 0, "reads": 0, "forced_reads": 0, "tie_reads": 0, "overflow_reads": 0}
    for case in range(40):
        tag = check_case(make_case(rng, case))
        seen["cases_with_forced" if "forced=" in tag else "plain"] += 1

        def num(pat):
            m = re.search(pat, tag)
            return int(m.group(1)) if m else 0
        seen["reads"] += num(r" reads=(\d+)")
        seen["forced_reads"] += num(r" forced=(\d+)")
        seen["tie_reads"] += num(r"tie-order reads=(\d+)")
        seen["overflow_reads"] += num(r"overflow reads=(\d+)")
    print("\nfuzz:", seen)
    # The relaxations are bounded.  Forced switches (outside the parity domain, DESIGN.md section 2: weaker checks only)
    # stay a minority of the cases and of the reads; reads that needed another tie order, or sat in the overflow
    # regime of the 400-slot vector, stay under 3 % of the reads compared (the sweeps of round 2 saw 11 in 500 cases).
    assert seen["plain"] >= 20, seen
    assert seen["forced_reads"] <= 0.25 * seen["reads"], seen
    assert seen["tie_reads"] + seen["overflow_reads"] <= 0.03 * (seen["reads"] - seen["forced_reads"]), seen


def test_random_medium_cases_match_oracle(gpu_lib, oracle):
    """15-60 kb genomes at k = 24..40, 70-260 reads of 300-1500 bases (several read groups; the oracle runs on a
    sample of each set that always holds the reads that took the rarer routes)."""
    rng = np.random.default_rng(20261006)
    for case in range(5):
        check_case_medium(make_case_medium(rng, case))
