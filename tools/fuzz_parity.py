"""Randomised parity sweep (dev tool): tests/fuzz_cases.py over many cases.
usage: python tools/fuzz_parity.py [n_cases] [seed] [only_case|-1] [medium]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import check_case, check_case_medium, make_case, make_case_medium  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    medium = len(sys.argv) > 4 and sys.argv[4] == "medium"
    O.build()
    rng = np.random.default_rng(seed)
    t0 = time.time()
    for case in range(n):
        c = make_case_medium(rng, case) if medium else make_case(rng, case)
        if only >= 0 and case != only:
            continue
        tag = check_case_medium(c) if medium else check_case(c)
        print("ok", tag, f"[{time.time() - t0:.0f} s]", flush=True)
    print("all", n, "cases passed")


if __name__ == "__main__":
    main()
