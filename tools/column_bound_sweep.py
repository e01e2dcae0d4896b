#!/usr/bin/env python3
"""aehip_backend_background_columns on many random single-sphere views (the generator of
tests/test_gpu_sharding.py at other seeds): outside the reported columns every pixel of a row equals the
row's background pixel, both builds, both passes, jittered, every sample walked or not.  GPU only.

    python tools/column_bound_sweep.py [--cases 300]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_sharding import test_outside_the_reported_columns_every_row_is_one_colour as check  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    a = ap.parse_args()
    for case in range(12, 12 + a.cases):
        check(case)
        if case % 50 == 0:
            print(f"# {case - 11} views hold", flush=True)
    print(f"{a.cases} views: every pixel outside the reported columns is its row's background", flush=True)


if __name__ == "__main__":
    main()
