#!/usr/bin/env python3
"""Random frames through aehip_trace_frame (1-6 bands sharing the GPU, random tile heights and sub-ranges of rows)
against the single render: the reference's scene and random single spheres (the gather of kept columns), scenes of
several shapes (whole rows), both builds, both passes, widths that are not multiples of four.  GPU only.

    python tools/frame_split_sweep.py [--cases 300] [--first 0]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    bad = 0
    for seed in range(a.first, a.first + a.cases):
        rng = np.random.default_rng(77000 + seed)
        w, h = int(rng.integers(17, 700)), int(rng.integers(9, 400))
        kind = int(rng.integers(0, 4))
        kw = dict(seed=seed)
        if kind == 0:
            kw.update(pass_=rt.PASS_PRIMARY)
        else:
            kw.update(spp=int(rng.choice([1, 4, 16, 70])), max_depth=int(rng.choice([1, 6])), jitter=bool(rng.integers(0, 2)))
        if kind == 2:       # a random single sphere: kept columns anywhere, or none
            s = np.zeros(1, dtype=rt.SPHERE_DTYPE)
            depth = float(rng.choice([-1.5, -3.0, -8.0]))
            s["center"][0] = (rng.uniform(-1.0, 1.0) * abs(depth), rng.uniform(-0.5, 0.5) * abs(depth), depth)
            s["radius"] = float(rng.choice([0.2, 0.6, 1.0]) * rng.uniform(0.7, 1.3))
            s["material"] = rt.MAT_NORMAL
            kw["spheres"] = s
        if kind == 3:
            kw["spheres"] = scenes.showcase_scene() if seed % 2 else scenes.divergence_scene()
            kw["spp"] = min(kw["spp"], 4)
        fast = bool(rng.integers(0, 2))
        tile_rows = int(rng.choice([0, 1, 3, 4, 8, 16]))
        T = tile_rows if tile_rows else 4
        row_begin = int(rng.integers(0, max(1, h // 3))) if rng.integers(0, 3) == 0 else 0
        row_count = int(rng.integers(1, h - row_begin + 1)) if row_begin or rng.integers(0, 4) == 0 else 0
        rows = row_count if row_count else h - row_begin
        bands = int(rng.integers(1, min(6, -(-rows // T)) + 1))
        whole, _ = rt.render(w, h, fast=fast, **{k: v for k, v in kw.items() if k != "pass_"}, pass_=kw.get("pass_", rt.PASS_PATH))
        split, st = rt.trace_frame(bands, w, h, fast=fast, tile_rows=tile_rows, row_begin=row_begin, row_count=row_count, **kw)
        if not np.array_equal(split, whole[row_begin:row_begin + rows]):
            bad += 1
            print(json.dumps({"mismatch": seed, "size": [w, h], "kind": kind, "fast": fast, "bands": bands, "tile_rows": tile_rows,
                              "rows": [row_begin, row_count], "pixels": int((split != whole[row_begin:row_begin + rows]).sum())}), flush=True)
        if (seed - a.first + 1) % 100 == 0:
            print(f"# {seed - a.first + 1} frames, {bad} mismatches", flush=True)
    print(json.dumps({"cases": a.cases, "first_seed": a.first, "mismatches": bad}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
