#!/usr/bin/env python3
"""Sweep of random single-sphere views on fine sample grids, where the fast build's hit loops run on Newton steps:
the fast build against the exact build and against itself with every sample walked in closed form, largest
difference of the averaged colour over every pixel of every view (the bars are 1e-5 and 8e-6).  Half of the views
have the sphere's centre in the camera's vertical plane (the hit loop's polynomial step), half beside it.  GPU only.

    python tools/newton_sweep.py [--views 200] [--first 0]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as rt  # noqa: E402
from test_gpu_parity import render_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    worst_exact = worst_closed = 0.0
    for seed in range(a.first, a.first + a.views):
        rng = np.random.default_rng(31000 + seed)
        cam = (float(rng.choice([0.0, rng.uniform(-1, 1)])), float(rng.choice([0.0, rng.uniform(-0.5, 0.5)])), float(rng.choice([1.0, 0.6, 2.0])))
        s = np.zeros(1, dtype=rt.SPHERE_DTYPE)
        x = cam[0] if seed % 2 == 0 else cam[0] + float(rng.uniform(-1.5, 1.5))
        s["center"][0] = (x, float(rng.choice([cam[1], rng.uniform(-1, 1)])), float(rng.uniform(-6, -1.5)))
        s["radius"] = float(rng.choice([0.3, 1.0, 1.4]) * rng.uniform(0.7, 1.2))
        s["material"] = rt.MAT_NORMAL
        w, h = [(1920, 1080), (2560, 1440), (1280, 720), (3840, 2160)][seed % 4]
        spp = int(rng.choice([256, 512, 1024, 2048]))
        kw = dict(spheres=s, cam=cam, spp=spp, max_depth=2, seed=seed, jitter=True)
        if (w, h) == (3840, 2160):
            kw.update(row_begin=int(rng.integers(0, 1700)), row_count=int(rng.integers(64, 400)))
        _, fast, _ = render_device(w, h, fast=True, **kw)
        _, closed, _ = render_device(w, h, fast=True, walk_all=True, **kw)
        _, exact, _ = render_device(w, h, **kw)
        d_exact = float(np.abs(fast[..., :3] - exact[..., :3]).max())
        d_closed = float(np.abs(fast - closed).max())
        worst_exact, worst_closed = max(worst_exact, d_exact), max(worst_closed, d_closed)
        print(json.dumps({"seed": seed, "size": [w, h], "spp": spp, "on_axis_x": seed % 2 == 0, "fast_vs_exact": d_exact, "fast_vs_closed_form": d_closed}), flush=True)
    print(json.dumps({"views": a.views, "worst_fast_vs_exact": worst_exact, "worst_fast_vs_closed_form": worst_closed,
                      "within_bars": worst_exact <= 1e-5 and worst_closed <= 8e-6}), flush=True)


if __name__ == "__main__":
    main()
