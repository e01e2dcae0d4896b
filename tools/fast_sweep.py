#!/usr/bin/env python3
"""Sweep of random single-sphere views: the fast build against the exact build, largest difference of
the averaged colour over every pixel of every view (the bar is 1e-5).  The views come from the two
generators of tests/test_gpu_parity.py (small frames with spheres anywhere around the camera;
log-uniform distance and distance / radius), at seeds the tests do not use.  GPU only.

    python tools/fast_sweep.py [--views 2000] [--first 1000]
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


def around_the_camera(seed):
    rng = np.random.default_rng(9100 + seed)
    s = np.zeros(1, dtype=rt.SPHERE_DTYPE)
    depth = float(rng.choice([-1.2, -2.0, -4.0, -12.0, -40.0, 0.5, 3.0]))
    spread = abs(depth) * float(rng.choice([0.0, 0.5, 1.2, 2.5]))
    s["center"][0] = (rng.uniform(-spread, spread), rng.uniform(-spread, spread) * 0.6, depth)
    s["radius"] = float(rng.choice([0.05, 0.3, 1.0, 3.0]) * rng.uniform(0.7, 1.3))
    s["material"] = rt.MAT_NORMAL
    cam = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.3, 0.3)), float(rng.choice([1.0, 0.4, 2.5])))
    w, h = [(384, 216), (401, 230), (258, 143), (640, 64)][seed % 4]
    kw = dict(spheres=s, cam=cam, spp=int(rng.choice([16, 64])), max_depth=2, seed=seed, jitter=bool(seed % 5))
    if seed % 3 == 0:
        r0 = int(rng.integers(0, h - 8))
        kw.update(row_begin=r0, row_count=int(rng.integers(1, h - r0 + 1)))
    return w, h, kw


def distances_and_radii(seed):
    rng = np.random.default_rng(1000 + 37 * seed)
    cam = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.3, 0.3), float(rng.choice([1.0, 0.4, 2.5]))])
    ratio = float(np.exp(rng.uniform(np.log(1.05), np.log(3000.0))))
    dist = float(np.exp(rng.uniform(np.log(0.5), np.log(60.0))))
    toward = np.array([rng.uniform(-0.9, 0.9), rng.uniform(-0.5, 0.5), -1.0])
    toward /= np.linalg.norm(toward)
    s = np.zeros(1, dtype=rt.SPHERE_DTYPE)
    s["center"][0] = tuple(cam + toward * dist)
    s["radius"] = dist / ratio
    s["material"] = rt.MAT_NORMAL
    w, h = [(384, 216), (401, 230), (640, 360)][seed % 3]
    kw = dict(spheres=s, cam=tuple(float(v) for v in cam), spp=int(rng.choice([1, 4, 64])), max_depth=2, seed=seed, jitter=bool(seed % 2))
    return w, h, kw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=2000, help="views per generator")
    ap.add_argument("--first", type=int, default=1000, help="first seed (the tests use 0..47 and 0..23)")
    a = ap.parse_args()
    for name, gen in (("around_the_camera", around_the_camera), ("distances_and_radii", distances_and_radii)):
        worst, worst_seed, over, unwritten, hist = 0.0, None, [], [], np.zeros(4, dtype=np.int64)
        for seed in range(a.first, a.first + a.views):
            w, h, kw = gen(seed)
            _, fast, _ = render_device(w, h, fast=True, **kw)
            _, exact, _ = render_device(w, h, **kw)
            err = float(np.abs(fast[..., :3] - exact[..., :3]).max())
            if float(fast[..., 3].min()) != 1.0:
                unwritten.append(seed)
            hist[int(np.searchsorted([2e-6, 5e-6, 1e-5], err, side="right"))] += 1
            if err > worst:
                worst, worst_seed = err, seed
            if err > 1e-5:
                over.append((seed, err))
            if (seed - a.first) % 250 == 249:
                print(f"# {name}: {seed - a.first + 1} views, worst so far {worst:.3e} (seed {worst_seed})", flush=True)
        print(json.dumps({"generator": name, "views": a.views, "first_seed": a.first, "worst": worst, "worst_seed": worst_seed,
                          "views_by_worst_pixel": {"<=2e-6": int(hist[0]), "<=5e-6": int(hist[1]), "<=1e-5": int(hist[2]), ">1e-5": int(hist[3])},
                          "over_1e-5": over, "views_with_unwritten_pixels": unwritten}), flush=True)


if __name__ == "__main__":
    main()
