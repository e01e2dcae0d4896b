#!/usr/bin/env python3
"""Sweep of random scenes and single-sphere views at seeds the tests do not use: the exact build against the CPU
oracle, every packed word, every averaged colour's bits and the ray count.  Prints one line per mismatch and a
summary; exit status 1 if anything differed.  GPU only.

    python tests/sweep_parity.py [--scenes 400] [--views 400] [--first 10000]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from test_gpu_parity import random_scene, render_device  # noqa: E402


def scene_case(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 33, 60, 130, 300]))
    sph = random_scene(rng, n)
    cam = (float(rng.uniform(-1, 1)), float(rng.uniform(-0.5, 0.5)), float(rng.choice([1.0, 0.3, 2.5, -1.0])))
    w, h = int(rng.choice([24, 40, 61, 97])), int(rng.choice([16, 33, 50]))
    kw = dict(spp=int(rng.choice([1, 3, 6, 70])), max_depth=int(rng.choice([1, 2, 12, 40])), seed=seed, jitter=bool(seed % 3))
    return w, h, dict(spheres=sph, cam=cam, **kw)


def view_case(seed):
    rng = np.random.default_rng(5000 + seed)
    s = np.zeros(1, dtype=rt.SPHERE_DTYPE)
    s["center"][0] = (rng.choice([0.0, rng.uniform(-3, 3)]), rng.choice([0.0, rng.uniform(-2, 2)]), rng.uniform(-8, -0.3))
    s["radius"] = float(rng.choice([0.01, 0.2, 1.0, 3.0, 50.0]) * rng.uniform(0.5, 1.5))
    s["material"] = rt.MAT_NORMAL
    cam = (float(rng.choice([0.0, rng.uniform(-2, 2)])), float(rng.choice([0.0, rng.uniform(-1, 1)])),
           float(rng.choice([1.0, 0.2, 0.6, 4.0, 30.0])))
    w, h = int(rng.choice([33, 64, 129, 257])), int(rng.choice([20, 48, 90]))
    kw = dict(spp=int(rng.choice([1, 5, 16, 70, 200])), max_depth=int(rng.choice([1, 4])), seed=seed, jitter=bool(seed % 4))
    return w, h, dict(spheres=s, cam=cam, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=400)
    ap.add_argument("--views", type=int, default=400)
    ap.add_argument("--first", type=int, default=10000)
    a = ap.parse_args()
    bad = []
    for kind, make, count in (("scene", scene_case, a.scenes), ("view", view_case, a.views)):
        for seed in range(a.first, a.first + count):
            w, h, kw = make(seed)
            want_argb, want_rgba, want_rays, _ = oracle.trace_paths(w, h, threads=8, want_rgba=True, **kw)
            sources = (0, 1, 2) if kind == "scene" else (0,)
            for source in sources:
                got_argb, got_rgba, st = render_device(w, h, count_rays=True, list_source=source, **kw)
                same = (np.array_equal(got_argb, want_argb) and np.array_equal(got_rgba.view(np.uint32), want_rgba.view(np.uint32))
                        and st.rays == want_rays)
                if not same:
                    bad.append((kind, seed, source))
                    print(json.dumps({"mismatch": kind, "seed": seed, "list_source": source, "size": [w, h],
                                      "pixels": int((got_argb != want_argb).sum()), "rays": [int(st.rays), int(want_rays)]}), flush=True)
            if (seed - a.first + 1) % 100 == 0:
                print(f"# {kind}s: {seed - a.first + 1} done, {len(bad)} mismatches so far", flush=True)
    print(json.dumps({"scenes": a.scenes, "views": a.views, "first_seed": a.first, "mismatches": len(bad), "which": bad[:20]}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
