#!/usr/bin/env python3
"""Secondary measurements on one GPU: the shape-list scenes of the extended renderer.

    python tools/bench_scenes.py [--mode fast|exact] [--case NAME ...]

Prints one JSON line per case: kernel time (HIP events inside the library), paths/s, rays/s
(device-counted ray segments) and the algorithmic FP32 rate of the intersection work,
rays x spheres x 18 flop (the reference quadratic with the per-ray terms hoisted) -- material,
RNG and shading work is not counted, so the figure is a lower bound.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (bench.py's CONFIG_CASES: the benchmark's other configurations at the sizes it names)
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

CASES = {
    # name: (width, height, spp, depth, scene factory)
    "default_1080p_64spp_d8": (1920, 1080, 64, 8, scenes.default_scene),
    "showcase_1080p_64spp_d8": (1920, 1080, 64, 8, scenes.showcase_scene),
    "stress10k_640x360_1spp_d4": (640, 360, 1, 4, lambda: scenes.stress_scene(10000, 42)),
    "stress10k_640x360_16spp_d4": (640, 360, 16, 4, lambda: scenes.stress_scene(10000, 42)),
    "stress10k_1080p_4spp_d4": (1920, 1080, 4, 4, lambda: scenes.stress_scene(10000, 42)),
    "stress10k_1080p_2spp_d1": (1920, 1080, 2, 1, lambda: scenes.stress_scene(10000, 42)),
    "stress2k_1080p_1spp_d1": (1920, 1080, 1, 1, lambda: scenes.stress_scene(2000, 42)),
    "stress2k_1080p_1spp_d4": (1920, 1080, 1, 4, lambda: scenes.stress_scene(2000, 42)),
    "stress2k_1080p_16spp_d4": (1920, 1080, 16, 4, lambda: scenes.stress_scene(2000, 42)),
    "divergence_1080p_16spp_d50": (1920, 1080, 16, 50, lambda: scenes.divergence_scene(300, 7)),
    "divergence600_1080p_8spp_d50": (1920, 1080, 8, 50, lambda: scenes.divergence_scene(600, 7)),
    "divergence1200_1080p_4spp_d50": (1920, 1080, 4, 50, lambda: scenes.divergence_scene(1200, 7)),
}
SEEDS = {}
_SCENES = {"default": scenes.default_scene, "stress10k": lambda: scenes.stress_scene(10000, 42), "divergence": lambda: scenes.divergence_scene(300, 7)}
for _name, (_w, _h, _spp, _depth, _seed, _scene) in bench.CONFIG_CASES.items():     # the cases bench.py times, same requests
    CASES[_name] = (_w, _h, _spp, _depth, _SCENES[_scene])
    SEEDS[_name] = _seed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["fast", "exact"], default="fast")
    ap.add_argument("--case", nargs="*", default=list(CASES))
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--list", type=int, default=0, help="shape-list source: 0 auto, 1 LDS, 2 scalar cache")
    ap.add_argument("--no-regen", action="store_true", help="disable path regeneration (lane utilisation study)")
    args = ap.parse_args()
    import torch

    for name in args.case:
        w, h, spp, depth, make = CASES[name]
        sph = make()
        back = rt.HipRaytracer(None)
        flags = rt.FLAG_JITTER | rt.FLAG_COUNT_RAYS | (rt.FLAG_FAST if args.mode == "fast" else 0) \
            | (rt.FLAG_NO_REGEN if args.no_regen else 0)
        assert back.setup(w, h, spp=spp, max_depth=depth, seed=SEEDS.get(name, 1), flags=flags, pass_=rt.PASS_PATH, spheres=sph, list_source=args.list), back.last_error()
        argb = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        best = None
        for _ in range(args.reps):
            back.trace_device(argb.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            st = back.stats()
            if best is None or st.kernel_ms < best[0]:
                best = (st.kernel_ms, st.rays, st.paths, st.vgprs, st.lds_bytes, st.lane_slots)
        ms, rays, paths, vgprs, lds, slots = best
        flop = rays * len(sph) * 18.0
        print(json.dumps({
            "case": name, "arithmetic": args.mode, "list": args.list, "regeneration": not args.no_regen,
            "lane_utilisation": rays / slots if slots else None, "spheres": int(len(sph)), "kernel_ms": ms,
            "mpaths_per_s": paths / ms / 1e3, "grays_per_s": rays / ms / 1e6, "rays_per_path": rays / paths,
            "sphere_tests_per_s": rays * len(sph) / (ms * 1e-3),
            "intersect_tflops": flop / (ms * 1e-3) / 1e12, "vgprs": vgprs, "lds_bytes": lds,
        }), flush=True)
        back.close()


if __name__ == "__main__":
    main()
