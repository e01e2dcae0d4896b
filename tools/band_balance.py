"""Kernel time of each rank's band of the benchmark frame, measured one after the other on one GPU:
shows how evenly a split divides the work (the default scene's sphere sits in the middle rows, so
contiguous bands are uneven and tiles dealt round-robin are not)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd.sharding import tiled_band

ap = argparse.ArgumentParser()
ap.add_argument("--bands", type=int, default=8)
ap.add_argument("--mode", default="fast")
ap.add_argument("--spp", type=int, default=1024)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--scene", default="default", help="default | showcase | divergence | stress:<count>")
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--tile-rows", type=int, default=0, help="0 = contiguous bands, T = tiles of T rows dealt round-robin")
a = ap.parse_args()
W, H = 3840, 2160
from raytracer_amd import scenes
if a.scene == "default":
    spheres = None
elif a.scene.startswith("stress:"):
    spheres = scenes.stress_scene(int(a.scene.split(":")[1]), 42)
else:
    spheres = scenes.SCENES[a.scene]()
flags = rt.FLAG_JITTER | (rt.FLAG_FAST if a.mode == "fast" else 0)
buf = torch.zeros((H, W), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
times = []
for r in range(a.bands):
    back = rt.HipRaytracer(None)
    band = tiled_band(H, a.bands, r, a.tile_rows)
    assert back.setup(W, H, spp=a.spp, max_depth=a.depth, seed=1, flags=flags, pass_=rt.PASS_PATH, spheres=spheres, **band)
    # launches back to back, as a render loop issues them (an idle GPU between launches clocks down)
    for _ in range(3):
        back.trace_device(buf.data_ptr(), None, stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        back.trace_device(buf.data_ptr(), None, stream)
    e1.record()
    torch.cuda.synchronize()
    best = e0.elapsed_time(e1) / a.reps
    back.close()
    times.append(best)
    print(f"band {r}: {band} {best:.4f} ms", flush=True)
print(f"sum {sum(times):.4f} ms  max {max(times):.4f} ms  balance (mean/max) {sum(times)/len(times)/max(times):.3f}")
