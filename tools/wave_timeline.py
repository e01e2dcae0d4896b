"""Per-wave timeline of the fast build's 4K headline launch, from the measurement build of the
library (make timeline -> tools/bin/libhip_raytracer_timeline.so): how many waves are resident
over the launch, how long the waves of each kind live, which waves end last.

    make timeline && RAYTRACER_AMD_LIB=$PWD/tools/bin/libhip_raytracer_timeline.so python tools/wave_timeline.py [spp [bands rank tile_rows]]

With bands: the launch of one rank's band of the frame (tiles of tile_rows rows dealt round-robin).
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt

lib = rt.load()
if not hasattr(lib, "aehip_wave_timeline"):
    sys.exit("this library has no timeline: build it with `make timeline` and point RAYTRACER_AMD_LIB at it")
W, H = (int(v) for v in os.environ.get("TIMELINE_SIZE", "3840x2160").split("x"))      # TIMELINE_SIZE=1920x1080: another frame size
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
back = rt.HipRaytracer(None)
band = {}
if len(sys.argv) > 4:
    from raytracer_amd.sharding import tiled_band
    band = tiled_band(H, int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
assert back.setup(W, H, spp, 8, 1, rt.FLAG_JITTER | rt.FLAG_FAST, rt.PASS_PATH, spheres=None, **band), back.last_error()
argb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for _ in range(int(os.environ.get("TIMELINE_LAUNCHES", "4"))):         # (the last launch's stamps are read; more launches, warmer clocks)
    back.trace_device(argb.data_ptr(), None, stream)
torch.cuda.synchronize()
cap = 1 << 18
buf = (ctypes.c_ulonglong * (3 * cap))()
n = lib.aehip_wave_timeline(buf, cap)
assert n > 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 3)[:n].astype(np.int64)
t0 = a[:, 0].min()
start, end, kind, clocks = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, a[:, 2] & 0xff, a[:, 2] >> 8        # microseconds
dur = end - start
span = end.max()
print(f"{n} waves, launch {span:.1f} us (the stamps themselves cost a few per cent)")
long_lived = dur > 20.0
if long_lived.any():
    ghz = clocks[long_lived] / dur[long_lived] / 1e3
    print(f"shader clock over the waves that live more than 20 us: median {np.median(ghz):.3f} GHz, 5 % {np.percentile(ghz, 5):.3f}, 95 % {np.percentile(ghz, 95):.3f}")
for k, name in ((1, "listed pixels"), (0, "rectangle"), (2, "background")):
    m = kind == k
    if m.any():
        d = dur[m]
        print(f"{name:14s} {int(m.sum()):7d} waves, life us: mean {d.mean():.1f} median {np.median(d):.1f} 90% {np.percentile(d, 90):.1f} max {d.max():.1f}; last start {start[m].max():.1f}, last end {end[m].max():.1f}")
bins = 50
edges = np.linspace(0.0, span, bins + 1)
for label, sel in (("all waves", np.ones(n, bool)), ("waves over 5 us", dur > 5.0), ("listed pixels", kind == 1)):
    conc = [np.clip(np.minimum(end[sel], edges[i + 1]) - np.maximum(start[sel], edges[i]), 0, None).sum() / (edges[i + 1] - edges[i]) for i in range(bins)]
    print(f"{label}, resident per 2 % of the launch: " + " ".join(f"{c:.0f}" for c in conc))
print("life histogram (us) 0-2-5-20-40-60-80-100-120-150-200+:", np.histogram(dur, bins=[0, 2, 5, 20, 40, 60, 80, 100, 120, 150, 200, 1e9])[0])
for i in np.argsort(-end)[:12]:
    print(f"  ends {end[i]:.1f} us: started {start[i]:.1f}, workgroup {i // 4}, wave {i % 4}, {('rectangle', 'listed pixels', 'background')[kind[i]]}")
back.close()
