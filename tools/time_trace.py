"""Wall time of the host-buffer call (setup once, trace() repeated): render + device->host copy
+ copy into the caller's framebuffer, as the reference's main loop would see it.

    python tools/time_trace.py [--width 3840 --height 2160 --spp 1024 --mode fast|exact --reps 20]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--mode", default="fast", choices=["fast", "exact"])
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    fb = np.zeros((a.height, a.width), dtype=np.uint32)
    r = rt.HipRaytracer(fb, 0)
    flags = rt.FLAG_JITTER | (rt.FLAG_FAST if a.mode == "fast" else 0)
    if not r.setup(a.width, a.height, a.spp, 1, 1, flags, rt.PASS_PATH):
        raise SystemExit("setup failed: " + r.last_error())
    for _ in range(3):
        r.trace()
    times = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        r.trace()
        times.append((time.perf_counter() - t0) * 1e3)
    st = r.stats()
    times.sort()
    print(f"{a.mode} {a.width}x{a.height} spp {a.spp}: trace() median {times[len(times) // 2]:.3f} ms, "
          f"min {times[0]:.3f} ms; kernels {st.kernel_ms:.3f} ms, copy tail {st.copy_ms:.3f} ms, "
          f"checksum {int(fb.astype(np.uint64).sum())}")
    r.close()


if __name__ == "__main__":
    main()
