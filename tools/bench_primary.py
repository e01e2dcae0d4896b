#!/usr/bin/env python3
"""Times the reference-pinned primary pass (1 centred ray per pixel) at the reference's sizes.
Prints one JSON line per (size, arithmetic): wall time per back-to-back launch, Mpaths/s and the
store bandwidth 4 B/pixel implies.  Run under rocprofv3 --kernel-trace --stats for the kernel's
own duration."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt  # noqa: E402


def main():
    import torch

    side = torch.cuda.Stream()          # a real stream: the library waits after every launch on the null stream
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    sizes = [(512, 512), (1920, 1080), (3840, 2160), (3838, 2160)]      # the last: one pixel per lane (width not a multiple of 4)
    if "--only-4k" in sys.argv:
        sizes = [(3840, 2160)]      # counter passes: one launch shape per kernel, so per-kernel means mean something
    for (w, h) in sizes:
        for fast in (False, True):
            back = rt.HipRaytracer(None)
            assert back.setup(w, h, flags=rt.FLAG_FAST if fast else 0, pass_=rt.PASS_PRIMARY)
            argb = torch.zeros((h, w), dtype=torch.int32, device="cuda")
            for _ in range(10):
                back.trace_device(argb.data_ptr(), None, stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 200
            e0.record()
            for _ in range(reps):
                back.trace_device(argb.data_ptr(), None, stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print(json.dumps({"size": [w, h], "arithmetic": "fast" if fast else "exact", "ms_per_launch_wall": ms,
                              "mpaths_per_s": w * h / ms / 1e3, "store_gb_per_s": w * h * 4 / (ms * 1e-3) / 1e9}), flush=True)
            back.close()


if __name__ == "__main__":
    main()
