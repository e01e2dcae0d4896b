"""Frames of one rank's share of the 4K headline frame (tiles dealt to 1, 2, 4, 8 ranks), queued on
one stream against consecutive frames on two streams in turn: what the overlap of one frame's
last workgroups with the next frame's first is worth.

    python tools/frame_overlap.py
"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd.sharding import tiled_band
W, H = 3840, 2160
dev = torch.device("cuda", 0)
for mode in ("fast", "exact"):
    for world in (1, 2, 4, 8):
        band = tiled_band(H, world, 0, 4)
        flags = rt.FLAG_JITTER | (rt.FLAG_FAST if mode == "fast" else 0)
        back = rt.HipRaytracer(None, device=0)
        assert back.setup(W, H, spp=1024, max_depth=8, seed=1, flags=flags, pass_=rt.PASS_PATH, **band)
        bufs = [torch.empty((band["row_count"], W), dtype=torch.int32, device=dev) for _ in range(2)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        res = []
        for nstreams in (1, 2):
            n = 60 if mode == "fast" else 24
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    back.trace_device(bufs[i & 1].data_ptr(), None, streams[i % nstreams].cuda_stream)
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) / n * 1e3
            res.append(t)
        print(f"{mode} world {world}: one stream {res[0]:.4f} ms/frame, two streams {res[1]:.4f} ms/frame", flush=True)
        back.close()
