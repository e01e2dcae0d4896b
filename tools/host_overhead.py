"""Host time per frame of the multi-rank loop of bench.py (launch, event records, asynchronous
RCCL gather, wait, row reorder), measured with a one-rank RCCL group on one GPU and a frame that
renders in microseconds, so that what is left is the host's work.

    python tools/host_overhead.py [--frames 300]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--timed", action="store_true", help="the loop as it was before: library timing events, streams looked up")
    ap.add_argument("--graph", action="store_true", help="bench.py's default: a frame's render and pack replayed from a HIP graph")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd.sharding import BandGather

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    side = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(side)
    W, H = 3840, 2160
    # the multi-rank code path on the single rank, on a frame as large as one rank's share of 8 (4 MB gathered and
    # put in row order a frame), so that the GPU is done long before the host
    g = BandGather(W, 270, device, always_collective=True)
    back = rt.HipRaytracer(None, device=0)
    # --timed: the library's own timing events around every launch, as callers without AEHIP_FLAG_NO_TIMING get them
    assert back.setup(W, 270, spp=1, max_depth=1, seed=1, flags=rt.FLAG_FAST | (0 if a.timed else rt.FLAG_NO_TIMING), pass_=rt.PASS_PATH)
    if not a.timed:
        g.set_columns(*back.background_columns())       # bench.py's form: the sphere's columns and a background pixel a row
    sides = [side, torch.cuda.Stream(device=device)]
    handles = [s.cuda_stream for s in sides]

    graphs = None
    if a.graph:
        assert not a.timed, "--graph needs the untimed request (AEHIP_FLAG_NO_TIMING)"
        back.trace_device_static(g.locals[0].data_ptr(), None, handles[0])
        graphs = []
        for slot in range(2):
            gr = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(gr, stream=sides[slot], capture_error_mode="thread_local"):
                back.trace_device_static(g.locals[slot].data_ptr(), None, torch.cuda.current_stream().cuda_stream)
                g.pack(slot, torch.cuda.current_stream().cuda_stream)
            graphs.append(gr)

    def loop(n):
        # bench.py's timed loop for several ranks: frames on two streams in turn, gather submitted behind each
        for i in range(n):
            side = sides[i & 1]
            torch.cuda.set_stream(side)
            band = g.begin(i, None if a.timed else side)
            if graphs is not None:
                graphs[i & 1].replay()
            else:
                back.trace_device(band.data_ptr(), None, handles[i & 1])
            g.submit(i, None if a.timed else side, packed=graphs is not None)
        torch.cuda.set_stream(sides[0])
        g.finish()

    loop(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(a.frames)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"host time per frame {host / a.frames * 1e3:.3f} ms; with the GPU drained {total / a.frames * 1e3:.3f} ms "
          f"({a.frames} frames of 3840x270, 1 spp)")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
