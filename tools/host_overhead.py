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
    g = BandGather(W, H, device)
    # force the multi-rank code path on the single rank: received bands stacked, rows reordered
    g.world = 1
    recv = torch.zeros((1, g.max_rows, W), dtype=torch.int32, device=device)
    image = g.image
    order = torch.arange(H, dtype=torch.int64, device=device)
    back = rt.HipRaytracer(None, device=0)
    assert back.setup(W, H, spp=1, max_depth=1, seed=1, flags=rt.FLAG_FAST, pass_=rt.PASS_PATH, row_begin=0, row_count=270)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.frames)]

    def loop(n):
        pending = None
        for i in range(n):
            band = g.locals[i & 1]
            ev[i][0].record()
            back.trace_device(band.data_ptr(), None, side.cuda_stream)
            ev[i][1].record()
            if pending is not None:
                pending.wait()
                torch.index_select(recv.view(g.max_rows, W), 0, order, out=image)
            pending = dist.gather(band, [recv[0]], dst=0, async_op=True)
        pending.wait()

    loop(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(a.frames)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"host time per frame {host / a.frames * 1e3:.3f} ms; with the GPU drained {total / a.frames * 1e3:.3f} ms "
          f"({a.frames} frames, 4K image, 270-row band, 1 spp)")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
