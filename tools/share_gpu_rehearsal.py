"""Two (or more) ranks that share ONE GPU, gathered over gloo: a rehearsal of BandGather's kept-columns gather on
device tensors between processes, for boxes with fewer GPUs than ranks (RCCL needs a GPU a rank).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
        tools/share_gpu_rehearsal.py [--frames 4] [--whole-rows] [--overlapped]

What it has shown on this image (ROCm 7.2, torch 2.10): both forms complete and verify in most runs (two ranks, 4 frames,
narrow buffers, blocking and overlapped), while the same overlapped exchange inside a longer loop (bench.py with two
ranks, a few dozen frames) ended with every rank inside Work.wait() of the same frame's gather and no rank ahead or
behind -- the ranks agreed on shapes and call counts (BandGather.set_columns checks the former), so the stall is inside
gloo's staging of device tensors, not in what the ranks asked for.  bench.py's shared-GPU rehearsal therefore hands
gloo host tensors; RCCL, the product path, has its own stream-ordered form of the gather.

Every rank prints what it is about to contribute (shape of the packed band, the receive buffers' shapes on the
destination, whether the narrow form is on) before each collective, dumps its stacks and exits if a call has not
returned after --watchdog seconds, and rank 0 checks the gathered frame against the frame it renders alone.
"""
import argparse
import faulthandler
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--whole-rows", action="store_true")
    ap.add_argument("--overlapped", action="store_true", help="begin / submit / finish instead of the blocking gather()")
    ap.add_argument("--watchdog", type=int, default=60)
    a = ap.parse_args()
    faulthandler.dump_traceback_later(a.watchdog, exit=True)
    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd.sharding import BandGather

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 3840, 2160
    g = BandGather(W, H, device)
    back = rt.HipRaytracer(None, device=0)
    flags = rt.FLAG_JITTER | rt.FLAG_FAST
    assert back.setup(W, H, spp=64, max_depth=8, seed=1, flags=flags, pass_=rt.PASS_PATH, **g.band), back.last_error()
    c0, c1 = back.background_columns()
    narrow = False if a.whole_rows else g.set_columns(c0, c1)

    def say(what):
        print(f"[rank {rank}] {what}", flush=True)

    say(f"columns {c0}..{c1} narrow={narrow} band rows {g.rows} of max {g.max_rows}; contributes {tuple(g._pack(0).shape)}"
        + (f"; receives {[tuple(p.shape) for p in g._receive_parts(0)]}" if rank == 0 else ""))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.frames):
        if a.overlapped:
            band = g.begin(i)
            back.trace_device(band.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            say(f"frame {i}: submit")
            g.submit(i)
        else:
            back.trace_device(g.local.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            say(f"frame {i}: gather")
            g.gather()
        say(f"frame {i}: returned")
    image = g.finish()
    torch.cuda.synchronize()
    say(f"{a.frames} frames in {(time.perf_counter() - t0) * 1e3:.1f} ms")
    if rank == 0:
        whole = rt.HipRaytracer(None, device=0)
        assert whole.setup(W, H, spp=64, max_depth=8, seed=1, flags=flags, pass_=rt.PASS_PATH)
        ref = torch.empty((H, W), dtype=torch.int32, device=device)
        whole.trace_device(ref.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        say(f"gathered frame equals the single-GPU frame: {bool(torch.equal(ref, image))}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
