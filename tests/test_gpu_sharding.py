"""The frame pipeline of the multi-rank loop (two band buffers, consecutive frames on two streams,
asynchronous RCCL gather, row reorder) on one GPU: a one-rank RCCL group with the collective path
forced, frames told apart by their contents."""
import os
import socket

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.sharding import BandGather

pytestmark = pytest.mark.gpu


@pytest.fixture()
def single_rank_rccl():
    import torch
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        yield dist
    finally:
        dist.destroy_process_group()
        torch.cuda.set_stream(torch.cuda.default_stream())


class CheckedGather(BandGather):
    """Keeps a sample of the image after every assembly, taken on the stream that assembled it."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.seen = []

    def _assemble(self, slot=0):
        out = super()._assemble(slot)
        self.seen.append(self.image[::37, ::41].clone())
        return out


@pytest.mark.parametrize("n_streams", [1, 2])
def test_frames_arrive_whole_and_in_order_with_frames_on_alternating_streams(single_rank_rccl, n_streams):
    import torch

    device = torch.device("cuda", 0)
    width, height, frames = 1024, 600, 24
    g = CheckedGather(width, height, device, always_collective=True)
    assert g.collective and g._row_source is not None
    streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
    for i in range(frames):
        torch.cuda.set_stream(streams[i % n_streams])
        band = g.begin(i)
        if i % 3 == 0:
            torch.cuda._sleep(2_000_000)        # uneven frame times: every overlap pattern comes up
        band.fill_(i + 1)                       # the "render": the whole band carries the frame's number
        if i % 5 == 1:
            torch.cuda._sleep(1_000_000)
        g.submit(i)
    image = g.finish()
    torch.cuda.synchronize()
    assert len(g.seen) == frames
    for i, sample in enumerate(g.seen):
        assert bool((sample == i + 1).all()), f"frame {i} arrived mixed with another frame"
    assert bool((image == frames).all())


@pytest.mark.parametrize("untimed", [False, True])
def test_rendered_frames_through_the_two_stream_pipeline_equal_the_single_launch(single_rank_rccl, untimed):
    """untimed: the benchmark's form of the loop -- the library records no timing events
    (AEHIP_FLAG_NO_TIMING) and begin() / submit() are handed the stream instead of looking it up."""
    import torch

    device = torch.device("cuda", 0)
    width, height = 1280, 720
    g = BandGather(width, height, device, always_collective=True)
    back = rt.HipRaytracer(None, device=0)
    flags = rt.FLAG_JITTER | (rt.FLAG_NO_TIMING if untimed else 0)
    assert back.setup(width, height, spp=16, max_depth=1, seed=3, flags=flags, pass_=rt.PASS_PATH, **g.band), back.last_error()
    streams = [torch.cuda.Stream(device=device) for _ in range(2)]
    for i in range(7):
        side = streams[i & 1]
        torch.cuda.set_stream(side)
        band = g.begin(i, side if untimed else None)
        back.trace_device(band.data_ptr(), None, side.cuda_stream)
        g.submit(i, side if untimed else None)
    image = g.finish()
    torch.cuda.synchronize()
    st = back.stats()
    assert (st.kernel_ms == 0.0) if untimed else (st.kernel_ms > 0.0)
    back.close()
    whole, _ = rt.render(width, height, spp=16, max_depth=1, seed=3, jitter=True)
    np.testing.assert_array_equal(image.cpu().numpy().view(np.uint32), whole)
