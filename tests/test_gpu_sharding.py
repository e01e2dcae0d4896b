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


@pytest.mark.parametrize("untimed,columns,graphed", [(False, False, False), (True, False, False), (True, True, False), (True, True, True),
                                                     (True, False, True)])
def test_rendered_frames_through_the_two_stream_pipeline_equal_the_single_launch(single_rank_rccl, untimed, columns, graphed):
    """untimed: the benchmark's form of the loop -- the library records no timing events
    (AEHIP_FLAG_NO_TIMING) and begin() / submit() are handed the stream instead of looking it up.
    columns: only the columns the library reports as more than background are gathered.
    graphed: a frame's render (through the entry point that allocates nothing) and the pack of its kept columns
    are captured once per band buffer into a HIP graph and replayed; the collective stays outside the graph."""
    import torch

    device = torch.device("cuda", 0)
    width, height = 1280, 720
    g = BandGather(width, height, device, always_collective=True)
    back = rt.HipRaytracer(None, device=0)
    flags = rt.FLAG_JITTER | (rt.FLAG_NO_TIMING if untimed else 0)
    assert back.setup(width, height, spp=16, max_depth=1, seed=3, flags=flags, pass_=rt.PASS_PATH, **g.band), back.last_error()
    if columns:
        c0, c1 = back.background_columns()
        assert 0 < c0 < c1 < width and (c1 - c0) < width * 0.6
        assert g.set_columns(c0, c1)
    streams = [torch.cuda.Stream(device=device) for _ in range(2)]
    graphs = []
    if graphed:
        for slot in range(2):
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=streams[slot], capture_error_mode="thread_local"):
                back.trace_device_static(g.locals[slot].data_ptr(), None, torch.cuda.current_stream().cuda_stream)
                g.pack(slot, torch.cuda.current_stream().cuda_stream)
            graphs.append(graph)
    for i in range(7):
        side = streams[i & 1]
        torch.cuda.set_stream(side)
        band = g.begin(i, side if untimed else None)
        if graphed:
            band.fill_(0)           # (whatever the buffer held: the replay must render the frame anew)
            graphs[i & 1].replay()
        else:
            back.trace_device(band.data_ptr(), None, side.cuda_stream)
        g.submit(i, side if untimed else None, packed=graphed)
    image = g.finish()
    del graphs
    torch.cuda.synchronize()
    st = back.stats()
    assert (st.kernel_ms == 0.0) if untimed else (st.kernel_ms > 0.0)
    back.close()
    whole, _ = rt.render(width, height, spp=16, max_depth=1, seed=3, jitter=True)
    np.testing.assert_array_equal(image.cpu().numpy().view(np.uint32), whole)


@pytest.mark.parametrize("case", range(12))
def test_outside_the_reported_columns_every_row_is_one_colour(case):
    """aehip_backend_background_columns on the default scene and on random single-sphere views, both builds,
    both passes, jittered or not, every sample walked or not: left of col_begin and from col_end on, every pixel
    of a row equals the row's other background pixels, in the packed image and in the averaged colours."""
    import torch

    rng = np.random.default_rng(4400 + case)
    s = None
    cam = None
    if case >= 3:
        s = np.zeros(1, dtype=rt.SPHERE_DTYPE)
        depth = float(rng.choice([-1.5, -3.0, -8.0, -30.0]))
        s["center"][0] = (rng.uniform(-0.8, 0.8) * abs(depth), rng.uniform(-0.4, 0.4) * abs(depth), depth)
        s["radius"] = float(rng.choice([0.2, 0.6, 1.0]) * rng.uniform(0.7, 1.3))
        s["material"] = int(rng.choice([rt.MAT_NORMAL, rt.MAT_LAMBERT, rt.MAT_METAL]))
        s["albedo"][0] = (0.7, 0.6, 0.5)
        cam = (float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.2, 0.2)), float(rng.choice([1.0, 0.5, 2.0])))
    width, height = [(512, 512), (1920, 1080), (640, 360), (401, 230)][case % 4]
    found = 0
    for fast in (False, True):
        for pass_, spp, flags in ((rt.PASS_PRIMARY, 1, 0), (rt.PASS_PATH, 16, rt.FLAG_JITTER), (rt.PASS_PATH, 64, rt.FLAG_JITTER | rt.FLAG_WALK_ALL)):
            back = rt.HipRaytracer(None, device=0)
            kw = {} if s is None else dict(spheres=s)
            if cam is not None:
                kw["cam"] = cam
            assert back.setup(width, height, spp=spp, max_depth=3, seed=case, flags=flags | (rt.FLAG_FAST if fast else 0), pass_=pass_, **kw), back.last_error()
            c0, c1 = back.background_columns()
            assert 0 <= c0 <= c1 <= width
            argb = torch.zeros((height, width), dtype=torch.int32, device="cuda")
            rgba = torch.zeros((height, width, 4), dtype=torch.float32, device="cuda")
            back.trace_device(argb.data_ptr(), rgba.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            back.close()
            if (c0, c1) == (0, width):
                continue
            found += 1
            bg = 0 if c0 > 0 else width - 1
            outside = torch.cat([torch.arange(0, c0), torch.arange(c1, width)]).to("cuda")
            assert bool((argb[:, outside] == argb[:, bg:bg + 1]).all())
            assert bool((rgba[:, outside] == rgba[:, bg:bg + 1]).all())
    if s is None:
        assert found == 6       # the reference's scene has its box at every size


def test_every_band_of_a_frame_reports_the_same_columns():
    from raytracer_amd.sharding import tiled_band

    whole = rt.HipRaytracer(None, device=0)
    assert whole.setup(3840, 2160, spp=64, flags=rt.FLAG_JITTER | rt.FLAG_FAST, pass_=rt.PASS_PATH)
    want = whole.background_columns()
    whole.close()
    assert want == (1152, 2688)         # the reference's sphere: 1,528 pixels across at 4K, centred
    for world, rank in ((2, 1), (8, 0), (8, 7), (540, 0), (540, 539)):      # (one tile of 4 rows a rank: sky only at either end)
        for flags in (rt.FLAG_JITTER | rt.FLAG_FAST, 0):
            back = rt.HipRaytracer(None, device=0)
            assert back.setup(3840, 2160, spp=64, flags=flags, pass_=rt.PASS_PATH, **tiled_band(2160, world, rank, 4))
            assert back.background_columns() == want
            back.close()


@pytest.mark.parametrize("shape", [(64, 3840, 1152, 2688), (37, 401, 0, 130), (9, 258, 100, 258), (5, 17, 3, 4), (270, 1922, 16, 1900)])
def test_pack_and_expand_kernels_against_their_definition(shape):
    """aehip_band_pack_columns / aehip_band_expand_columns on random words: kept columns at either edge, widths
    that are not multiples of four (scalar stores), rows taken in a shuffled order."""
    import torch

    rows, width, c0, c1 = shape
    lib = rt.load()
    gen = torch.Generator(device="cpu").manual_seed(rows * 7919 + width)
    band = torch.randint(-2**31, 2**31 - 1, (rows, width), dtype=torch.int32, generator=gen).cuda()
    bg = 0 if c0 > 0 else width - 1
    kept = c1 - c0 + 1
    packed = torch.zeros((rows, kept), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.aehip_band_pack_columns(band.data_ptr(), rows, width, c0, c1, bg, packed.data_ptr(), stream) == 0
    want = torch.cat([band[:, bg:bg + 1], band[:, c0:c1]], dim=1)
    assert torch.equal(packed, want)
    order = torch.randperm(rows, generator=gen).cuda()
    image = torch.zeros((rows, width), dtype=torch.int32, device="cuda")
    assert lib.aehip_band_expand_columns(packed.data_ptr(), rows, order.data_ptr(), rows, width, c0, c1, image.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    rebuilt = band[order][:, bg:bg + 1].expand(rows, width).clone()
    rebuilt[:, c0:c1] = band[order][:, c0:c1]
    assert torch.equal(image, rebuilt)
    assert lib.aehip_band_pack_columns(band.data_ptr(), rows, width, c1, c0, bg, packed.data_ptr(), stream) == -2
    assert lib.aehip_band_expand_columns(None, rows, order.data_ptr(), rows, width, c0, c1, image.data_ptr(), stream) == -2
    # a background column among the kept ones would send a pixel of the shape as the row's background
    assert lib.aehip_band_pack_columns(band.data_ptr(), rows, width, c0, c1, c0, packed.data_ptr(), stream) == -2
    assert b"inside the kept columns" in lib.aehip_last_error()
    # row indices outside the packed rows are never read through: those image rows keep what they held
    wild = order.clone()
    wild[0], wild[rows - 1] = -1, rows + 5
    image.fill_(-7)
    assert lib.aehip_band_expand_columns(packed.data_ptr(), rows, wild.data_ptr(), rows, width, c0, c1, image.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    assert bool((image[0] == -7).all()) and bool((image[rows - 1] == -7).all())
    assert torch.equal(image[1:rows - 1], rebuilt[1:rows - 1])
    assert lib.aehip_band_expand_columns(packed.data_ptr(), 0, order.data_ptr(), rows, width, c0, c1, image.data_ptr(), stream) == -2
