"""The N>1 path of bench.py on CPU: row-band partition + gather to rank 0 over torch.distributed
(gloo here, RCCL on the GPU box).  Bands are filled by the CPU oracle, which stands in for the
per-rank render; what is under test is the sharding and the collective."""
import hashlib
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from raytracer_amd.sharding import BandGather, band_bounds, band_image_rows, band_of, tiled_band

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
META = json.load(open(os.path.join(GOLDEN, "reference_images.json")))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_band_bounds_cover_the_image_without_overlap():
    for height in (4, 8, 270, 1080, 2160, 2161):
        for world in (1, 2, 3, 4, 7, 8):
            b = band_bounds(height, world)
            assert b[0] == 0 and b[-1] == height and all(b[i] <= b[i + 1] for i in range(world))
            sizes = [b[i + 1] - b[i] for i in range(world)]
            assert max(sizes) - min(sizes) <= 1
            assert band_of(height, world, world - 1) == (b[-2], sizes[-1])


def test_tiled_bands_cover_the_image_without_overlap():
    for height in (4, 8, 72, 270, 1080, 2160, 2161, 2163):
        for world in (1, 2, 3, 4, 7, 8):
            for tile_rows in (0, 1, 4, 5, 16):
                if tile_rows and world > -(-height // tile_rows):
                    continue                      # more ranks than tiles
                bands = [tiled_band(height, world, r, tile_rows) for r in range(world)]
                rows = [y for b in bands for y in band_image_rows(b)]
                assert sorted(rows) == list(range(height)), (height, world, tile_rows)
                sizes = [b["row_count"] for b in bands]
                assert max(sizes) - min(sizes) <= max(1, tile_rows)
                for b in bands:                   # the same count the library derives for row_count 0
                    import raytracer_amd as rt
                    if b["tile_rows"] == 0:
                        continue
                    assert rt.band_row_count(height, b["row_begin"], b["tile_rows"], b["tile_stride"]) == b["row_count"]


def _oracle_band(width, height, g):
    """The rank's band cut out of the oracle's image, in band order."""
    full = oracle.trace_primary(width, height)
    return np.ascontiguousarray(full[band_image_rows(g.band)])


def _worker(rank, world, port, width, height, out_path, tile_rows):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = BandGather(width, height, torch.device("cpu"), tile_rows=tile_rows)
        band = _oracle_band(width, height, g)
        assert band.shape[0] == g.rows
        g.local[: g.rows].copy_(torch.from_numpy(band.view(np.int32)))
        image = g.gather()
        if rank == 0:
            np.save(out_path, image.numpy().view(np.uint32))
        else:
            assert image is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size,tile_rows", [(2, (128, 72), 4), (3, (100, 200), 4), (4, (512, 64), 4),
                                                  (3, (100, 200), 16), (2, (128, 72), 0), (3, (100, 200), 0)])
def test_gather_of_bands_rebuilds_reference_image(tmp_path, world, size, tile_rows):
    width, height = size
    out = str(tmp_path / "image.npy")
    mp.spawn(_worker, args=(world, free_port(), width, height, out, tile_rows), nprocs=world, join=True)
    image = np.load(out)
    want = next(e["sha256"] for e in META["images"] if (e["width"], e["height"]) == (width, height))
    assert hashlib.sha256(image.tobytes()).hexdigest() == want


def _worker_overlapped(rank, world, port, width, height, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = BandGather(width, height, torch.device("cpu"))
        band = torch.from_numpy(_oracle_band(width, height, g).view(np.int32))
        frames = 5
        for i in range(frames):
            buf = g.begin(i)
            # frames 0..3 carry a marker, the last frame carries the real band
            buf[: g.rows].copy_(band if i == frames - 1 else torch.full_like(band, i + 1))
            g.submit(i)
        image = g.finish()
        if rank == 0:
            np.save(out_path, image.numpy().view(np.uint32))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size", [(2, (128, 72)), (4, (512, 64))])
def test_overlapped_gather_delivers_the_last_frame(tmp_path, world, size):
    """The double-buffered begin/submit/finish form used by bench.py."""
    width, height = size
    out = str(tmp_path / "image.npy")
    mp.spawn(_worker_overlapped, args=(world, free_port(), width, height, out), nprocs=world, join=True)
    image = np.load(out)
    want = next(e["sha256"] for e in META["images"] if (e["width"], e["height"]) == (width, height))
    assert hashlib.sha256(image.tobytes()).hexdigest() == want


def test_single_rank_renders_straight_into_the_image():
    """Without a process group there is nothing to gather: both band buffers are the image itself
    and the frame loop's calls do no copying."""
    assert not dist.is_initialized()
    g = BandGather(32, 12, torch.device("cpu"))
    assert g.band == dict(row_begin=0, row_count=12, tile_rows=0, tile_stride=0)
    assert not g.collective
    for frame in range(3):
        band = g.begin(frame)
        assert band.data_ptr() == g.image.data_ptr() and band.shape == (12, 32)
        band.fill_(frame + 1)
        g.submit(frame)
    image = g.finish()
    assert image is g.image and bool((image == 3).all())
    g.local.fill_(7)
    assert bool((g.gather() == 7).all())


# ------------------------------------------------------------------ kept columns (fewer bytes on the wire)

def _columns_with_content(full):
    """Columns outside of which every row of `full` is one value (what the library derives from the scene's geometry)."""
    differs = (full != full[:, :1]).any(axis=0) | (full != full[:, -1:]).any(axis=0)
    xs = np.flatnonzero(differs)
    return (int(xs[0]), int(xs[-1]) + 1) if len(xs) else (0, 0)


def _worker_columns(rank, world, port, width, height, out_path, overlapped, synthetic):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = BandGather(width, height, torch.device("cpu"))
        if synthetic:
            # a row gradient with a block of other values against the left edge: the background pixel is the row's last
            full = np.repeat(np.arange(height, dtype=np.uint32)[:, None] * 3 + 7, width, axis=1)
            full[height // 4: height // 2, : width // 3] = np.arange(width // 3, dtype=np.uint32)[None, :] + 1000
        else:
            full = oracle.trace_primary(width, height)
        c0, c1 = _columns_with_content(full)
        narrow = g.set_columns(c0, c1)
        assert narrow == ((c1 - c0 + 1) * 4 <= width * 3)
        band = torch.from_numpy(np.ascontiguousarray(full[band_image_rows(g.band)]).view(np.int32))
        if overlapped:
            for i in range(5):
                buf = g.begin(i)
                buf[: g.rows].copy_(band if i == 4 else torch.full_like(band, i + 1))
                g.submit(i)
            image = g.finish()
        else:
            g.local[: g.rows].copy_(band)
            image = g.gather()
        if rank == 0:
            assert np.array_equal(image.numpy().view(np.uint32), full)
            # switching the narrow form off again gathers whole rows as before
            assert not g.set_columns(0, width)
        else:
            g.set_columns(0, width)
        g.local[: g.rows].copy_(band)
        image = g.gather()
        if rank == 0:
            np.save(out_path, image.numpy().view(np.uint32))
            np.save(out_path + ".want.npy", full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size,overlapped,synthetic", [(2, (128, 72), False, False), (3, (100, 200), True, False),
                                                             (4, (512, 64), True, False), (2, (96, 40), False, True),
                                                             (3, (96, 41), True, True)])
def test_gather_of_kept_columns_rebuilds_the_image(tmp_path, world, size, overlapped, synthetic):
    """BandGather.set_columns: ranks send the columns that hold more than background and one background pixel a row;
    the destination writes the background around them.  Both forms of the gather, a block against the frame's left
    edge (background taken from the last column), and back to whole rows."""
    width, height = size
    out = str(tmp_path / "image.npy")
    mp.spawn(_worker_columns, args=(world, free_port(), width, height, out, overlapped, synthetic), nprocs=world, join=True)
    assert np.array_equal(np.load(out), np.load(out + ".want.npy"))


def _worker_disagreeing(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = BandGather(128, 72, torch.device("cpu"))
        try:
            g.set_columns(32 + 16 * rank, 96)           # every rank with a first column of its own
            outcome = "accepted"
        except RuntimeError as exc:
            outcome = str(exc)
        with open(f"{out_path}.{rank}", "w") as f:
            f.write(outcome)
    finally:
        dist.destroy_process_group()


def test_ranks_that_disagree_on_the_columns_are_told_so_instead_of_waiting_in_the_gather(tmp_path):
    """A rank that packed other columns than its peers would send another number of bytes than the destination
    expects, and the gather would never complete: set_columns compares the columns over the ranks first."""
    out = str(tmp_path / "outcome")
    mp.spawn(_worker_disagreeing, args=(2, free_port(), out), nprocs=2, join=True)
    for rank in range(2):
        text = open(f"{out}.{rank}").read()
        assert "ranks disagree" in text and f"rank {rank}" in text
