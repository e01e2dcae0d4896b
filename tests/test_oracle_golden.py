"""The CPU oracle (oracle/rt_oracle.c) against fixtures produced by the reference's own code.

Fixtures: tests/golden/reference_images.json (sha256 of raw ARGB images rendered by the
reference's software_raytracer::trace_tile), ref_image_*.npy (small full images) and
ref_floats_*.npy (per-pixel floats from the reference's ray / sphere / remap / lerp / pack
functions).  All made by tests/golden/make_golden.py.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
META = json.load(open(os.path.join(GOLDEN, "reference_images.json")))


@pytest.mark.parametrize("entry", META["images"], ids=lambda e: f"{e['width']}x{e['height']}")
def test_primary_image_hash(entry):
    w, h = entry["width"], entry["height"]
    img = oracle.trace_primary(w, h)
    assert hashlib.sha256(img.tobytes()).hexdigest() == entry["sha256"]
    if "file" in entry:
        gold = np.load(os.path.join(GOLDEN, entry["file"]))
        np.testing.assert_array_equal(img, gold)


@pytest.mark.parametrize("entry", META["images"][:2] + META["images"][3:], ids=lambda e: f"{e['width']}x{e['height']}")
def test_extended_renderer_degenerates_to_reference(entry):
    """Default scene, one sample at the pixel centre, depth 1: the extended renderer must
    reproduce the reference image bit for bit."""
    w, h = entry["width"], entry["height"]
    img, _, rays, _ = oracle.trace_paths(w, h, spp=1, max_depth=1, jitter=False, threads=4)
    assert hashlib.sha256(img.tobytes()).hexdigest() == entry["sha256"]
    assert rays == w * h


def test_scheduler_runs_match_probe():
    """The reference's real multithreaded runs gave the same bytes as the single-threaded probe."""
    by_size = {(e["width"], e["height"]): e["sha256"] for e in META["images"]}
    for run in META["scheduler_runs"]:
        assert by_size[(run["width"], run["height"])] == run["sha256"]


@pytest.mark.parametrize("size", [(512, 512), (1920, 1080), (100, 200)])
def test_float_level_records(size):
    w, h = size
    rec = np.load(os.path.join(GOLDEN, f"ref_floats_{w}x{h}.npy"))
    argb, rgba = oracle.trace_primary(w, h, want_rgba=True)
    xs = rec[:, 0].astype(np.int64)
    ys = rec[:, 1].astype(np.int64)
    # colour handed to the pack: bit-exact
    np.testing.assert_array_equal(rgba[ys, xs, :3].view(np.uint32), rec[:, 13:16].view(np.uint32))
    np.testing.assert_array_equal(argb[ys, xs], rec[:, 16].view(np.uint32))


def test_band_rendering_matches_full_image():
    full = oracle.trace_primary(128, 72)
    band = oracle.trace_primary(128, 72, row_begin=20, row_count=17)
    np.testing.assert_array_equal(band, full[20:37])
    p_full, _, _, _ = oracle.trace_paths(128, 72, spp=4, max_depth=3, jitter=True, seed=9)
    p_band, _, _, _ = oracle.trace_paths(128, 72, spp=4, max_depth=3, jitter=True, seed=9, row_begin=20, row_count=17)
    np.testing.assert_array_equal(p_band, p_full[20:37])


def test_threads_do_not_change_result():
    a, _, ra, _ = oracle.trace_paths(96, 64, spp=3, max_depth=4, jitter=True, seed=5, threads=1)
    b, _, rb, _ = oracle.trace_paths(96, 64, spp=3, max_depth=4, jitter=True, seed=5, threads=5)
    np.testing.assert_array_equal(a, b)
    assert ra == rb
