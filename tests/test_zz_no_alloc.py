"""aehip_backend_trace_device_static: the allocation-free twin of aehip_backend_trace_device.

Called through the C ABI, its output must equal the existing entry point's bit for bit (packed
words, floats, ray counts), on calls in a row that share the GPU's fixed scratch array -- the
second with a smaller input, so whatever the first call left in the array must not leak into it.
Covers the streamed path pass (the scratch's user) with one and several passes over the array,
with and without the first-hit pre-pass, more than one sample block, both arithmetic builds, and
the passes that use no scratch at all.
"""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def both(back, w, h):
    """Renders the backend's request through both entry points; returns the two result triples."""
    import torch

    out = []
    for call in (back.trace_device, back.trace_device_static):
        argb = torch.full((h, w), 0x55, dtype=torch.int32, device="cuda")
        rgba = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
        call(argb.data_ptr(), rgba.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        st = back.stats()
        out.append((argb.cpu().numpy().view(np.uint32), rgba.cpu().numpy(), st.rays))
    return out


def check_equal(a, b):
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert a[2] == b[2]


@pytest.mark.parametrize("fast", [False, True], ids=["exact", "fast"])
def test_static_scratch_entry_point_equals_the_allocating_one_on_calls_in_a_row(fast):
    flags = rt.FLAG_JITTER | rt.FLAG_COUNT_RAYS | (rt.FLAG_FAST if fast else 0)
    back = rt.HipRaytracer(None)
    try:
        # 640x360 at 70 samples: 3,600 strips of 49,920 bytes each -- three passes over the 64 MiB array, two sample blocks
        sph = scenes.divergence_scene()
        assert back.setup(640, 360, 70, 12, 21, flags, rt.PASS_PATH, spheres=sph), back.last_error()
        first = both(back, 640, 360)
        check_equal(*first)
        assert first[0][2] > 640 * 360 * 70
        # a smaller input straight after, fewer samples (another stride in the array), ragged size
        assert back.setup(200, 120, 5, 50, 7, flags, rt.PASS_PATH, spheres=sph), back.last_error()
        second = both(back, 200, 120)
        check_equal(*second)
        # and once more the static call alone: same bits as before, whatever the array held
        again = both(back, 200, 120)
        check_equal(second[1], again[1])
        # long list: first-hit pre-pass (and the list prefilter in the exact build) through the same array
        long_list = scenes.stress_scene(1100, 42)
        assert back.setup(100, 52, 5, 6, 3, flags, rt.PASS_PATH, spheres=long_list), back.last_error()
        check_equal(*both(back, 100, 52))
    finally:
        back.close()


def test_static_scratch_entry_point_on_passes_without_scratch():
    back = rt.HipRaytracer(None)
    try:
        assert back.setup(512, 512, pass_=rt.PASS_PRIMARY)
        check_equal(*both(back, 512, 512))
        assert back.setup(256, 144, 64, 8, 1, rt.FLAG_JITTER | rt.FLAG_FAST, rt.PASS_PATH)            # camera pass, fast build
        check_equal(*both(back, 256, 144))
        assert back.setup(96, 54, 4, 6, 3, rt.FLAG_JITTER | rt.FLAG_COUNT_RAYS, rt.PASS_PATH, spheres=scenes.showcase_scene())   # per-lane pass, counters
        a, b = both(back, 96, 54)
        check_equal(a, b)
        assert a[2] > 96 * 54 * 4
    finally:
        back.close()


def test_static_scratch_entry_point_rejects_a_band_beyond_its_limit():
    back = rt.HipRaytracer(None)
    try:
        import torch

        # 65,535 x 5,600 pixels at 64 samples: 5.7 million strips, beyond 4096 passes of 1,344 strips
        assert back.setup(65535, 5600, 64, 4, 1, rt.FLAG_JITTER, rt.PASS_PATH, spheres=scenes.divergence_scene()), back.last_error()
        argb = torch.zeros(16, dtype=torch.int32, device="cuda")          # never written: the call must refuse first
        with pytest.raises(rt.HipRaytracerError, match="static scratch"):
            back.trace_device_static(argb.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    finally:
        back.close()


def test_two_backends_on_two_streams_take_turns_on_the_gpus_scratch_array():
    """The fixed array is one per GPU, whichever backend calls.  Two backends whose static calls go to two streams,
    queued in alternation without waiting in between, must each get its own allocating call's image and ray count:
    the library orders scratch users on a GPU after one another (an event per GPU), and a backend reads its ray
    counters back before another backend's next launch can clear them (stats() waits for the GPU first)."""
    import torch

    sph = scenes.divergence_scene()
    flags = rt.FLAG_JITTER | rt.FLAG_COUNT_RAYS
    requests = [(320, 180, 6, 12, 21), (200, 120, 5, 20, 7)]
    backs, wants = [], []
    try:
        for (w, h, spp, depth, seed) in requests:
            back = rt.HipRaytracer(None)
            assert back.setup(w, h, spp, depth, seed, flags, rt.PASS_PATH, spheres=sph), back.last_error()
            ref = torch.zeros((h, w), dtype=torch.int32, device="cuda")
            back.trace_device(ref.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            backs.append(back)
            wants.append((ref, back.stats().rays))
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [torch.zeros_like(ref) for ref, _ in wants]
        torch.cuda.synchronize()
        for round_ in range(3):
            for k in (0, 1):
                outs[k].fill_(0x55)
                streams[k].wait_stream(torch.cuda.current_stream())
                backs[k].trace_device_static(outs[k].data_ptr(), None, streams[k].cuda_stream)
            for k in (1, 0):
                streams[k].synchronize()
                assert torch.equal(outs[k], wants[k][0]), (round_, k)
        # ray counts: the array's counters belong to whoever launched last, so they are read per backend, launch by launch
        for k in (0, 1):
            backs[k].trace_device_static(outs[k].data_ptr(), None, streams[k].cuda_stream)
            assert backs[k].stats().rays == wants[k][1]
    finally:
        for back in backs:
            back.close()
