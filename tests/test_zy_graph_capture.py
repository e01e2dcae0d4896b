"""aehip_backend_trace_device_static under HIP stream capture, and proof that the call allocates nothing.

The entry point exists for callers that cannot call hipMalloc.  Two checks that equality with the
allocating entry point alone would not give:
  * the call is captured into a HIP graph (torch.cuda.graph, capture mode "thread_local": a hipMalloc, a
    synchronisation or any other call that capture forbids, made by the capturing thread -- the one the
    library's call runs on -- fails the capture; other threads of the test process are left out of it) and
    the graph's replays reproduce the allocating call's image bit for bit;
  * free device memory is the same before and after plain calls.
"""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu

CASES = {
    # name: (width, height, spp, depth, seed, extra flags, scene)
    "camera_fast": (1280, 720, 128, 8, 1, rt.FLAG_FAST, None),                                  # the benchmark's kernel: no scratch
    "camera_exact": (320, 180, 70, 8, 1, 0, None),
    "primary": (512, 512, 1, 1, 1, 0, "primary"),
    "streamed": (320, 180, 6, 12, 21, rt.FLAG_COUNT_RAYS, "divergence"),                        # scratch + counters + queue
    "streamed_blocks": (160, 90, 70, 12, 21, rt.FLAG_FAST | rt.FLAG_COUNT_RAYS, "divergence"),    # two sample blocks
    "first_hits": (100, 52, 5, 6, 3, rt.FLAG_COUNT_RAYS, "long_list"),                           # first-hit pre-pass
}


def request(name):
    w, h, spp, depth, seed, extra, scene = CASES[name]
    kw = dict(spp=spp, max_depth=depth, seed=seed, flags=rt.FLAG_JITTER | rt.FLAG_NO_TIMING | extra, pass_=rt.PASS_PATH)
    if scene == "primary":
        kw = dict(flags=rt.FLAG_NO_TIMING, pass_=rt.PASS_PRIMARY)
    elif scene == "divergence":
        kw["spheres"] = scenes.divergence_scene()
    elif scene == "long_list":
        kw["spheres"] = scenes.stress_scene(1100, 42)
    return w, h, kw


@pytest.mark.parametrize("name", list(CASES))
def test_static_call_captured_into_a_graph_replays_to_the_allocating_calls_bits(name):
    import torch

    w, h, kw = request(name)
    back = rt.HipRaytracer(None)
    try:
        assert back.setup(w, h, **kw), back.last_error()
        want = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        back.trace_device(want.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want_rays = back.stats().rays

        out = torch.full((h, w), 0x55, dtype=torch.int32, device="cuda")
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            back.trace_device_static(out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert bool((out == 0x55).all()), "the captured call ran instead of being recorded"
        for _ in range(2):
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, want)
            assert back.stats().rays == want_rays
            out.fill_(0x55)
        del graph
    finally:
        back.close()


def test_a_timed_request_is_refused_under_capture():
    import torch

    back = rt.HipRaytracer(None)
    try:
        assert back.setup(256, 144, spp=16, flags=rt.FLAG_JITTER, pass_=rt.PASS_PATH)       # no AEHIP_FLAG_NO_TIMING
        out = torch.zeros((144, 256), dtype=torch.int32, device="cuda")
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            with pytest.raises(rt.HipRaytracerError, match="AEHIP_FLAG_NO_TIMING"):
                back.trace_device_static(out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            out.add_(1)                 # (a graph needs a node)
        graph.replay()
        torch.cuda.synchronize()
        assert bool((out == 1).all())
    finally:
        back.close()


@pytest.mark.parametrize("name", ["camera_fast", "streamed", "first_hits"])
def test_static_call_leaves_free_device_memory_as_it_was(name):
    import torch

    w, h, kw = request(name)
    back = rt.HipRaytracer(None)
    try:
        assert back.setup(w, h, **kw), back.last_error()
        out = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        back.trace_device_static(out.data_ptr(), None, stream)         # (the first launch of a kernel loads its code object)
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info()
        for _ in range(3):
            back.trace_device_static(out.data_ptr(), None, stream)
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info()
        assert free_after == free_before
    finally:
        back.close()
