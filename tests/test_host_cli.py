"""Host program (raytracer_amd/bin/raytracer_hip): command-line and output-file compatibility
with the reference program.  The CPU half checks argument quirks, the TGA header and file size,
the software backend's image (the reference's, byte for byte; the extended model against the
oracle) and the fallback from --hip to it on a box without a GPU; the GPU half checks the HIP
backend's payload through the same program."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "raytracer_amd", "bin", "raytracer_hip")
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
META = json.load(open(os.path.join(GOLDEN, "reference_images.json")))


def run_cli(tmp_path, *args):
    res = subprocess.run([CLI, *args], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    return res


def header_fields(path):
    with open(path, "rb") as f:
        h = f.read(18)
    return h, int.from_bytes(h[12:14], "little"), int.from_bytes(h[14:16], "little")


# What the reference program itself does with each argument list (fixture made by running the
# reference: defaults 512x512; malformed or out-of-range values fall back to the default; the
# first valid occurrence of a flag wins; sizes round up to a multiple of 4, minimum 4).
CLI_CASES = json.load(open(os.path.join(GOLDEN, "reference_cli.json")))["cases"]


def payload(path, w, h):
    data = np.fromfile(path, dtype=np.uint8, count=18 + w * h * 4)
    return data[18:]


@pytest.mark.parametrize("hip", [False, True], ids=["default", "hip_flag"])
@pytest.mark.parametrize("case", CLI_CASES, ids=lambda c: "_".join(c["args"]) or "defaults")
def test_cli_sizes_header_file_length_and_image(tmp_path, case, hip):
    """What the reference program does with the same arguments: size, header, file length, and
    the image of its pixel kernel at that size.  Without --hip the software backend renders; with
    it the HIP backend does, or -- on a box without a GPU -- the program falls back to the
    software backend as the reference does from its GPU backend (src/main.cpp:40-52): same image
    and exit status 0 either way."""
    res = run_cli(tmp_path, *case["args"], *(["--hip"] if hip else []))
    assert res.returncode == 0, res.stderr
    path = tmp_path / "output.tga"
    header, hw, hh = header_fields(path)
    assert (hw, hh) == (case["width"], case["height"])
    assert header.hex() == case["header_hex"]
    # the reference's size formula multiplies header size by payload size (src/output.cpp:11)
    assert os.path.getsize(path) == case["file_size"] == 18 * hw * hh * 4
    assert hashlib.sha256(payload(path, hw, hh).tobytes()).hexdigest() == case["payload_sha256"]


def test_cli_header_equals_reference_run():
    ref = next(r for r in META["scheduler_runs"] if (r["width"], r["height"]) == (512, 512))
    assert ref["tga_header_hex"] == bytes([0, 0, 2] + [0] * 9 + [0, 2, 0, 2, 32, 0]).hex()
    assert ref["file_size"] == 18 * 512 * 512 * 4


def test_cli_tight_file_and_output_name(tmp_path):
    run_cli(tmp_path, "--width", "64", "--height", "32", "--tight-file", "--output", "img.tga")
    assert os.path.getsize(tmp_path / "img.tga") == 18 + 64 * 32 * 4
    assert not (tmp_path / "output.tga").exists()


def test_cli_hip_flag_without_gpu_falls_back_to_the_software_backend(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    res = run_cli(tmp_path, "--hip", "--width", "64", "--height", "32", "--stats")
    assert res.returncode == 0
    assert "using the software backend" in res.stderr           # one line saying why; the reference says nothing
    stats = json.loads(res.stderr.strip().splitlines()[-1])
    assert stats["backend"] == "software" and stats["paths"] == 64 * 32
    np.testing.assert_array_equal(payload(tmp_path / "output.tga", 64, 32).view(np.uint32).reshape(32, 64), oracle.trace_primary(64, 32))


@pytest.mark.parametrize("size", [(512, 512), (1920, 1080), (1004, 4), (100, 200), (36, 1000)])
def test_cli_software_backend_image_equals_reference(tmp_path, size):
    w, h = size
    res = run_cli(tmp_path, "--width", str(w), "--height", str(h), "--stats")
    assert res.returncode == 0, res.stderr
    assert json.loads(res.stderr.strip().splitlines()[-1])["backend"] == "software"
    want = next(e["sha256"] for e in META["images"] if (e["width"], e["height"]) == (w, h))
    assert hashlib.sha256(payload(tmp_path / "output.tga", w, h).tobytes()).hexdigest() == want


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_cli_software_backend_is_independent_of_the_thread_count(tmp_path, threads):
    """Rows are dealt to joinable threads by one atomic counter and written in place: no tile can
    be lost or late whatever the thread count (the reference's queue-and-flag scheduler can return
    before its last tiles are copied, src/software_raytracer.cpp:173-177,259)."""
    for _ in range(3):
        res = run_cli(tmp_path, "--width", "1004", "--height", "4", "--threads", str(threads))
        assert res.returncode == 0
        np.testing.assert_array_equal(payload(tmp_path / "output.tga", 1004, 4).view(np.uint32).reshape(4, 1004),
                                      np.load(os.path.join(GOLDEN, "ref_image_1004x4.npy")))


@pytest.mark.parametrize("scene,args", [
    ("showcase", ["--spp", "4", "--depth", "6", "--seed", "3", "--jitter"]),
    ("divergence", ["--spp", "2", "--depth", "50", "--seed", "7", "--jitter"]),
    ("showcase", ["--spp", "70", "--depth", "3", "--seed", "11"]),
    (None, ["--spp", "8", "--depth", "8", "--seed", "1", "--jitter"]),
    ("showcase", ["--spp", "70", "--depth", "4", "--seed", "5", "--jitter"]),
], ids=["materials", "depth50", "two_sample_blocks_centred", "default_scene_jittered", "two_blocks_partial_grid_row"])
def test_cli_software_backend_extended_model_matches_oracle(tmp_path, scene, args):
    """Scene file, samples, depth, seed and jitter through the software backend: the same image,
    bit for bit, as the CPU oracle (and so as the HIP backend's exact build), ray count included."""
    from raytracer_amd import scenes

    sph = scenes.SCENES[scene]() if scene else None
    extra = []
    if sph is not None:
        scenes.save_scene(str(tmp_path / "s.txt"), sph)
        extra = ["--scene", "s.txt"]
    w, h = 96, 64
    res = run_cli(tmp_path, "--width", str(w), "--height", str(h), *extra, *args, "--stats", "--tight-file")
    assert res.returncode == 0, res.stderr
    stats = json.loads(res.stderr.strip().splitlines()[-1])
    got = np.fromfile(tmp_path / "output.tga", dtype=np.uint8)[18:].view(np.uint32).reshape(h, w)
    kw = {args[i].lstrip("-"): int(args[i + 1]) for i in range(0, len(args) - 1, 2) if args[i] != "--jitter"}
    want, _, rays, _ = oracle.trace_paths(w, h, spp=kw["spp"], max_depth=kw["depth"], seed=kw["seed"], jitter="--jitter" in args,
                                          spheres=sph, threads=4)
    np.testing.assert_array_equal(got, want)
    assert stats["backend"] == "software" and stats["rays"] == rays and stats["paths"] == w * h * kw["spp"]


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(512, 512), (1920, 1080), (1004, 4), (100, 200)])
def test_cli_image_equals_reference(tmp_path, size):
    w, h = size
    res = run_cli(tmp_path, "--hip", "--width", str(w), "--height", str(h), "--stats")
    assert res.returncode == 0, res.stderr
    assert json.loads(res.stderr.strip().splitlines()[-1])["backend"] == "hip"
    want = next(e["sha256"] for e in META["images"] if (e["width"], e["height"]) == (w, h))
    assert hashlib.sha256(payload(tmp_path / "output.tga", w, h).tobytes()).hexdigest() == want


@pytest.mark.gpu
def test_cli_scene_file_spp_depth_match_oracle(tmp_path):
    from raytracer_amd import scenes

    sph = scenes.showcase_scene()
    scenes.save_scene(str(tmp_path / "s.txt"), sph)
    res = run_cli(tmp_path, "--hip", "--width", "96", "--height", "64", "--scene", "s.txt", "--spp", "4", "--depth", "6",
                  "--seed", "3", "--jitter", "--stats", "--tight-file")
    assert res.returncode == 0, res.stderr
    stats = json.loads(res.stderr.strip().splitlines()[-1])
    assert stats["backend"] == "hip"
    got = np.fromfile(tmp_path / "output.tga", dtype=np.uint8)[18:].view(np.uint32).reshape(64, 96)
    want, _, rays, _ = oracle.trace_paths(96, 64, spp=4, max_depth=6, seed=3, jitter=True, spheres=sph, threads=4)
    np.testing.assert_array_equal(got, want)
    assert stats["rays"] == rays and stats["paths"] == 96 * 64 * 4
    # the same command line without --hip: the software backend's file is the same file
    os.rename(tmp_path / "output.tga", tmp_path / "gpu.tga")
    res = run_cli(tmp_path, "--width", "96", "--height", "64", "--scene", "s.txt", "--spp", "4", "--depth", "6",
                  "--seed", "3", "--jitter", "--tight-file")
    assert res.returncode == 0
    assert (tmp_path / "output.tga").read_bytes() == (tmp_path / "gpu.tga").read_bytes()


def test_cli_bad_scene_file_is_an_error(tmp_path):
    (tmp_path / "bad.txt").write_text("sphere 0 0 -2 1 plastic 1 1 1 0\n")
    res = run_cli(tmp_path, "--scene", "bad.txt", "--width", "16", "--height", "16")
    assert res.returncode == 2 and "bad.txt:1" in res.stderr


@pytest.mark.gpu
def test_cli_gpus_beyond_the_box_falls_back_to_the_software_backend(tmp_path):
    import torch

    n = torch.cuda.device_count()
    res = run_cli(tmp_path, "--hip", "--width", "64", "--height", "64", "--gpus", str(n + 1), "--stats")
    # the GPU backend's setup() refuses; the program falls back as the reference does (src/main.cpp:47-49)
    assert res.returncode == 0 and "--gpus" in res.stderr and "using the software backend" in res.stderr
    assert json.loads(res.stderr.strip().splitlines()[-1])["backend"] == "software"
    np.testing.assert_array_equal(payload(tmp_path / "output.tga", 64, 64).view(np.uint32).reshape(64, 64), oracle.trace_primary(64, 64))
