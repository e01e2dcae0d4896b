"""Host program (raytracer_amd/bin/raytracer_hip): command-line and output-file compatibility
with the reference program.  The CPU half checks what happens before any GPU work (argument
quirks, TGA header and file size); the GPU half checks the rendered payload."""
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


@pytest.mark.parametrize("case", CLI_CASES, ids=lambda c: "_".join(c["args"]) or "defaults")
def test_cli_sizes_header_and_file_length(tmp_path, case):
    """Runs with or without a GPU: the output file is created before the backend starts."""
    run_cli(tmp_path, *case["args"])
    path = tmp_path / "output.tga"
    header, hw, hh = header_fields(path)
    assert (hw, hh) == (case["width"], case["height"])
    assert header.hex() == case["header_hex"]
    # the reference's size formula multiplies header size by payload size (src/output.cpp:11)
    assert os.path.getsize(path) == case["file_size"] == 18 * hw * hh * 4


def test_cli_header_equals_reference_run():
    ref = next(r for r in META["scheduler_runs"] if (r["width"], r["height"]) == (512, 512))
    assert ref["tga_header_hex"] == bytes([0, 0, 2] + [0] * 9 + [0, 2, 0, 2, 32, 0]).hex()
    assert ref["file_size"] == 18 * 512 * 512 * 4


def test_cli_tight_file_and_output_name(tmp_path):
    run_cli(tmp_path, "--width", "64", "--height", "32", "--tight-file", "--output", "img.tga")
    assert os.path.getsize(tmp_path / "img.tga") == 18 + 64 * 32 * 4
    assert not (tmp_path / "output.tga").exists()


def test_cli_reports_failure_loudly_without_gpu(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    res = run_cli(tmp_path)
    assert res.returncode != 0
    assert "raytracer_hip:" in res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(512, 512), (1920, 1080), (1004, 4), (100, 200)])
def test_cli_image_equals_reference(tmp_path, size):
    w, h = size
    res = run_cli(tmp_path, "--width", str(w), "--height", str(h))
    assert res.returncode == 0, res.stderr
    data = np.fromfile(tmp_path / "output.tga", dtype=np.uint8, count=18 + w * h * 4)
    want = next(e["sha256"] for e in META["images"] if (e["width"], e["height"]) == (w, h))
    assert hashlib.sha256(data[18:].tobytes()).hexdigest() == want


@pytest.mark.gpu
def test_cli_scene_file_spp_depth_match_oracle(tmp_path):
    from raytracer_amd import scenes

    sph = scenes.showcase_scene()
    scenes.save_scene(str(tmp_path / "s.txt"), sph)
    res = run_cli(tmp_path, "--width", "96", "--height", "64", "--scene", "s.txt", "--spp", "4", "--depth", "6",
                  "--seed", "3", "--jitter", "--stats", "--tight-file")
    assert res.returncode == 0, res.stderr
    stats = json.loads(res.stderr.strip().splitlines()[-1])
    got = np.fromfile(tmp_path / "output.tga", dtype=np.uint8)[18:].view(np.uint32).reshape(64, 96)
    want, _, rays, _ = oracle.trace_paths(96, 64, spp=4, max_depth=6, seed=3, jitter=True, spheres=sph, threads=4)
    np.testing.assert_array_equal(got, want)
    assert stats["rays"] == rays and stats["paths"] == 96 * 64 * 4


@pytest.mark.gpu
def test_cli_bad_scene_file_is_an_error(tmp_path):
    (tmp_path / "bad.txt").write_text("sphere 0 0 -2 1 plastic 1 1 1 0\n")
    res = run_cli(tmp_path, "--scene", "bad.txt", "--width", "16", "--height", "16")
    assert res.returncode == 2 and "bad.txt:1" in res.stderr


@pytest.mark.gpu
def test_cli_gpus_beyond_the_box_is_an_error(tmp_path):
    import torch

    n = torch.cuda.device_count()
    res = run_cli(tmp_path, "--width", "64", "--height", "64", "--gpus", str(n + 1))
    assert res.returncode == 1 and "--gpus" in res.stderr
