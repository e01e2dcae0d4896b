"""The benchmark's committed line (profiles/r03_b_bench_line*.json, bench.py run on the GPU box): what a reader of
the driver's record -- which keeps the parsed standard keys, the roofline and cpu_baseline objects and the last
2,000 characters of the line -- must be able to find there."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = ["r03_b_bench_line.json", "r03_b_bench_line_steps20_warmup5.json"]


def load(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        text = f.read().strip().splitlines()[-1]
    return text, json.loads(text)


@pytest.mark.parametrize("name", LINES)
def test_the_lines_tail_holds_the_summary(name):
    text, line = load(name)
    tail = text[-2000:]
    start = tail.find('"summary": ')
    assert start >= 0, "the summary does not fit the 2,000 characters the driver keeps"
    summary = json.loads(tail[start + len('"summary": '):-1])         # (the line's closing brace follows it)
    assert summary == line["summary"]
    for key in ("headline", "exact"):
        assert set(summary[key]) == {"value", "ms_per_step", "kernel_ms", "frac"} and summary[key]["frac"] is not None
    for mode in ("fast", "exact"):
        assert summary["every_sample_walked"][mode]["frac"] is not None
    assert all(summary["gathered_frame_equals_single_gpu_frame"].values()) and len(summary["gathered_frame_equals_single_gpu_frame"]) == 4
    assert 0 < summary["samples_walked_per_frame"] < 3840 * 2160 * 1024
    import bench
    for case in bench.CONFIG_CASES:
        for mode in ("fast", "exact"):
            entry = summary["configs"][f"{case}_{mode}"]
            assert entry["frac"] is not None and entry["kernel_ms"] > 0


@pytest.mark.parametrize("name", LINES)
def test_the_line_follows_the_contract(name):
    _, line = load(name)
    assert line["metric"] == "Mpaths/sec at 3840x2160" and line["unit"] == "Mpaths/s" and line["n_gpus"] == 1
    assert line["higher_is_better"] is True and line["vs_baseline"] is None and line["dtype"] == "f32" and line["data"] == "synthetic"
    assert "3840x2160" in line["config"]["workload"] and "1024 spp" in line["config"]["workload"]
    paths = 3840 * 2160 * 1024
    assert abs(line["value"] - paths / (line["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * line["value"]
    roof = line["roofline"]
    assert roof["peak"] == 157.3 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and 0 < roof["frac"] < 1
    assert roof["traffic"] >= 3840 * 2160 * 4               # the frame is stored once
    for key in ("exact_frac", "walked_fast_frac", "walked_exact_frac"):     # scalars the driver's parse keeps
        assert 0 < roof[key] < 1
    base = line["cpu_baseline"]
    assert base["kind"] == "port" and base["cores"] >= 1 and base["value"] > 0
    assert line["prewarm_ms"] >= 100.0
    assert line["samples_walked_per_frame"] == line["summary"]["samples_walked_per_frame"]
    assert abs(line["rays_per_s"] - line["samples_walked_per_frame"] / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["rays_per_s"]


def test_the_drivers_short_run_matches_the_long_one():
    """--steps 20 --warmup 5 (what the driver passes) against the defaults (200 / 50), same box: within 3 %."""
    _, long_run = load(LINES[0])
    _, short_run = load(LINES[1])
    assert abs(short_run["ms_per_step"] / long_run["ms_per_step"] - 1.0) <= 0.03
    assert abs(short_run["roofline"]["frac"] / long_run["roofline"]["frac"] - 1.0) <= 0.03
