"""The C-ABI library builds, loads and exports every symbol include/hip_raytracer.h declares.
No compute calls: these run without a GPU."""
import ctypes
import os
import re

import pytest

import raytracer_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hip_raytracer.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aehip_[a-z_]+)\s*\(", text)))


def test_header_declares_what_python_binds():
    assert declared_symbols() == sorted(raytracer_amd.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = raytracer_amd.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_struct_layouts_match_header():
    assert ctypes.sizeof(raytracer_amd.AehipParams) == 48
    assert ctypes.sizeof(raytracer_amd.AehipScene) == 64
    assert ctypes.sizeof(raytracer_amd.AehipStats) == 56
    assert raytracer_amd.SPHERE_DTYPE.itemsize == 48


def test_field_offsets_match_the_compiled_header(tmp_path):
    """The header itself, compiled as C, against the ctypes mirrors: size and offset of every field."""
    import subprocess

    structs = {"aehip_params": raytracer_amd.AehipParams, "aehip_scene": raytracer_amd.AehipScene,
               "aehip_stats": raytracer_amd.AehipStats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hip_raytracer.h"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, *_ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname.rstrip("_")}));')     # pass_ mirrors `pass`
    lines += ['  printf("aehip_sphere size %zu\\n", sizeof(aehip_sphere));', "  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out if l}
    for cname, cls in structs.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls)
        for fname, *_ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
    assert got[("aehip_sphere", "size")] == raytracer_amd.SPHERE_DTYPE.itemsize


def test_enum_values_match_the_compiled_header(tmp_path):
    """Flags, passes and materials as the header's compiler sees them against the Python constants."""
    import subprocess

    names = {"AEHIP_FLAG_JITTER": raytracer_amd.FLAG_JITTER, "AEHIP_FLAG_FAST": raytracer_amd.FLAG_FAST,
             "AEHIP_FLAG_COUNT_RAYS": raytracer_amd.FLAG_COUNT_RAYS, "AEHIP_FLAG_NO_REGEN": raytracer_amd.FLAG_NO_REGEN,
             "AEHIP_FLAG_WALK_ALL": raytracer_amd.FLAG_WALK_ALL, "AEHIP_FLAG_NO_TIMING": raytracer_amd.FLAG_NO_TIMING,
             "AEHIP_PASS_PRIMARY": raytracer_amd.PASS_PRIMARY, "AEHIP_PASS_PATH": raytracer_amd.PASS_PATH,
             "AEHIP_MAT_NORMAL": raytracer_amd.MAT_NORMAL, "AEHIP_MAT_LAMBERT": raytracer_amd.MAT_LAMBERT,
             "AEHIP_MAT_METAL": raytracer_amd.MAT_METAL, "AEHIP_MAT_DIELECTRIC": raytracer_amd.MAT_DIELECTRIC}
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    declared = set(re.findall(r"\b(AEHIP_(?:FLAG|PASS|MAT)_[A-Z_]+)\s*=", text))
    assert declared == set(names), "an enumerator of the header has no Python constant (or the other way round)"
    lines = ['#include <stdio.h>', '#include "hip_raytracer.h"', "int main(void) {"]
    lines += [f'  printf("{n} %ld\\n", (long){n});' for n in names]
    lines += ["  return 0;", "}"]
    src = tmp_path / "enums.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "enums"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {l.split()[0]: int(l.split()[1]) for l in out if l}
    assert got == names


def test_null_arguments_are_rejected_without_touching_a_device():
    lib = raytracer_amd.load()
    assert lib.aehip_backend_create(0, None, None) == -2
    assert b"null" in lib.aehip_last_error()
    assert lib.aehip_backend_setup(None, None, None) == -2
    assert lib.aehip_backend_trace(None) == -2
    assert lib.aehip_backend_trace_device(None, None, None, None) == -2
    lib.aehip_backend_destroy(None)
    lib.aehip_terminate()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(raytracer_amd, "_lib", None)
    monkeypatch.setattr(raytracer_amd, "LIB_PATH", "/nonexistent/libhip_raytracer.so")
    with pytest.raises(raytracer_amd.HipRaytracerError):
        raytracer_amd.load()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("rows,tile", [(2160, 4), (1080, 4), (75, 7), (41, 16), (64, 1), (130, 5)])
def test_frame_band_arithmetic_equals_the_python_sharding(n, rows, tile):
    """aehip_frame_band (the band layout aehip_trace_frame renders and gathers by) against
    raytracer_amd.sharding.tiled_band, which bench.py's ranks use: same first row, same row count,
    bands stored densely in band order, every image row in exactly one band.  No GPU involved."""
    import ctypes

    from raytracer_amd.sharding import band_image_rows, tiled_band

    L = raytracer_amd.load()
    if n > -(-rows // tile):
        first, count, before = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        assert L.aehip_frame_band(rows, tile, n, 0, ctypes.byref(first), ctypes.byref(count), ctypes.byref(before)) != 0
        return
    seen, expect_before = [], 0
    for band in range(n):
        first, count, before = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        assert L.aehip_frame_band(rows, tile, n, band, ctypes.byref(first), ctypes.byref(count), ctypes.byref(before)) == 0
        want = tiled_band(rows, n, band, tile) if n > 1 else dict(row_begin=0, row_count=rows, tile_rows=0, tile_stride=0)
        assert (first.value, count.value) == (want["row_begin"], want["row_count"])
        assert before.value == expect_before
        expect_before += count.value
        seen += band_image_rows(dict(row_begin=first.value, row_count=count.value, tile_rows=tile if n > 1 else 0, tile_stride=tile * n))
    assert sorted(seen) == list(range(rows))


def test_committed_counters_were_measured_on_these_kernel_sources():
    """bench.py prices its roofline figures from profiles/r03_counters.json and refuses a file
    collected on another version of the kernels: keep the committed file current (re-run
    tools/profile_bench.sh and tools/profile_scenes.sh on the GPU box after changing a kernel)."""
    import hashlib
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r03_counters.json")) as f:
        doc = json.load(f)
    h = hashlib.sha256()
    for rel in doc["kernel_sources"]:
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    assert doc["kernel_source_sha256"] == h.hexdigest()
    import bench

    keys = ["headline_fast", "headline_exact", "headline_fast_walked", "headline_exact_walked"]
    keys += [f"{name}_{mode}" for name in bench.CONFIG_CASES for mode in ("fast", "exact")]     # every workload bench.py prices
    for key in keys:
        kernels = doc["workloads"][key]["kernels"]
        assert any("executed_fp32_flop_per_frame" in k for k in kernels.values()), key
