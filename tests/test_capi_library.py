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
    assert ctypes.sizeof(raytracer_amd.AehipStats) == 48
    assert raytracer_amd.SPHERE_DTYPE.itemsize == 48


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
