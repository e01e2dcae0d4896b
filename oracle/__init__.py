"""TEST INFRASTRUCTURE ONLY: ctypes access to the CPU oracle (oracle/_build/liboracle.so).

Imported by tests/, by ``__graft_entry__.smoke()`` and by ``bench.py``'s cpu_baseline leg, never
by the product package.  Nothing here touches the GPU or the HIP library.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
REF_BINARY = os.path.join(_HERE, "_ref", "raytracer_ref")
REF_PROBE = os.path.join(_HERE, "_ref", "ref_probe")

MAT_NORMAL, MAT_LAMBERT, MAT_METAL, MAT_DIELECTRIC = 0, 1, 2, 3
FLAG_JITTER = 1

# one record per sphere, 48 bytes; the HIP library's aehip_sphere has the same layout
SPHERE_DTYPE = np.dtype(
    [
        ("center", np.float32, 3),
        ("radius", np.float32),
        ("albedo", np.float32, 3),
        ("param", np.float32),
        ("material", np.uint32),
        ("reserved", np.uint32, 3),
    ]
)
assert SPHERE_DTYPE.itemsize == 48


class OrcScene(ctypes.Structure):
    _fields_ = [
        ("background0", ctypes.c_float * 4),
        ("background1", ctypes.c_float * 4),
        ("camera_pos", ctypes.c_float * 4),
        ("sphere_count", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("spheres", ctypes.c_void_p),
    ]


class OrcParams(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_uint32),
        ("height", ctypes.c_uint32),
        ("row_begin", ctypes.c_uint32),
        ("row_count", ctypes.c_uint32),
        ("spp", ctypes.c_uint32),
        ("max_depth", ctypes.c_uint32),
        ("seed", ctypes.c_uint32),
        ("flags", ctypes.c_uint32),
    ]


def build() -> None:
    """Compile the C restatement (and, where /root/reference exists, oracle/_ref)."""
    subprocess.run(["make", "-C", _HERE, "port"], check=True, capture_output=True)
    if os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        u32, f32p, u32p = ctypes.c_uint32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32)
        L.orc_setup_viewport.argtypes = [u32, u32, f32p, f32p]
        L.orc_pack_argb32.argtypes = [ctypes.c_float] * 4
        L.orc_pack_argb32.restype = u32
        L.orc_trace_primary.argtypes = [ctypes.POINTER(OrcScene), u32, u32, u32, u32, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_trace_paths.argtypes = [ctypes.POINTER(OrcScene), ctypes.POINTER(OrcParams), ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.orc_trace_paths_mt.argtypes = [ctypes.POINTER(OrcScene), ctypes.POINTER(OrcParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.orc_trace_paths_mt.restype = ctypes.c_double
        L.orc_trace_primary_mt.argtypes = [ctypes.POINTER(OrcScene), u32, u32, ctypes.c_int, ctypes.c_void_p]
        L.orc_trace_primary_mt.restype = ctypes.c_double
        L.orc_pcg_hash.argtypes = [u32]
        L.orc_pcg_hash.restype = u32
        L.orc_path_key.argtypes = [u32, u32, u32]
        L.orc_path_key.restype = u32
        L.orc_pixel_term.argtypes = [u32, u32]
        L.orc_pixel_term.restype = u32
        L.orc_sample_term.argtypes = [u32, u32]
        L.orc_sample_term.restype = u32
        L.orc_sample_offset.argtypes = [u32, u32, u32, f32p, f32p]
        L.orc_sample_row_length_log2.argtypes = [u32]
        L.orc_sample_row_length_log2.restype = u32
        L.orc_rng_next.argtypes = [u32p]
        L.orc_rng_next.restype = u32
        L.orc_cossin_2pi.argtypes = [ctypes.c_float, f32p, f32p]
        L.orc_unit_vector.argtypes = [u32, f32p]
        L.orc_online_cpus.restype = ctypes.c_int
        _lib = L
    return _lib


def default_spheres() -> np.ndarray:
    """The reference's built-in scene: one unit sphere at (0,0,-2) shaded by its normal
    (src/raytracer.cpp:13)."""
    s = np.zeros(1, dtype=SPHERE_DTYPE)
    s["center"][0] = (0.0, 0.0, -2.0)
    s["radius"][0] = 1.0
    s["albedo"][0] = (1.0, 1.0, 1.0)
    s["material"][0] = MAT_NORMAL
    return s


def make_scene(spheres: np.ndarray | None = None, bg0=None, bg1=None, cam=None) -> tuple[OrcScene, np.ndarray]:
    """Returns (scene struct, sphere array kept alive by the caller)."""
    if spheres is None:
        spheres = default_spheres()
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    sc = OrcScene()
    # src/raytracer.cpp:10-12; background1 = color(0xff4da6f0): byte/255.0f (src/color.cpp:32-36)
    b0 = bg0 if bg0 is not None else (1.0, 1.0, 1.0)
    b1 = bg1 if bg1 is not None else tuple(np.float32(v) / np.float32(255.0) for v in (0x4D, 0xA6, 0xF0))
    c = cam if cam is not None else (0.0, 0.0, 1.0)
    for i in range(3):
        sc.background0[i] = b0[i]
        sc.background1[i] = b1[i]
        sc.camera_pos[i] = c[i]
    sc.background0[3] = 1.0
    sc.background1[3] = 1.0
    sc.camera_pos[3] = 0.0
    sc.sphere_count = len(spheres)
    sc.spheres = spheres.ctypes.data
    return sc, spheres


def trace_primary(width, height, row_begin=0, row_count=None, spheres=None, want_rgba=False, **scene_kw):
    row_count = height - row_begin if row_count is None else row_count
    sc, keep = make_scene(spheres, **scene_kw)
    argb = np.empty((row_count, width), dtype=np.uint32)
    rgba = np.empty((row_count, width, 4), dtype=np.float32) if want_rgba else None
    lib().orc_trace_primary(ctypes.byref(sc), width, height, row_begin, row_count,
                            argb.ctypes.data, rgba.ctypes.data if want_rgba else None)
    del keep
    return (argb, rgba) if want_rgba else argb


def trace_paths(width, height, spp=1, max_depth=1, seed=1, jitter=False, row_begin=0, row_count=None,
                spheres=None, threads=1, want_rgba=False, **scene_kw):
    """Returns (argb, rgba-or-None, ray_count, wall_ms)."""
    row_count = height - row_begin if row_count is None else row_count
    sc, keep = make_scene(spheres, **scene_kw)
    p = OrcParams(width, height, row_begin, row_count, spp, max_depth, seed, FLAG_JITTER if jitter else 0)
    argb = np.empty((row_count, width), dtype=np.uint32)
    rgba = np.empty((row_count, width, 4), dtype=np.float32) if want_rgba else None
    rays = ctypes.c_uint64(0)
    ms = lib().orc_trace_paths_mt(ctypes.byref(sc), ctypes.byref(p), int(threads), argb.ctypes.data,
                                  rgba.ctypes.data if want_rgba else None, ctypes.byref(rays))
    del keep
    return argb, rgba, rays.value, ms
