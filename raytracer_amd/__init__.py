"""raytracer_amd: ctypes binding of libhip_raytracer.so (include/hip_raytracer.h).

The product is the C ABI and the C++ host layer in ``raytracer_amd/host``; this module only lets
Python drivers (tests, bench.py) call the same entry points.  There is no CPU fallback: if the
library is missing or no GPU is usable, calls raise.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAYTRACER_AMD_LIB: another build of the same library (A/B timing of two builds in one session)
LIB_PATH = os.environ.get("RAYTRACER_AMD_LIB") or os.path.join(_HERE, "lib", "libhip_raytracer.so")

MAT_NORMAL, MAT_LAMBERT, MAT_METAL, MAT_DIELECTRIC = 0, 1, 2, 3
FLAG_JITTER, FLAG_FAST, FLAG_COUNT_RAYS, FLAG_NO_REGEN, FLAG_WALK_ALL, FLAG_NO_TIMING = 1, 2, 4, 8, 16, 32
PASS_PRIMARY, PASS_PATH = 0, 1

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


class AehipScene(ctypes.Structure):
    _fields_ = [
        ("background0", ctypes.c_float * 4),
        ("background1", ctypes.c_float * 4),
        ("camera_pos", ctypes.c_float * 4),
        ("sphere_count", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("spheres", ctypes.c_void_p),
    ]


class AehipParams(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_uint32),
        ("height", ctypes.c_uint32),
        ("row_begin", ctypes.c_uint32),
        ("row_count", ctypes.c_uint32),
        ("spp", ctypes.c_uint32),
        ("max_depth", ctypes.c_uint32),
        ("seed", ctypes.c_uint32),
        ("flags", ctypes.c_uint32),
        ("pass_", ctypes.c_uint32),
        ("list_source", ctypes.c_uint32),
        ("tile_rows", ctypes.c_uint32),
        ("tile_stride", ctypes.c_uint32),
    ]


class AehipStats(ctypes.Structure):
    _fields_ = [
        ("kernel_ms", ctypes.c_float),
        ("copy_ms", ctypes.c_float),
        ("rays", ctypes.c_uint64),
        ("paths", ctypes.c_uint64),
        ("lane_slots", ctypes.c_uint64),
        ("vgprs", ctypes.c_uint32),
        ("sgprs", ctypes.c_uint32),
        ("lds_bytes", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("samples_walked", ctypes.c_uint64),
    ]


# every symbol include/hip_raytracer.h declares
EXPORTS = (
    "aehip_init",
    "aehip_terminate",
    "aehip_device_count",
    "aehip_last_error",
    "aehip_backend_create",
    "aehip_backend_destroy",
    "aehip_backend_setup",
    "aehip_backend_trace",
    "aehip_backend_trace_device",
    "aehip_backend_trace_device_static",
    "aehip_backend_stats",
    "aehip_backend_background_columns",
    "aehip_band_pack_columns",
    "aehip_band_expand_columns",
    "aehip_trace_frame",
    "aehip_frame_band",
)


class HipRaytracerError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Loads the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipRaytracerError(f"{LIB_PATH} is missing: build it with `make lib` (hipcc, gfx950)")
        # A process must hold ONE HIP runtime.  PyTorch wheels bundle their own libamdhip64.so.7;
        # if this library pulled in /opt/rocm's copy first, torch would later fail with "No HIP
        # GPUs are available".  Python drivers use torch for device memory and collectives, so
        # let torch load its runtime first; this library then binds to the same one by soname.
        if os.environ.get("RAYTRACER_AMD_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = ctypes.CDLL(LIB_PATH)
        vp = ctypes.c_void_p
        L.aehip_init.restype = ctypes.c_int
        L.aehip_terminate.restype = None
        L.aehip_device_count.restype = ctypes.c_int
        L.aehip_last_error.restype = ctypes.c_char_p
        L.aehip_backend_create.argtypes = [ctypes.c_int, vp, ctypes.POINTER(vp)]
        L.aehip_backend_destroy.argtypes = [vp]
        L.aehip_backend_destroy.restype = None
        L.aehip_backend_setup.argtypes = [vp, ctypes.POINTER(AehipScene), ctypes.POINTER(AehipParams)]
        L.aehip_backend_trace.argtypes = [vp]
        L.aehip_backend_trace_device.argtypes = [vp, vp, vp, vp]
        L.aehip_backend_trace_device_static.argtypes = [vp, vp, vp, vp]
        L.aehip_backend_stats.argtypes = [vp, ctypes.POINTER(AehipStats)]
        L.aehip_backend_background_columns.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
        u32 = ctypes.c_uint32
        L.aehip_band_pack_columns.argtypes = [vp, u32, u32, u32, u32, u32, vp, vp]
        L.aehip_band_expand_columns.argtypes = [vp, ctypes.c_uint64, vp, u32, u32, u32, u32, vp, vp]
        L.aehip_trace_frame.argtypes = [ctypes.c_int, ctypes.POINTER(AehipScene), ctypes.POINTER(AehipParams), vp,
                                        ctypes.POINTER(AehipStats)]
        L.aehip_frame_band.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int] + [ctypes.POINTER(ctypes.c_uint32)] * 3
        _lib = L
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise HipRaytracerError(f"{what} failed ({rc}): {load().aehip_last_error().decode()}")


def default_spheres() -> np.ndarray:
    """ae::raytracer::sphere of the reference (src/raytracer.cpp:13)."""
    s = np.zeros(1, dtype=SPHERE_DTYPE)
    s["center"][0] = (0.0, 0.0, -2.0)
    s["radius"][0] = 1.0
    s["albedo"][0] = (1.0, 1.0, 1.0)
    s["material"][0] = MAT_NORMAL
    return s


def make_scene(spheres=None, bg0=None, bg1=None, cam=None):
    """Scene struct with the reference's globals as defaults (src/raytracer.cpp:10-13)."""
    if spheres is None:
        spheres = default_spheres()
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    sc = AehipScene()
    b0 = bg0 if bg0 is not None else (1.0, 1.0, 1.0)
    b1 = bg1 if bg1 is not None else tuple(np.float32(v) / np.float32(255.0) for v in (0x4D, 0xA6, 0xF0))
    c = cam if cam is not None else (0.0, 0.0, 1.0)
    for i in range(3):
        sc.background0[i] = b0[i]
        sc.background1[i] = b1[i]
        sc.camera_pos[i] = c[i]
    sc.background0[3] = 1.0
    sc.background1[3] = 1.0
    sc.sphere_count = len(spheres)
    sc.spheres = spheres.ctypes.data
    return sc, spheres


class HipRaytracer:
    """Python twin of ``ae::hip_raytracer``: ctor(framebuffer) / setup() / trace()
    (src/raytracer.h:11-33), plus trace_device() for device-resident output."""

    def __init__(self, framebuffer: np.ndarray | None = None, device: int = 0):
        self._lib = load()
        _check(self._lib.aehip_init(), "aehip_init")
        self._fb = framebuffer
        if framebuffer is not None:
            assert framebuffer.dtype == np.uint32 and framebuffer.flags["C_CONTIGUOUS"]
        self._h = ctypes.c_void_p()
        _check(self._lib.aehip_backend_create(device, framebuffer.ctypes.data if framebuffer is not None else None,
                                              ctypes.byref(self._h)), "aehip_backend_create")
        self._keep = None
        self.params = None

    def setup(self, width, height, spp=1, max_depth=1, seed=1, flags=0, pass_=PASS_PRIMARY, row_begin=0,
              row_count=0, spheres=None, list_source=0, tile_rows=0, tile_stride=0, **scene_kw) -> bool:
        sc, keep = make_scene(spheres, **scene_kw)
        p = AehipParams(width, height, row_begin, row_count, spp, max_depth, seed, flags, pass_)
        p.list_source = list_source      # 0 = library's choice, 1 = LDS staging, 2 = scalar-cache stream
        p.tile_rows, p.tile_stride = tile_rows, tile_stride     # 0, 0 = contiguous band
        rc = self._lib.aehip_backend_setup(self._h, ctypes.byref(sc), ctypes.byref(p))
        del keep
        self.params = p
        self.rows = row_count if row_count else band_row_count(height, row_begin, tile_rows, tile_stride)
        self.width = width
        return rc == 0

    def last_error(self) -> str:
        return self._lib.aehip_last_error().decode()

    def trace(self) -> None:
        _check(self._lib.aehip_backend_trace(self._h), "aehip_backend_trace")

    def trace_device(self, argb_ptr: int, rgba_ptr: int | None = None, stream: int | None = None) -> None:
        _check(self._lib.aehip_backend_trace_device(self._h, argb_ptr, rgba_ptr, stream), "aehip_backend_trace_device")

    def trace_device_static(self, argb_ptr: int, rgba_ptr: int | None = None, stream: int | None = None) -> None:
        """trace_device through the library's fixed scratch array: no device allocation in the call."""
        _check(self._lib.aehip_backend_trace_device_static(self._h, argb_ptr, rgba_ptr, stream), "aehip_backend_trace_device_static")

    def stats(self) -> AehipStats:
        st = AehipStats()
        _check(self._lib.aehip_backend_stats(self._h, ctypes.byref(st)), "aehip_backend_stats")
        return st

    def background_columns(self) -> tuple[int, int]:
        """(col_begin, col_end): outside these columns every pixel of a band row is the row's background."""
        c0, c1 = ctypes.c_uint32(), ctypes.c_uint32()
        _check(self._lib.aehip_backend_background_columns(self._h, ctypes.byref(c0), ctypes.byref(c1)), "aehip_backend_background_columns")
        return c0.value, c1.value

    def close(self) -> None:
        if self._h:
            self._lib.aehip_backend_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def band_row_count(height, row_begin, tile_rows=0, tile_stride=0) -> int:
    """Rows of the band that starts at row_begin and runs to the image's last row: every row
    (tile_rows 0), or the rows of tiles of tile_rows rows placed every tile_stride rows."""
    span = max(0, height - row_begin)
    if tile_rows == 0:
        return span
    return span // tile_stride * tile_rows + min(span % tile_stride, tile_rows)


def render(width, height, spp=1, max_depth=1, seed=1, jitter=False, fast=False, count_rays=False,
           pass_=PASS_PATH, row_begin=0, row_count=0, spheres=None, device=0, no_regen=False,
           tile_rows=0, tile_stride=0, **scene_kw):
    """One-shot helper: render a band into host memory through setup()/trace().
    Returns (argb uint32 [rows, width] in band order, stats)."""
    rows = row_count if row_count else band_row_count(height, row_begin, tile_rows, tile_stride)
    fb = np.zeros((rows, width), dtype=np.uint32)
    rt = HipRaytracer(fb, device)
    flags = (FLAG_JITTER if jitter else 0) | (FLAG_FAST if fast else 0) | (FLAG_COUNT_RAYS if count_rays else 0) \
        | (FLAG_NO_REGEN if no_regen else 0)
    try:
        if not rt.setup(width, height, spp, max_depth, seed, flags, pass_, row_begin, row_count, spheres,
                        tile_rows=tile_rows, tile_stride=tile_stride, **scene_kw):
            raise HipRaytracerError("setup failed: " + rt.last_error())
        rt.trace()
        st = rt.stats()
    finally:
        rt.close()
    return fb, st


def trace_frame(n_devices, width, height, spp=1, max_depth=1, seed=1, jitter=False, fast=False, count_rays=False,
                pass_=PASS_PATH, row_begin=0, row_count=0, spheres=None, tile_rows=0, **scene_kw):
    """One frame cut into row tiles dealt to n_devices bands by the library itself
    (aehip_trace_frame): rendered on n_devices GPUs of this node from this process, gathered on
    device 0, copied to host in image order.  Returns (argb uint32 [rows, width], stats)."""
    L = load()
    _check(L.aehip_init(), "aehip_init")
    rows = row_count if row_count else height - row_begin
    fb = np.zeros((rows, width), dtype=np.uint32)
    sc, keep = make_scene(spheres, **scene_kw)
    flags = (FLAG_JITTER if jitter else 0) | (FLAG_FAST if fast else 0) | (FLAG_COUNT_RAYS if count_rays else 0)
    p = AehipParams(width, height, row_begin, row_count, spp, max_depth, seed, flags, pass_)
    p.tile_rows = tile_rows
    st = AehipStats()
    _check(L.aehip_trace_frame(n_devices, ctypes.byref(sc), ctypes.byref(p), fb.ctypes.data, ctypes.byref(st)), "aehip_trace_frame")
    del keep
    return fb, st
