"""Deterministic scene generators shared by every driver (tests, bench, CLI scene files).

The reference hard-codes one sphere (src/raytracer.cpp:10-13); the larger scenes here are this
build's own.  Randomness comes from the same PCG hash the renderer uses, evaluated with numpy
uint32 arithmetic, so a scene is a pure function of (name, seed, count) on any machine.
"""
from __future__ import annotations

import numpy as np

from . import MAT_DIELECTRIC, MAT_LAMBERT, MAT_METAL, MAT_NORMAL, SPHERE_DTYPE, default_spheres


def pcg_hash(v: np.ndarray) -> np.ndarray:
    v = np.asarray(v, dtype=np.uint32)
    with np.errstate(over="ignore"):
        s = v * np.uint32(747796405) + np.uint32(2891336453)
        w = ((s >> ((s >> np.uint32(28)) + np.uint32(4))) ^ s) * np.uint32(277803737)
    return (w >> np.uint32(22)) ^ w


def _uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n floats in [0,1), 24 bits each, from counter (seed, stream, index)."""
    idx = np.arange(n, dtype=np.uint32)
    base = pcg_hash(np.uint32(seed) + pcg_hash(np.uint32(stream)))
    bits = pcg_hash(idx + base)
    return (bits >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def _mix_materials(s: np.ndarray, seed: int, kinds=(MAT_LAMBERT, MAT_METAL, MAT_DIELECTRIC)) -> None:
    n = len(s)
    pick = (_uniform(seed, 10, n) * len(kinds)).astype(np.int64).clip(0, len(kinds) - 1)
    kind = np.asarray(kinds, dtype=np.uint32)[pick]
    s["material"] = kind
    alb = np.stack([_uniform(seed, 11, n), _uniform(seed, 12, n), _uniform(seed, 13, n)], axis=1)
    s["albedo"] = np.float32(0.25) + np.float32(0.7) * alb
    fuzz = _uniform(seed, 14, n) * np.float32(0.3)
    s["param"] = np.where(kind == MAT_METAL, fuzz, np.float32(0.0))
    s["param"] = np.where(kind == MAT_DIELECTRIC, np.float32(1.5), s["param"])
    s["albedo"][kind == MAT_DIELECTRIC] = 1.0


def default_scene() -> np.ndarray:
    return default_spheres()


def stress_scene(count: int = 10000, seed: int = 42) -> np.ndarray:
    """About `count` small spheres in a slab in front of the camera: x,y in [-20,20],
    z in [-60,-3], radius in [0.05,0.5], materials mixed in equal parts.  Brute-force
    intersection load: every ray tests every sphere."""
    s = np.zeros(count, dtype=SPHERE_DTYPE)
    s["center"][:, 0] = _uniform(seed, 1, count) * np.float32(40.0) - np.float32(20.0)
    s["center"][:, 1] = _uniform(seed, 2, count) * np.float32(40.0) - np.float32(20.0)
    s["center"][:, 2] = np.float32(-3.0) - _uniform(seed, 3, count) * np.float32(57.0)
    s["radius"] = np.float32(0.05) + _uniform(seed, 4, count) * np.float32(0.45)
    _mix_materials(s, seed)
    return s


def divergence_scene(count: int = 300, seed: int = 7) -> np.ndarray:
    """A ground sphere plus `count` spheres of mixed diffuse / metal / dielectric response close
    to the camera axis, so neighbouring pixels follow paths of very different length."""
    s = np.zeros(count + 1, dtype=SPHERE_DTYPE)
    s["center"][0] = (0.0, -1000.6, -3.0)
    s["radius"][0] = 1000.0
    s["center"][1:, 0] = _uniform(seed, 1, count) * np.float32(8.0) - np.float32(4.0)
    s["center"][1:, 1] = _uniform(seed, 2, count) * np.float32(1.6) - np.float32(0.45)
    s["center"][1:, 2] = np.float32(-1.5) - _uniform(seed, 3, count) * np.float32(7.0)
    s["radius"][1:] = np.float32(0.08) + _uniform(seed, 4, count) * np.float32(0.22)
    _mix_materials(s[1:], seed)
    s["material"][0] = MAT_LAMBERT
    s["albedo"][0] = (0.5, 0.5, 0.5)
    s["param"][0] = 0.0
    return s


def showcase_scene() -> np.ndarray:
    """The reference sphere (normal-shaded) with one sphere of each other material around it
    and a diffuse ground: a small scene that exercises every material branch."""
    s = np.zeros(5, dtype=SPHERE_DTYPE)
    s["center"][0] = (0.0, 0.0, -2.0); s["radius"][0] = 1.0; s["material"][0] = MAT_NORMAL; s["albedo"][0] = (1, 1, 1)
    s["center"][1] = (0.0, -101.0, -2.0); s["radius"][1] = 100.0; s["material"][1] = MAT_LAMBERT; s["albedo"][1] = (0.8, 0.8, 0.0)
    s["center"][2] = (-2.1, 0.0, -2.2); s["radius"][2] = 1.0; s["material"][2] = MAT_DIELECTRIC; s["albedo"][2] = (1, 1, 1); s["param"][2] = 1.5
    s["center"][3] = (2.1, 0.0, -2.2); s["radius"][3] = 1.0; s["material"][3] = MAT_METAL; s["albedo"][3] = (0.8, 0.6, 0.2); s["param"][3] = 0.25
    s["center"][4] = (0.0, 1.7, -2.6); s["radius"][4] = 0.6; s["material"][4] = MAT_LAMBERT; s["albedo"][4] = (0.1, 0.2, 0.5)
    return s


def grazing_scene() -> np.ndarray:
    """Adversarial for the intersection arithmetic: the camera (0,0,1) sits on the surface of a
    huge diffuse ground sphere and next to large mirrors, so many rays start on, or graze, the
    sphere they are tested against and discriminants hover around zero."""
    s = np.zeros(6, dtype=SPHERE_DTYPE)
    s["center"][0] = (0.0, -1000.0, 1.0); s["radius"][0] = 1000.0; s["material"][0] = MAT_LAMBERT; s["albedo"][0] = (0.7, 0.7, 0.7)
    s["center"][1] = (0.0, 500.0, -900.0); s["radius"][1] = 700.0; s["material"][1] = MAT_METAL; s["albedo"][1] = (0.9, 0.9, 0.9)
    s["center"][2] = (-3.0, 1.0, -4.0); s["radius"][2] = 1.0; s["material"][2] = MAT_METAL; s["albedo"][2] = (0.8, 0.8, 0.9); s["param"][2] = 0.05
    s["center"][3] = (3.0, 1.0, -4.0); s["radius"][3] = 1.0; s["material"][3] = MAT_DIELECTRIC; s["albedo"][3] = (1, 1, 1); s["param"][3] = 1.5
    s["center"][4] = (0.0, 0.25, -2.0); s["radius"][4] = 0.25; s["material"][4] = MAT_NORMAL; s["albedo"][4] = (1, 1, 1)
    s["center"][5] = (0.0, 1.0, -6.0); s["radius"][5] = 1.0; s["material"][5] = MAT_LAMBERT; s["albedo"][5] = (0.9, 0.3, 0.3)
    return s


SCENES = {
    "grazing": grazing_scene,
    "default": default_scene,
    "stress10k": stress_scene,
    "divergence": divergence_scene,
    "showcase": showcase_scene,
}


_MATERIAL_NAMES = {MAT_NORMAL: "normal", MAT_LAMBERT: "lambert", MAT_METAL: "metal", MAT_DIELECTRIC: "dielectric"}


def save_scene(path: str, spheres: np.ndarray, camera=None, background0=None, background1=None) -> None:
    """Writes the text scene format the host CLI reads with --scene.  Floats are printed with 9
    significant digits, which round-trips binary32 exactly."""
    f9 = lambda v: " ".join(f"{float(x):.9g}" for x in v)
    with open(path, "w") as f:
        f.write("# raytracer_hip scene\n")
        if camera is not None:
            f.write(f"camera {f9(camera)}\n")
        if background0 is not None:
            f.write(f"background0 {f9(background0)}\n")
        if background1 is not None:
            f.write(f"background1 {f9(background1)}\n")
        for s in spheres:
            f.write(f"sphere {f9(s['center'])} {float(s['radius']):.9g} {_MATERIAL_NAMES[int(s['material'])]} "
                    f"{f9(s['albedo'])} {float(s['param']):.9g}\n")
