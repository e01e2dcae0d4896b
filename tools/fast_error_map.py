#!/usr/bin/env python3
"""Where does the fast build leave the reference's floats?  For jitter-free single-sphere renders
(primary pass and the sampling kernel at 1 centred sample) prints, per decade of the conditioning
value

    q = disc * dd * r^2 / M^2,   disc = h^2 - dd c,   M = h^2 + dd (oc.oc + r^2)

(unnormalised ray `draw` from the camera, h = draw.oc, dd = draw.draw, c = oc.oc - r^2; worked out
here in float64), the number of hit pixels and the largest |fast - exact| colour difference.  The
reference's quadratic amplifies rounding by about M / (r sqrt(dd disc)) = 1 / sqrt(q), so q is the
value a kernel can test to decide whether its own rounding is good enough.  GPU only; JSON lines."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt  # noqa: E402


def device_rgba(w, h, fast, pass_, spheres=None, cam=None, spp=1):
    import torch

    back = rt.HipRaytracer(None)
    kw = {} if cam is None else {"cam": cam}
    assert back.setup(w, h, spp=spp, max_depth=1, flags=rt.FLAG_FAST if fast else 0, pass_=pass_, spheres=spheres, **kw), back.last_error()
    argb = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    rgba = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    back.trace_device(argb.data_ptr(), rgba.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    back.close()
    return rgba.cpu().numpy()[..., :3]


def conditioning(w, h, center, radius, cam):
    vw, vh = (w / h, 1.0) if w > h else (1.0, h / w)
    xs = (np.arange(w, dtype=np.float64) + 0.5) * (vw / w) - vw * 0.5 - cam[0]
    ys = (np.arange(h, dtype=np.float64) + 0.5) * (vh / h) - vh * 0.5 - cam[1]
    dx, dy = np.meshgrid(xs, ys)
    dz = -cam[2]
    oc = np.asarray(center, dtype=np.float64) - np.asarray(cam, dtype=np.float64)
    r2 = float(radius) ** 2
    dd = dx * dx + dy * dy + dz * dz
    hh = dx * oc[0] + dy * oc[1] + dz * oc[2]
    occ = float(oc @ oc)
    disc = hh * hh - dd * (occ - r2)
    big = hh * hh + dd * (occ + r2)
    return disc * dd * r2 / (big * big), disc / big


def report(tag, w, h, center=(0.0, 0.0, -2.0), radius=1.0, cam=(0.0, 0.0, 1.0)):
    sph = None
    if (center, radius) != ((0.0, 0.0, -2.0), 1.0):
        sph = np.zeros(1, dtype=rt.SPHERE_DTYPE)
        sph["center"][0] = center
        sph["radius"] = radius
    q, rel = conditioning(w, h, center, radius, cam)
    for kernel, pass_ in (("primary", rt.PASS_PRIMARY), ("sampling", rt.PASS_PATH)):
        if pass_ == rt.PASS_PRIMARY and sph is not None and False:
            continue
        exact = device_rgba(w, h, False, pass_, sph, cam)
        fast = device_rgba(w, h, True, pass_, sph, cam)
        err = np.abs(fast - exact).max(axis=-1)
        rows = []
        edges = [-np.inf, -1e-3, -1e-5, -1e-7, 0.0] + [10.0 ** e for e in range(-12, 1)] + [np.inf]
        for lo, hi in zip(edges[:-1], edges[1:]):
            m = (q >= lo) & (q < hi)
            if m.any():
                rows.append({"q": [lo, hi], "pixels": int(m.sum()), "max_err": float(err[m].max()), "gt_1e-5": int((err[m] > 1e-5).sum()),
                             "gt_5e-6": int((err[m] > 5e-6).sum())})
        bad = err > 1e-5
        print(json.dumps({"case": tag, "kernel": kernel, "size": [w, h], "max_err": float(err.max()), "n_gt_1e-5": int(bad.sum()),
                          "largest_q_with_err_gt_1e-5": float(q[bad].max()) if bad.any() else None,
                          "largest_q_with_err_gt_5e-6": float(q[err > 5e-6].max()) if (err > 5e-6).any() else None,
                          "largest_q_with_err_gt_2e-6": float(q[err > 2e-6].max()) if (err > 2e-6).any() else None,
                          "most_negative_rel_disc_that_hits_in_exact": float(rel[(exact != fast).any(axis=-1) & (rel < 0)].min())
                          if ((exact != fast).any(axis=-1) & (rel < 0)).any() else None,
                          "bins": rows}), flush=True)


def main():
    for (w, h) in [(512, 512), (1920, 1080), (3840, 2160)]:
        report("default", w, h)
    # other views: far and small spheres (amplification ~ (distance / radius)^2), off-axis cameras
    report("far_small", 1024, 576, (0.3, 0.1, -6.0), 0.5)
    report("very_far", 1024, 576, (0.0, 0.0, -20.0), 1.0)
    report("large_near", 1024, 576, (0.5, -0.2, -3.0), 2.5)
    report("off_axis_camera", 1024, 576, (0.0, 0.0, -2.0), 1.0, (0.6, 0.3, 1.4))
    report("grazing", 1024, 576, (1.0, 0.0, -2.0), 1.0)


if __name__ == "__main__":
    main()
