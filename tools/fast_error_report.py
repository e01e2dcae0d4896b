#!/usr/bin/env python3
"""Reports how far the fast build (FMA, v_rcp/v_rsq/v_sqrt) lands from the exact build on the
jitter-free primary-hit pass: distribution of the per-channel float difference and of the packed
byte difference.  GPU only; prints JSON lines."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt  # noqa: E402


def device_render(w, h, fast):
    import torch

    back = rt.HipRaytracer(None)
    assert back.setup(w, h, flags=rt.FLAG_FAST if fast else 0, pass_=rt.PASS_PRIMARY), back.last_error()
    argb = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    rgba = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    back.trace_device(argb.data_ptr(), rgba.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    back.close()
    return argb.cpu().numpy().view(np.uint32), rgba.cpu().numpy()


def main():
    for (w, h) in [(512, 512), (1920, 1080), (3840, 2160)]:
        ea, er = device_render(w, h, False)
        fa, fr = device_render(w, h, True)
        d = np.abs(fr[..., :3] - er[..., :3]).max(axis=-1)
        ch = lambda a: np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], -1).astype(np.int32)
        bd = np.abs(ch(fa) - ch(ea)).max(axis=-1)
        print(json.dumps({
            "size": [w, h], "max_abs": float(d.max()), "p999999": float(np.quantile(d, 0.999999)),
            "n_gt_1e-5": int((d > 1e-5).sum()), "n_gt_1e-6": int((d > 1e-6).sum()), "n_gt_1e-4": int((d > 1e-4).sum()),
            "pixels": w * h, "byte_max": int(bd.max()), "byte_diff_pixels": int((bd > 0).sum()),
        }))


if __name__ == "__main__":
    main()
