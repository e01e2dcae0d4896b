#!/usr/bin/env python3
"""Regenerates the fixtures in this directory from the reference's own code.

Runs only where /root/reference exists (the build container): it drives oracle/_ref/ref_probe
and oracle/_ref/raytracer_ref, both compiled in place from the reference sources by
oracle/Makefile.  The fixtures are data only: image hashes, small raw images, and per-pixel
float records.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
REFBIN = os.path.join(ROOT, "oracle", "_ref", "raytracer_ref")

# (requested width, requested height) as typed on the reference's command line; the reference
# aligns both up to a multiple of 4, minimum 4 (src/commands.cpp:83-92)
IMAGE_SIZES = [
    (512, 512), (1920, 1080), (3840, 2160),
    (1001, 3), (512, 64), (0, 8), (4, 4), (64, 64), (128, 72), (100, 200), (64, 4), (36, 1000),
]
KEEP_FULL_IMAGE = {(64, 64), (128, 72), (100, 200), (4, 4), (1004, 4), (4, 8)}
FLOAT_PROBE_SIZES = [(512, 512), (1920, 1080), (100, 200)]
PROBES_PER_SIZE = 384


def run_probe(mode, w, h, extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.bin")
        res = subprocess.run([PROBE, mode, str(w), str(h), out, *map(str, extra)], check=True,
                             capture_output=True, text=True)
        kv = dict(tok.split("=") for tok in res.stdout.split())
        aw, ah = int(kv["width"]), int(kv["height"])
        data = np.fromfile(out, dtype=np.uint32 if mode == "image" else np.float32)
    return aw, ah, data


def main():
    if not (os.path.exists(PROBE) and os.path.exists(REFBIN)):
        sys.exit("oracle/_ref is not built: run `make -C oracle ref` in the build container")

    meta = {"generator": "tests/golden/make_golden.py", "images": [], "scheduler_runs": []}

    for (w, h) in IMAGE_SIZES:
        aw, ah, img = run_probe("image", w, h)
        assert img.size == aw * ah
        digest = hashlib.sha256(img.tobytes()).hexdigest()
        entry = {"requested": [w, h], "width": aw, "height": ah, "sha256": digest}
        if (aw, ah) in KEEP_FULL_IMAGE:
            name = f"ref_image_{aw}x{ah}.npy"
            np.save(os.path.join(HERE, name), img.reshape(ah, aw))
            entry["file"] = name
        meta["images"].append(entry)
        print(f"{w}x{h} -> {aw}x{ah} {digest}")

    # The real multithreaded scheduler path (src/software_raytracer.cpp:93-203), three runs per
    # size: must be stable and must equal the single-threaded probe image.
    for (w, h) in [(512, 512), (1920, 1080)]:
        digests = set()
        for _ in range(3):
            with tempfile.TemporaryDirectory() as td:
                subprocess.run([REFBIN, "--width", str(w), "--height", str(h)], check=True, cwd=td)
                with open(os.path.join(td, "output.tga"), "rb") as f:
                    header = f.read(18)
                    payload = f.read(w * h * 4)
                size = os.path.getsize(os.path.join(td, "output.tga"))
            digests.add(hashlib.sha256(payload).hexdigest())
        assert len(digests) == 1, digests
        probe = next(e for e in meta["images"] if e["requested"] == [w, h])
        assert probe["sha256"] in digests
        meta["scheduler_runs"].append({"width": w, "height": h, "sha256": digests.pop(),
                                       "tga_header_hex": header.hex(), "file_size": size})

    # Command-line behaviour of the reference program itself: (arguments) -> size in the TGA
    # header it writes and the length of the file.
    cli_cases = [
        [],
        ["--width", "1001", "--height", "3"],
        ["--width", "abc", "--height", "64"],
        ["--width", "0", "--height", "8"],
        ["--width", "64", "--width", "128", "--height", "32"],
        ["--width", "12x", "--width", "128", "--height", "32"],
        ["--height", "100", "--width"],
        ["--bogus", "--width", "256", "stray", "--height", "16"],
        ["--width", "+20", "--height", " 24"],
        ["--width", "18446744073709551615", "--height", "40"],
        ["--width", "-4294967200", "--height", "12"],
        ["--width", "70", "--height", "6", "--compute"],
    ]
    cli = []
    for args in cli_cases:
        with tempfile.TemporaryDirectory() as td:
            subprocess.run([REFBIN, *args], check=True, cwd=td)
            with open(os.path.join(td, "output.tga"), "rb") as f:
                header = f.read(18)
            size = os.path.getsize(os.path.join(td, "output.tga"))
        cw, ch = int.from_bytes(header[12:14], "little"), int.from_bytes(header[14:16], "little")
        # the image the reference's pixel kernel gives at that size (single-threaded probe: at
        # tiny sizes the reference's scheduler sometimes returns before its last tiles are copied)
        pw, ph, img = run_probe("image", cw, ch)
        assert (pw, ph) == (cw, ch)
        cli.append({"args": args, "width": cw, "height": ch, "header_hex": header.hex(), "file_size": size,
                    "payload_sha256": hashlib.sha256(img.tobytes()).hexdigest()})
        print("cli", args, cw, ch, size)
    with open(os.path.join(HERE, "reference_cli.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "cases": cli}, f, indent=1)

    rng = np.random.default_rng(20241220)
    for (w, h) in FLOAT_PROBE_SIZES:
        xs = rng.integers(0, w, PROBES_PER_SIZE - 8)
        ys = rng.integers(0, h, PROBES_PER_SIZE - 8)
        corners = [(0, 0), (w - 1, 0), (0, h - 1), (w - 1, h - 1), (w // 2, h // 2),
                   (w // 2 - 1, h // 2 - 1), (w // 2, h // 4), (w // 4, h // 2)]
        pts = list(zip(xs.tolist(), ys.tolist())) + corners
        flat = [c for p in pts for c in p]
        aw, ah, rec = run_probe("floats", w, h, flat)
        assert (aw, ah) == (w, h)
        rec = rec.reshape(-1, 20)
        np.save(os.path.join(HERE, f"ref_floats_{w}x{h}.npy"), rec)
        print(f"float probes {w}x{h}: {rec.shape[0]} records, {int(rec[:, 5].sum())} hits")

    with open(os.path.join(HERE, "reference_images.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
