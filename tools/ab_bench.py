"""A/B timing of library builds on one GPU box: tools/ab_bench.py [--mode fast] [--reps 3] [--args "..."] lib_a.so lib_b.so ...

Runs bench.py once per library and repetition, interleaved (a b a b ...), and prints each library's
ms_per_step values with their minimum and median: run-to-run and box-to-box spread of a single
bench.py run is several per cent, more than many of the changes one wants to compare."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="fast")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--args", default="")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
times = {lib: [] for lib in a.libs}
for rep in range(a.reps):
    for lib in a.libs:
        env = dict(os.environ, RAYTRACER_AMD_LIB=os.path.abspath(lib))
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--mode", a.mode, "--steps", str(a.steps), "--warmup", "20",
               "--no-cpu-baseline", "--no-aux", "--no-second-mode"] + a.args.split()
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            print(lib, "failed", out.stderr[-500:])
            sys.exit(1)
        times[lib].append(json.loads(line[-1])["ms_per_step"])
for lib, t in times.items():
    print(f"{os.path.basename(lib):40s} min {min(t):.4f} median {statistics.median(t):.4f}  all " + " ".join(f"{x:.4f}" for x in t), flush=True)
