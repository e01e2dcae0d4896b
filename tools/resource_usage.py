#!/usr/bin/env python3
"""Registers, spills, scratch and occupancy of every kernel of the library, from the compiler's own remarks.

    python tools/resource_usage.py [out.txt] [extra hipcc flags ...]

Compiles raytracer_amd/csrc/hip_raytracer_kernels.hip for gfx950 with the Makefile's flags plus
-Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one line a kernel.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Iinclude"]
FIELDS = [("SGPRs", "sgpr"), ("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("ScratchSize [bytes/lane]", "scratch"),
          ("Occupancy [waves/SIMD]", "occ"), ("SGPRs Spill", "sgpr_spill"), ("VGPRs Spill", "vgpr_spill"),
          ("LDS Size [bytes/block]", "lds")]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else None
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-Rpass-analysis=kernel-resource-usage", "-shared", "-o", os.path.join(td, "lib.so"),
               "raytracer_amd/csrc/hip_raytracer_kernels.hip"]
        res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
    if res.returncode != 0:
        sys.exit(res.stderr[-4000:])
    blocks = re.split(r"remark: [^\n]*Function Name: ", res.stderr)[1:]
    lines = []
    for b in blocks:
        mangled = b.split("\n")[0].strip().split()[0]
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void \(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name)
        vals = {}
        for label, key in FIELDS:
            m = re.search(re.escape(label) + r": (\S+)", b)
            vals[key] = m.group(1) if m else "?"
        lines.append(f"{name[:78]:78s} vgpr {vals['vgpr']:>3} agpr {vals['agpr']:>2} sgpr {vals['sgpr']:>3} sgpr_spill {vals['sgpr_spill']:>3} "
                     f"vgpr_spill {vals['vgpr_spill']:>3} scratch {vals['scratch']:>4} occ {vals['occ']:>2} lds {vals['lds']:>6}")
    text = "\n".join(lines) + "\n"
    print(text, end="")
    if out:
        with open(out, "w") as f:
            f.write("# hipcc -Rpass-analysis=kernel-resource-usage, flags: " + " ".join(FLAGS + extra) + "\n" + text)


if __name__ == "__main__":
    main()
