#!/usr/bin/env python3
"""Folds the rocprofv3 --pmc outputs of tools/profile_bench.sh / tools/profile_scenes.sh into
profiles/r03_counters.json, the file bench.py prices its roofline figures from.

    python tools/collect_counters.py <workload key> <frames | auto> <profile dir> [<profile dir> ...]

<frames>: frames rendered in each profiled run; "auto" = the number of dispatches of the kernel with the
largest total duration (bench.py's headline frames are one dispatch of the camera kernel each, however
many frames its time-based warm-up renders).

<profile dir> holds pmc_*/pmc_results.db (rocpd sqlite).  For every renderer kernel the file keeps
the per-FRAME totals of each counter (sum over the run's dispatches / frames rendered in it), the
number of dispatches per frame and the mean dispatch duration.  The file also records the sha256
of the kernel sources it was measured on; bench.py refuses figures from another version.

Executed FP32 flops of a kernel = 64 * (FLOPS_FP32 + TRANS_F32).  SQ_INSTS_VALU_FLOPS_FP32 is the
hardware's own flop weight of the wave-level FP32 instructions issued: 1 for an add or a multiply,
2 for an FMA, and twice that for the packed forms (v_pk_fma_f32 counts once in FMA_F32 but works
on two values a lane: measured on the camera table's walk, FLOPS_FP32 = 2 MUL_F32 + 4 FMA_F32
there).  Where a profile has no FLOPS_FP32 the sum ADD_F32 + MUL_F32 + 2 FMA_F32 stands in, which
is the same number for kernels without packed instructions.  Every lane of a wave64 instruction
counts whether or not its EXEC bit is set, so this is issue-slot work -- an upper bound of useful
flops and exactly what the VALU roofline prices.
"""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles", "r03_counters.json")
KERNEL_SOURCES = ["raytracer_amd/csrc/hip_raytracer_kernels.hip", "raytracer_amd/csrc/rt_exact_math.h"]
KERNELS = ["camera_hybrid_kernel", "camera_kernel", "primary_kernel", "path_stream_first_hits_kernel", "path_stream_kernel",
           "first_hit_kernel", "path_kernel", "resolve_block_kernel", "ring_pixels_kernel", "walked_pixels_kernel"]


def kernel_source_sha256():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def short_name(full):
    for k in KERNELS:
        if k + "<" in full or k + "(" in full or full.endswith(k) or ("::" + k) in full:
            return k
    return None


def main():
    key, frames_arg = sys.argv[1], sys.argv[2]
    kernels = {}
    for d in sys.argv[3:]:
        for sub in sorted(os.listdir(d)):
            db = os.path.join(d, sub, "pmc_results.db")
            if not os.path.exists(db):
                continue
            con = sqlite3.connect(db)
            rows = con.execute("select kernel_name, counter_name, sum(value), count(*), avg(duration) from counters_collection "
                               "group by kernel_name, counter_name").fetchall()
            con.close()
            if frames_arg == "auto":
                # this pass's frames: dispatches of the renderer kernel that took the most time in it
                busiest = max((r for r in rows if short_name(r[0]) is not None), key=lambda r: r[3] * r[4], default=None)
                if busiest is None:
                    continue
                frames = busiest[3]
            else:
                frames = int(frames_arg)
            for full, counter, total, n, dur in rows:
                name = short_name(full)
                if name is None:
                    continue
                k = kernels.setdefault(name, {"full_names": []})
                if full not in k["full_names"]:
                    k["full_names"].append(full)
                k[counter] = total / frames
                k["dispatches_per_frame"] = n / frames
                k.setdefault("avg_dispatch_us", {})[sub] = dur / 1e3
    for k in kernels.values():
        k["full_names"] = [n[:160] for n in k["full_names"]]
        if "SQ_INSTS_VALU_FLOPS_FP32" in k and "SQ_INSTS_VALU_TRANS_F32" in k:
            k["executed_fp32_flop_per_frame"] = 64.0 * (k["SQ_INSTS_VALU_FLOPS_FP32"] + k["SQ_INSTS_VALU_TRANS_F32"])
        elif all(c in k for c in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32")):
            k["executed_fp32_flop_per_frame"] = 64.0 * (k["SQ_INSTS_VALU_ADD_F32"] + k["SQ_INSTS_VALU_MUL_F32"]
                                                        + 2.0 * k["SQ_INSTS_VALU_FMA_F32"] + k["SQ_INSTS_VALU_TRANS_F32"])
    doc = {"workloads": {}}
    if os.path.exists(OUT):
        with open(OUT) as f:
            doc = json.load(f)
    sha = kernel_source_sha256()
    if doc.get("kernel_source_sha256") != sha:
        doc = {"workloads": {}}                    # counters of another version of the kernels are worthless
    doc["kernel_source_sha256"] = sha
    doc["kernel_sources"] = KERNEL_SOURCES
    doc["note"] = ("per-frame totals of rocprofv3 --pmc counters (SQ counters are summed over the chip); collected by "
                   "tools/profile_bench.sh and tools/profile_scenes.sh, folded by tools/collect_counters.py")
    doc["workloads"][key] = {"frames_in_run": frames_arg if frames_arg == "auto" else int(frames_arg), "kernels": kernels}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    flops = {n: k.get("executed_fp32_flop_per_frame") for n, k in kernels.items()}
    print(key, json.dumps(flops))


if __name__ == "__main__":
    main()
