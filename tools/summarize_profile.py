#!/usr/bin/env python3
"""Turns the rocprofv3 (rocpd sqlite) outputs of tools/profile_bench.sh into a text summary fit
for committing under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_exact profiles/r01_bench_exact.txt

Kernel statistics come from the --kernel-trace --stats run; counters from the separate --pmc
runs.  HBM byte counters are corrected as measured on gfx950: FETCH_SIZE under-reports wide
streaming reads by 2x (reported in KiB, so bytes = value * 1024 * 2); WRITE_SIZE is exact for
16-byte-per-lane stores (bytes = value * 1024).
"""
import os
import sqlite3
import sys


def q(db, sql):
    c = sqlite3.connect(db)
    try:
        return c.execute(sql).fetchall()
    finally:
        c.close()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = []
    trace = os.path.join(src, "trace", "trace_results.db")
    if os.path.exists(trace):
        out.append("== rocprofv3 --kernel-trace --stats : kernels by total time ==")
        out.append(f"{'calls':>6} {'total_us':>12} {'avg_us':>12} {'pct':>7}  name")
        for name, calls, total, avg, pct in q(trace, "select name,total_calls,total_duration,average,percentage from top_kernels"):
            out.append(f"{calls:6d} {total:12.1f} {avg:12.1f} {pct:7.2f}  {name[:150]}")
        out.append("")
        out.append("== per-dispatch resources of the renderer's kernels ==")
        rows = q(trace, "select name,grid_x,grid_y,workgroup_x,vgpr_count,accum_vgpr_count,sgpr_count,lds_size,scratch_size,"
                        "min(duration),avg(duration),max(duration),count(*) from kernels "
                        "where name like '%path_kernel%' or name like '%primary_kernel%' or name like '%camera_kernel%' or name like '%camera_hybrid_kernel%' or name like '%ring_pixels_kernel%' "
                        "or name like '%path_stream%' or name like '%first_hit_kernel%' or name like '%resolve_block_kernel%' group by name,grid_x,grid_y")
        for r in rows:
            out.append(f"{r[0][:90]}\n    grid {r[1]}x{r[2]} wg {r[3]} vgpr {r[4]} agpr {r[5]} sgpr {r[6]} lds {r[7]} scratch {r[8]}"
                       f" | dur us min {r[9] / 1e3:.1f} avg {r[10] / 1e3:.1f} max {r[11] / 1e3:.1f} over {r[12]} launches")
        out.append("")
    for sub in ("pmc_sq", "pmc_flops", "pmc_fetch", "pmc_write"):
        db = os.path.join(src, sub, "pmc_results.db")
        if not os.path.exists(db):
            continue
        out.append(f"== rocprofv3 --pmc ({sub}) : per-launch mean over the renderer's kernel launches ==")
        rows = q(db, "select kernel_name,counter_name,avg(value),count(*),avg(duration) from counters_collection "
                     "where kernel_name like '%path_kernel%' or kernel_name like '%primary_kernel%' or kernel_name like '%camera_kernel%' or kernel_name like '%camera_hybrid_kernel%' or kernel_name like '%ring_pixels_kernel%' "
                     "or kernel_name like '%path_stream%' or kernel_name like '%first_hit_kernel%' or kernel_name like '%resolve_block_kernel%' "
                     "group by kernel_name,counter_name")
        last = None
        flops = {}
        for kname, cname, val, n, dur in rows:
            if kname != last:
                out.append(f"  {kname[:120]}")
                last = kname
            # the hardware's flop weight of the FP32 instructions issued (packed forms count twice), plus the transcendentals
            if cname in ("SQ_INSTS_VALU_FLOPS_FP32", "SQ_INSTS_VALU_TRANS_F32"):
                flops[kname] = flops.get(kname, 0.0) + 64.0 * val
                flops[(kname, "us")] = dur / 1e3
            extra = ""
            if cname == "FETCH_SIZE":
                extra = f"  -> HBM read bytes/launch (x1024 x2 gfx950 correction) = {val * 1024 * 2:.3e}"
            if cname == "WRITE_SIZE":
                extra = f"  -> HBM write bytes/launch (x1024) = {val * 1024:.3e}"
            out.append(f"{cname:24s} {val:18.1f}  (n={n}, kernel avg {dur / 1e3:.1f} us){extra}")
        for kname, total in flops.items():
            if isinstance(kname, tuple):
                continue
            us = flops[(kname, "us")]
            out.append(f"  executed FP32 flop/launch = 64 x (FLOPS_FP32 + TRANS) = {total:.4e}  -> {total / (us * 1e-6) / 1e12:.2f} TFLOP/s "
                       f"= {total / (us * 1e-6) / 157.3e12:.3f} of the 157.3 TFLOP/s FP32 vector peak (this profiled run's {us:.1f} us)  [{kname[:60]}]")
        out.append("")
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    with open(dst, "w") as f:
        f.write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
