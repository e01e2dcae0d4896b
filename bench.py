#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s rendering the default scene at 3840x2160, 1024 samples per pixel.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode fast|exact] [--spp S]

One step = one full frame: every rank renders its row band with the HIP path kernel (scene and
framebuffer resident in HBM), then the bands are gathered to rank 0 (RCCL) when N > 1.  Total
work per step is fixed (one 4K frame), so scaling over N is "strong".  Rank 0 prints ONE JSON
line.  With N > 1 the driver launches this file under torch.distributed.run, one rank per GPU.

The line also carries
  roofline      the path kernel's algorithmic FP32 flops / its mean launch time (HIP events on
                the launch stream, inside the timed region) against the FP32 vector peak, and
  cpu_baseline  the CPU oracle (a port of the reference's pixel kernel extended with sampling)
                timed on this host's cores on a bounded sample of the same frame, plus the real
                reference binary's 1-sample figure where oracle/_ref was shipped.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 3840, 2160
FP32_VECTOR_PEAK_TFLOPS = 157.3     # MI355X datasheet FP32 vector peak (256 CU x 256 flop/clk x 2.4 GHz)

# Algorithmic FP32 flops of the default-scene path pass, counted on the reference's formulas with
# div and sqrt = 1 flop each and integer / compare / convert = 0:
#   per sample: primary ray 9 + normalise 9 + sphere test 15 = 33, + 3 adds into the pixel sum;
#   per sample that hits: + hit point / normal 22 + normal->colour 9;
#   per pixel: row background 13 + mean and pack 6.
FLOP_SAMPLE_MISS = 36.0
FLOP_SAMPLE_HIT = 67.0
FLOP_PIXEL = 19.0
HIT_FRACTION_16_9 = 1832112.0 / 8294400.0   # sphere pixels in the reference's own 4K image


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["fast", "exact"], default="fast")
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def cpu_baseline(target_seconds: float):
    """Times the CPU oracle on the same frame at a bounded sample count; rank 0 only."""
    import oracle

    threads = max(1, min(16, os.cpu_count() or 1))
    # calibrate on a thin band, then size the sample count for about target_seconds
    _, _, _, ms = oracle.trace_paths(WIDTH, HEIGHT, spp=1, max_depth=8, seed=1, jitter=True, row_begin=1000,
                                     row_count=64, threads=threads)
    per_spp_s = ms * 1e-3 * HEIGHT / 64.0
    spp = max(1, min(1024, int(target_seconds / max(per_spp_s, 1e-6))))
    _, _, rays, ms = oracle.trace_paths(WIDTH, HEIGHT, spp=spp, max_depth=8, seed=1, jitter=True, threads=threads)
    out = {
        "value": WIDTH * HEIGHT * spp / (ms * 1e-3) / 1e6,
        "unit": "Mpaths/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{WIDTH}x{HEIGHT} default scene, {spp} of 1024 spp, jittered, row-parallel CPU oracle, {ms * 1e-3:.1f} s",
    }
    # the real reference (its own tile scheduler, 1 sample at the pixel centre), if its binary travelled
    if os.path.exists(oracle.REF_BINARY):
        try:
            with tempfile.TemporaryDirectory() as td:
                res = subprocess.run([oracle.REF_BINARY, "--width", str(WIDTH), "--height", str(HEIGHT), "--repeat", "3"],
                                     cwd=td, capture_output=True, text=True, timeout=120, check=True)
            kv = dict(tok.split("=") for tok in res.stdout.split())
            out["reference_1spp"] = {
                "value": WIDTH * HEIGHT / (float(kv["ref_trace_ms"]) * 1e-3) / 1e6,
                "unit": "Mpaths/s",
                "cores": int(kv["nproc"]),
                "kind": "reference",
                "sample": f"{WIDTH}x{HEIGHT}, 1 sample at pixel centre, reference software backend trace(), best of 3",
            }
        except Exception as exc:  # the baseline must never take the benchmark down
            out["reference_1spp"] = {"error": str(exc)[:200]}
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd.sharding import BandGather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    gather = BandGather(WIDTH, HEIGHT, device)
    flags = rt.FLAG_JITTER | (rt.FLAG_FAST if args.mode == "fast" else 0)
    back = rt.HipRaytracer(None, device=local_rank)
    if not back.setup(WIDTH, HEIGHT, spp=args.spp, max_depth=args.depth, seed=1, flags=flags, pass_=rt.PASS_PATH,
                      row_begin=gather.row_begin, row_count=gather.rows):
        sys.exit("setup failed: " + back.last_error())
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        back.trace_device(gather.local.data_ptr(), None, stream)
        return gather.gather()

    for _ in range(args.warmup):
        step()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        back.trace_device(gather.local.data_ptr(), None, stream)
        ev[i][1].record()
        image = gather.gather()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    t = torch.tensor([elapsed, kernel_ms * 1e-3], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_s = float(t[0]), float(t[1])

    if rank == 0:
        st = back.stats()
        paths_total = WIDTH * HEIGHT * args.spp
        ms_per_step = elapsed / args.steps * 1e3
        # flops of the band the slowest rank renders ~ total / world (bands are equal at 1,2,4,8)
        band_paths = paths_total / world
        band_pixels = WIDTH * HEIGHT / world
        flops = band_paths * (HIT_FRACTION_16_9 * FLOP_SAMPLE_HIT + (1 - HIT_FRACTION_16_9) * FLOP_SAMPLE_MISS) + band_pixels * FLOP_PIXEL
        achieved = flops / kernel_s / 1e12
        line = {
            "metric": "Mpaths/sec at 3840x2160",
            "value": paths_total / (elapsed / args.steps) / 1e6,
            "unit": "Mpaths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"default scene {WIDTH}x{HEIGHT}, {args.spp} spp jittered, max depth {args.depth}, "
                            f"row bands over {world} GPU(s), gather to rank 0",
                "arithmetic": args.mode,
                "rays_per_path": 1,
            },
            "kernel_ms": kernel_s * 1e3,
            "rays_per_s": paths_total / (elapsed / args.steps),
            "roofline": {
                "bound": "valu_fp32",
                "achieved": achieved,
                "peak": FP32_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP32_VECTOR_PEAK_TFLOPS,
                "traffic": None,
                "kernel": "path_kernel",
                "flop_per_launch": flops,
                "vgprs": st.vgprs,
                "lds_bytes": st.lds_bytes,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)

    back.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
