#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s rendering the default scene at 3840x2160, 1024 samples per pixel.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode fast|exact] [--spp S]

One step = one full frame: every rank renders its row band with the HIP camera/path kernel
(scene and framebuffer resident in HBM), then the bands are gathered to rank 0 (RCCL) when
N > 1.  Total work per step is fixed (one 4K frame), so scaling over N is "strong".  Rank 0
prints ONE JSON line.  With N > 1 the driver launches this file under torch.distributed.run,
one rank per GPU.

Clocks: the GPU raises its clocks only after a few milliseconds of load, and a frame lasts a
fraction of a millisecond.  Before the W counted warm-up frames every timed run therefore renders
frames for a fixed time (--prewarm-ms, reported as prewarm_ms); K and W stay as passed.

Arithmetic: `value` is measured with the fast build (explicit FMA, v_rcp/v_rsq/v_sqrt), whose
1024-spp image is held to a PSNR bound against the bit-exact build and the CPU oracle
(tests/test_gpu_parity.py).  The same frame rendered by the exact build (bit-identical to the
oracle, and to the reference wherever the reference defines the result) is timed in the same
run and reported under "exact".  --mode exact swaps the two.

What `value` counts: nominal paths (W x H x spp) per second.  The single-sphere camera pass decides
the pixels whose samples all miss from bounds over the pixel's sample positions and never walks
their samples; `samples_walked_per_frame` (counted on the device with the kernels' own
classification, aehip_stats.samples_walked), `mpaths_walked_per_s` and `rays_per_s` give the rate of
the samples that did go through a sample loop, and "every_sample_walked" is the same frame with that
decision switched off (AEHIP_FLAG_WALK_ALL) -- the like-for-like figure for other scenes.

The line also carries
  roofline      the FP32 operations the kernel EXECUTES per launch (rocprofv3 counters
                SQ_INSTS_VALU_FLOPS_FP32 and SQ_INSTS_VALU_TRANS_F32 of the same command,
                profiles/r03_counters.json: 64 x (FLOPS_FP32 + TRANS)) / its mean launch time (HIP events on
                the launch stream, inside the timed region) against the FP32 vector peak -- a fraction that
                cannot exceed 1; the flops of the reference's formulas for the same frame are reported
                beside it as algorithmic_tflops / work_reduction,
  cpu_baseline  the CPU oracle (a port of the reference's pixel kernel extended with sampling)
                timed on this host's cores on a bounded sample of the same frame, plus the real
                reference binary's 1-sample figure where oracle/_ref was shipped,
  aux           (N=1, unless --no-aux) the reference-pinned primary pass at 4K against HBM store
                bandwidth, the host-buffer call's wall time, and the benchmark's other configurations at the
                sizes it names: 1920x1080 / 64 spp / depth 8 on the default scene, the 10,000-sphere
                brute-force scene and the depth-50 mixed-material scene at 3840x2160, each against the FP32
                vector peak from counters of the same launches (mean of settled launches, not best-of),
  summary       LAST in the line, compact (the driver keeps the line's tail): exact / every-sample-walked /
                other configurations as (value, ms, kernel ms, fraction), and the verified flags.

With several ranks consecutive frames go to two streams in turn and the gather of a frame runs
beside the next frame's rendering (raytracer_amd/sharding.py); a rank sends the columns that hold
more than the row's background and one background pixel a row, rank 0 rebuilds whole rows ("gather"
in the line names the columns and the bytes a peer sends a frame).  --graph: a rank's frame (render through the
allocation-free entry point, then the pack of its kept columns) is captured once into a HIP graph and
replayed per frame, the collective staying outside the graph (on this runtime a replay costs the host as much
as the plain launches or more -- 0.073 against 0.055 ms a frame on one box, 0.053 against 0.052 on another,
tools/host_overhead.py -- so plain launches are the default).
Rehearsals on a one-GPU box: BENCH_FORCE_COLLECTIVE=1 runs the multi-rank loop with a one-rank RCCL group;
BENCH_SHARE_GPUS=1 under torch.distributed.run lets several ranks share the GPU, with gloo in RCCL's place (the same
control flow: every rank must make the same collective calls in the same order).
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
# tools/valu_peak.hip (profiles/r02_valu_peak.txt): back-to-back v_fma_f32 / v_mul_f32 / v_add_f32 issue every
# 2.19 shader clocks per SIMD at the 2.32 GHz the chip holds under that load: 69.4 T lane-operations/s.
# Conversions, compares, 3-operand integer operations and anything with a scalar-register operand issue at
# half that rate (4.22 clocks), v_rcp / v_sqrt / v_rsq at a quarter (8.24).
VALU_FULL_RATE_MEASURED = 69.4e12
CYCLES_FULL_RATE, CYCLES_HALF_RATE, CYCLES_QUARTER_RATE = 2.19, 4.22, 8.24
SIMDS = 256 * 4
SUSTAINED_CLOCK_HZ = 2.32e9
HBM_PEAK_GBS = 8000.0
COUNTERS_FILE = os.path.join(ROOT, "profiles", "r03_counters.json")
KERNEL_SOURCES = ["raytracer_amd/csrc/hip_raytracer_kernels.hip", "raytracer_amd/csrc/rt_exact_math.h"]
SETUP_KERNELS = ("ring_pixels_kernel", "walked_pixels_kernel", "row_miss_sums_kernel", "tile_cost_kernel")

# the benchmark's other configurations at the sizes it names, shared with tools/bench_scenes.py (the counters'
# workload keys are "<name>_<mode>"): name -> (width, height, spp, depth, seed, scene)
CONFIG_CASES = {
    "config2_1080p_64spp_d8": (1920, 1080, 64, 8, 1, "default"),
    "stress10k_4k_2spp_d1": (3840, 2160, 2, 1, 1, "stress10k"),
    "stress10k_4k_4spp_d4": (3840, 2160, 4, 4, 1, "stress10k"),
    "divergence_4k_16spp_d50": (3840, 2160, 16, 50, 7, "divergence"),
}


def load_counters():
    """profiles/r03_counters.json if it was measured on the kernels this tree holds, else the reason it is unusable."""
    import hashlib

    try:
        with open(COUNTERS_FILE) as f:
            doc = json.load(f)
    except (OSError, ValueError) as exc:
        return None, f"no counter file: {exc}"
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    if doc.get("kernel_source_sha256") != h.hexdigest():
        return None, "profiles/r03_counters.json was collected on another version of the kernel sources: re-run tools/profile_bench.sh"
    return doc, None


def executed_work(doc, key, seconds_per_frame, share=1.0):
    """Executed FP32 rate of workload `key` from the committed counters and this run's frame time.
    `share`: fraction of the profiled frame one rank renders (tiles dealt round-robin: same mix of rows)."""
    w = (doc or {}).get("workloads", {}).get(key)
    if not w:
        return None
    flop = valu = slots = cycles = busy_us = 0.0
    kernels = []
    for name, k in w["kernels"].items():
        # kernels of the timed frame only: set-up's (the ring list, the walked-pixel count, the row table) run once per backend
        if "executed_fp32_flop_per_frame" not in k or k.get("dispatches_per_frame", 1.0) < 0.5 or name in SETUP_KERNELS:
            continue
        flop += k["executed_fp32_flop_per_frame"]
        n = k.get("SQ_INSTS_VALU", 0.0)               # the file holds per-frame totals
        valu += n
        # issue cycles these instructions take at the rates each class reaches alone (tools/valu_peak.hip):
        # quarter-rate transcendentals, half-rate conversions, everything else priced at full rate.  A lower
        # bound: compares, scalar-register operands and a transcendental among other instructions cost more.
        trans = k.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        cvt = k.get("SQ_INSTS_VALU_CVT", 0.0)
        slots += (n - trans - cvt) * CYCLES_FULL_RATE + trans * CYCLES_QUARTER_RATE + cvt * CYCLES_HALF_RATE
        kernels.append(name)
        # shader clock while the kernel ran: GRBM_GUI_ACTIVE (summed over the 8 XCDs) over the launch's duration
        # in the same counter pass; launches under 0.2 ms are left out (the counter also runs between launches)
        us = k.get("avg_dispatch_us", {}).get("pmc_flops", 0.0) * k.get("dispatches_per_frame", 1.0)
        if k.get("GRBM_GUI_ACTIVE") and us >= 200.0:
            cycles += k["GRBM_GUI_ACTIVE"] / 8.0
            busy_us += us
    if flop <= 0.0:
        return None
    flop *= share
    valu *= share
    out = {"fp32_flop_per_frame": flop, "tflops": flop / seconds_per_frame / 1e12, "kernels": sorted(kernels),
           "source": f"profiles/r03_counters.json:{key}"}
    if valu > 0.0:
        lane_ops = valu * 64.0 / seconds_per_frame
        out["valu_wave_instructions_per_frame"] = valu
        out["valu_lane_ops_per_s"] = lane_ops
        out["valu_issue_vs_measured_full_rate"] = lane_ops / VALU_FULL_RATE_MEASURED
        out["valu_issue_cycles_lower_bound_frac"] = slots / (SIMDS * SUSTAINED_CLOCK_HZ * seconds_per_frame)
        # The shader clock of THIS run is not known here (2.0 GHz on a cold chip, 2.3-2.4 GHz in a settled run of the
        # headline: tools/wave_timeline.py stamps it inside the kernel).  What is known: the clock below which the
        # launch's vector instructions could not have issued in the time the launch took.
        out["valu_issue_cycles_per_simd"] = slots / SIMDS
        out["shader_clock_ghz_at_which_issue_would_fill_the_launch"] = slots / SIMDS / seconds_per_frame / 1e9
    if busy_us > 0.0:
        # clock of the COUNTER pass (profiled launches, no pre-warm): not this run's
        out["counter_pass_shader_clock_ghz"] = cycles / busy_us / 1e3
    return out


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
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="every timed run first renders frames for this long, so that its frames run at settled clocks")
    ap.add_argument("--mode", choices=["fast", "exact"], default="fast")
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true")
    ap.add_argument("--no-second-mode", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay a rank's frame (render + pack) from a captured HIP graph instead of launching it (measured: a replay "
                         "costs the host more than the two launches it replaces, tools/host_overhead.py, so it is not the default)")
    ap.add_argument("--walk-all", action="store_true",
                    help="measurement aid: the timed run walks every sample of every pixel (AEHIP_FLAG_WALK_ALL)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--watchdog", type=int, default=900,
                    help="seconds after which a rank dumps its Python stacks and exits (0 = off)")
    return ap.parse_args()


def cpu_baseline(target_seconds: float):
    """Times the CPU oracle on the same frame at a bounded sample count; rank 0 only."""
    import oracle

    threads = max(1, min(16, os.cpu_count() or 1))
    # calibrate on a thin band, then size the sample count for about target_seconds
    _, _, _, ms = oracle.trace_paths(WIDTH, HEIGHT, spp=4, max_depth=8, seed=1, jitter=True, row_begin=800,
                                     row_count=512, threads=threads)
    per_spp_s = ms * 1e-3 * HEIGHT / 512.0 / 4.0
    spp = max(1, min(1024, int(target_seconds / max(per_spp_s, 1e-6))))
    _, _, rays, ms = oracle.trace_paths(WIDTH, HEIGHT, spp=spp, max_depth=8, seed=1, jitter=True, threads=threads)
    out = {
        "value": WIDTH * HEIGHT * spp / (ms * 1e-3) / 1e6,
        "unit": "Mpaths/s",
        "cores": threads, "host_cpus_online": os.cpu_count(),
        "kind": "port",
        "sample": f"{WIDTH}x{HEIGHT} default scene, {spp} of 1024 spp, jittered, row-parallel CPU oracle (walks every sample), {ms * 1e-3:.1f} s",
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
    else:
        out["reference_1spp"] = {"error": "oracle/_ref/raytracer_ref not built (make -C oracle ref needs the reference checkout)"}
    return out


def device_facts(torch, index):
    """What the peak figures assume: 256 CUs at 2.4 GHz."""
    p = torch.cuda.get_device_properties(index)
    facts = {"name": p.name, "arch": getattr(p, "gcnArchName", ""), "compute_units": p.multi_processor_count,
             "hbm_gib": round(p.total_memory / 2 ** 30, 1), "uuid": str(getattr(p, "uuid", ""))}
    try:
        facts["clock_mhz"] = torch.cuda.clock_rate(index)       # current shader clock, from the SMI library
    except Exception:
        pass
    return facts


def frame_flops(paths, pixels):
    return paths * (HIT_FRACTION_16_9 * FLOP_SAMPLE_HIT + (1 - HIT_FRACTION_16_9) * FLOP_SAMPLE_MISS) + pixels * FLOP_PIXEL


def sig(x, digits=4):
    """A float at `digits` significant digits (the summary must fit the tail of the line the driver keeps)."""
    if x is None:
        return None
    return float(f"{x:.{digits}g}")


def main():
    args = parse()
    # stdout carries the one JSON line and nothing else: libraries that write to file descriptor 1
    # (RCCL prints a version banner when its first communicator comes up) go to stderr instead
    result_fd = os.dup(1)
    os.dup2(2, 1)
    if args.watchdog > 0:
        # a rank stuck in a collective leaves a stack trace and ends instead of holding the node
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)
    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd.sharding import BandGather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU.  BENCH_SHARE_GPUS=1 (a rehearsal of the multi-rank control flow on a box with fewer GPUs than
    # ranks): ranks share devices, and the bands are gathered as host tensors over gloo (every frame copied to the
    # host first), since RCCL needs a GPU of its own for every rank.  (gloo's gather of DEVICE tensors between
    # processes that share a GPU completes in most runs and, in some, never returns from wait() on any rank --
    # tools/share_gpu_rehearsal.py -- so the rehearsal keeps it out of its way.)
    share = os.environ.get("BENCH_SHARE_GPUS") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # BENCH_FORCE_COLLECTIVE=1: the multi-rank code path (process group, two render streams,
    # stream-ordered gather, row reorder, barriers) with however many ranks there are -- a one-rank
    # RCCL group rehearses it on a one-GPU box
    grouped = world > 1 or os.environ.get("BENCH_FORCE_COLLECTIVE") == "1"
    if grouped:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    # everything runs on side streams: the library treats the null stream as "use your own and
    # wait", and launches are to queue back to back without the host in between.  With several
    # ranks a frame is a fraction of a millisecond and its ramp and tail weigh: consecutive frames
    # go to two streams in turn, so frame i+1 fills the compute units that frame i's last
    # workgroups leave idle (tools/frame_overlap.py).  One rank keeps one stream, so that a launch's
    # duration means the same here and in the profiler's kernel trace.
    sides = [torch.cuda.Stream(device=device) for _ in range(2 if grouped else 1)]
    torch.cuda.set_stream(sides[0])
    stream = sides[0].cuda_stream
    handles = [s.cuda_stream for s in sides]
    gather = BandGather(WIDTH, HEIGHT, torch.device("cpu") if share else device, always_collective=grouped)
    collective_device = torch.device("cpu") if share else device       # where the loop's own small all-reduces live
    host_frames = share             # the rehearsal: render into a device buffer, hand the band over as a host tensor
    device_bands = [torch.zeros((gather.max_rows, WIDTH), dtype=torch.int32, device=device) for _ in range(2)] if host_frames else None

    verified = {}
    kept_columns = {}
    graphed = {}

    def timed_run(mode, steps, warmup, walk_all=False):
        """K steps of render-band + gather; returns a dict: elapsed_s and kernel_s (max over ranks), stats, prewarm_ms.
        walk_all: every sample of every pixel goes through the sample loop (AEHIP_FLAG_WALK_ALL)."""
        name = mode + ("_every_sample_walked" if walk_all else "")
        flags = rt.FLAG_JITTER | (rt.FLAG_FAST if mode == "fast" else 0) | (rt.FLAG_WALK_ALL if walk_all else 0)
        back = rt.HipRaytracer(None, device=dev_index)
        # frames are timed by this loop's own events: the library records none of its own (FLAG_NO_TIMING)
        if not back.setup(WIDTH, HEIGHT, spp=args.spp, max_depth=args.depth, seed=1, flags=flags | rt.FLAG_NO_TIMING,
                          pass_=rt.PASS_PATH, **gather.band):
            sys.exit("setup failed: " + back.last_error())
        # what crosses the links: the columns outside of which a band row is one colour stay at home (every rank's
        # library derives the same columns from the scene, the camera and the frame's size; set_columns checks that)
        c0, c1 = back.background_columns()
        narrow = gather.set_columns(c0, c1)
        kept_columns[name] = [c0, c1] if narrow else None

        # A rank's frame as a HIP graph, one per band buffer: the render through the entry point that
        # allocates nothing, then the pack of the kept columns; replayed on the stream of the frame's turn.
        # (capture mode "thread_local": the process group's watchdog thread may touch the runtime meanwhile)
        graphs = None
        # (one plain launch first: the library reads the kernel's resource figures outside capture only)
        back.trace_device_static((device_bands[0] if host_frames else gather.locals[0]).data_ptr(), None, stream)
        torch.cuda.synchronize()
        if args.graph and not host_frames:
            graphs = []
            for slot in range(2 if grouped else 1):
                side = sides[slot % len(sides)]
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    back.trace_device_static(gather.locals[slot].data_ptr(), None, torch.cuda.current_stream().cuda_stream)
                    if grouped:
                        gather.pack(slot, torch.cuda.current_stream().cuda_stream)
                graphs.append(g)
            torch.cuda.set_stream(sides[0])
        graphed[name] = graphs is not None

        def frame(i, first=None, last=None):
            side = sides[i % len(sides)]
            torch.cuda.set_stream(side)
            band = gather.begin(i, side)
            target = device_bands[i & 1] if host_frames else band
            if first is not None:
                first.record()
            if graphs is not None:
                graphs[i % len(graphs)].replay()
            else:
                back.trace_device(target.data_ptr(), None, handles[i % len(sides)])
            if last is not None:
                last.record()
            if host_frames:
                band.copy_(target)          # (waits for the render)
            gather.submit(i, side, packed=graphs is not None and grouped)

        def drain():
            torch.cuda.set_stream(sides[0])
            gather.finish()
            torch.cuda.synchronize()

        # settled clocks: frames for a fixed time, then the counted warm-up.  Every frame is a collective, so the
        # ranks must render the same number of them: they go round after round until no rank's clock says "more"
        t_pre = time.perf_counter()
        n_pre = 0
        more = args.prewarm_ms > 0.0
        while more:
            for _ in range(16):         # (an even number of frames a round: frame numbering starts on buffer 0 again)
                frame(n_pre)
                n_pre += 1
            torch.cuda.synchronize()
            more = (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms
            if grouped:
                flag = torch.tensor([1 if more else 0], dtype=torch.int32, device=collective_device)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                more = bool(flag.item())
        drain()
        prewarm_ms = (time.perf_counter() - t_pre) * 1e3
        for i in range(warmup + (warmup & 1)):
            frame(i)
        drain()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        # two streams: only the frames' common span is read below, so only its two ends are recorded
        # (at 8 ranks a frame lasts 50 us and the host's work per frame is what bounds the rate)
        every = len(sides) == 1
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            frame(i, ev[i][0] if every or i == 0 else None, ev[i][1] if every or i == steps - 1 else None)
        torch.cuda.set_stream(sides[0])
        gather.finish()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if len(sides) > 1:
            # launches overlap: a launch's own span counts the time it shared the GPU, so take the
            # frames' common span per frame instead
            kernel_ms = ev[0][0].elapsed_time(ev[-1][1]) / steps
        else:
            kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
        mine = kernel_ms
        t = torch.tensor([elapsed, kernel_ms * 1e-3], dtype=torch.float64, device=collective_device)
        if grouped:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        st = back.stats()
        per_rank = None
        if grouped:
            # every rank's device and its own kernel time: that RCCL saw `world` ranks on `world` distinct GPUs
            per_rank = [None] * world
            dist.all_gather_object(per_rank, {"rank": rank, "uuid": device_facts(torch, dev_index)["uuid"], "kernel_ms": mine,
                                              "samples_walked": int(st.samples_walked)})
        graphs = None
        back.close()
        # outside the timed region: the gathered frame must be the frame one GPU renders alone
        # (sample keys depend on the absolute pixel only, so the split cannot change a bit)
        if rank == 0:
            whole = rt.HipRaytracer(None, device=dev_index)
            whole.setup(WIDTH, HEIGHT, spp=args.spp, max_depth=args.depth, seed=1, flags=flags, pass_=rt.PASS_PATH)
            ref = torch.empty((HEIGHT, WIDTH), dtype=torch.int32, device=device)
            torch.cuda.synchronize()
            whole.trace_device(ref.data_ptr(), None, stream)
            torch.cuda.synchronize()
            whole.close()
            got = gather.image.to(device)
            verified[name] = same = bool(torch.equal(ref, got))
            if not same:
                ne = ref != got
                idx = ne.nonzero()[:6].tolist()
                print("MISMATCH", mode, int(ne.sum()), idx, [(hex(ref[y, x].item() & 0xffffffff), hex(got[y, x].item() & 0xffffffff)) for y, x in idx], file=sys.stderr, flush=True)
        walked = sum(r["samples_walked"] for r in per_rank) if per_rank else int(st.samples_walked)
        return {"elapsed_s": float(t[0]), "kernel_s": float(t[1]), "stats": st, "steps": steps, "prewarm_ms": prewarm_ms,
                "samples_walked": walked, "per_rank": per_rank}

    other = "exact" if args.mode == "fast" else "fast"
    head = timed_run(args.mode, args.steps, args.warmup, walk_all=args.walk_all)
    second = None if args.no_second_mode else timed_run(other, max(2, args.steps // 2), 2)
    # the same frames with every sample of every pixel walked: what the pixels decided by bounds save
    walked = {} if args.no_second_mode else {m: timed_run(m, max(2, args.steps // 4), 2, walk_all=True) for m in (args.mode, other)}

    if rank == 0:
        paths_total = WIDTH * HEIGHT * args.spp
        counters, counters_error = load_counters()
        headline = args.spp == 1024 and args.depth == 8          # the workload the counters were collected on

        def roofline_of(mode, run, walked_run=False):
            """Executed FP32 work (counters) over this run's kernel time; the reference formulas' flops beside it."""
            seconds, stats = run["kernel_s"], run["stats"]
            # every rank renders 1/world of the frame, the same mix of rows (tiles dealt round-robin)
            algorithmic = frame_flops(paths_total / world, WIDTH * HEIGHT / world) / seconds / 1e12
            key = f"headline_{mode}" + ("_walked" if walked_run else "")
            ex = executed_work(counters, key, seconds, 1.0 / world) if headline else None
            r = {"bound": "valu_fp32", "achieved": ex["tflops"] if ex else None, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": ex["tflops"] / FP32_VECTOR_PEAK_TFLOPS if ex else None, "traffic": None,
                 "kernel": "camera_hybrid_kernel" if mode == "fast" else "camera_kernel", "counters": key,
                 "algorithmic_tflops": algorithmic, "vgprs": stats.vgprs, "lds_bytes": stats.lds_bytes}
            if ex:
                r["work_reduction"] = algorithmic / ex["tflops"]
                r["valu_issue_cycles_lower_bound_frac"] = ex.get("valu_issue_cycles_lower_bound_frac")
                r["shader_clock_ghz_at_which_issue_would_fill_the_launch"] = ex.get("shader_clock_ghz_at_which_issue_would_fill_the_launch")
                r["executed"] = ex
                k = counters["workloads"][key]["kernels"].get(r["kernel"], {})
                if "WRITE_SIZE" in k and "FETCH_SIZE" in k:
                    # HBM bytes per launch: WRITE_SIZE in KiB; FETCH_SIZE in KiB, x2 on gfx950 (it under-reports wide reads)
                    r["traffic"] = int((k["WRITE_SIZE"] * 1024 + k["FETCH_SIZE"] * 2048) / world)
                r["note"] = ("achieved = 64 x (SQ_INSTS_VALU_FLOPS_FP32 + TRANS_F32) per launch (committed rocprofv3 counters of this "
                             "command) / this run's launch time; algorithmic_tflops prices the reference's formulas for every nominal sample")
            else:
                r["note"] = counters_error or "executed-operation counters exist for the default workload (1024 spp, depth 8) only"
            return r

        def four(run, roof):
            """value, ms per step, kernel ms, fraction: one run in the summary."""
            per = run["elapsed_s"] / run["steps"]
            return {"value": sig(paths_total / per / 1e6, 5), "ms_per_step": sig(per * 1e3), "kernel_ms": sig(run["kernel_s"] * 1e3),
                    "frac": sig(roof["frac"])}

        per_step = head["elapsed_s"] / args.steps
        head_roof = roofline_of(args.mode, head, args.walk_all)
        line = {
            "metric": "Mpaths/sec at 3840x2160",
            "value": paths_total / per_step / 1e6,
            "unit": "Mpaths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": per_step * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"default scene {WIDTH}x{HEIGHT}, {args.spp} spp jittered, max depth {args.depth}, "
                            + (f"4-row tiles dealt round-robin to {world} GPUs, gather to rank 0" if world > 1 else "one GPU"),
                "arithmetic": args.mode,
                "rays_per_path": 1,
            },
            "kernel_ms": head["kernel_s"] * 1e3,
            "prewarm_ms": head["prewarm_ms"],
            "frame_as_hip_graph": graphed,
            # the samples that went through a sample loop (the rest were decided from bounds over their pixel), per second
            "samples_walked_per_frame": head["samples_walked"],
            "mpaths_walked_per_s": head["samples_walked"] / per_step / 1e6,
            "rays_per_s": head["samples_walked"] / per_step,
            "rays_per_s_nominal": paths_total / per_step,
            "roofline": head_roof,
        }
        if args.walk_all:
            line["config"]["workload"] += ", every sample of every pixel walked (AEHIP_FLAG_WALK_ALL)"
        line["roofline"]["measured_full_rate_valu_lane_ops_per_s"] = VALU_FULL_RATE_MEASURED
        summary = {"headline": four(head, head_roof)}
        if second is not None:
            roof2 = roofline_of(other, second)
            line[other] = dict(four(second, roof2), unit="Mpaths/s", steps=second["steps"], roofline=roof2)
            summary[other] = four(second, roof2)
            line["roofline"][f"{other}_frac"] = roof2["frac"]            # (scalars here survive the driver's parse)
            line["roofline"][f"{other}_kernel_ms"] = second["kernel_s"] * 1e3
        if walked:
            line["every_sample_walked"] = {}
            summary["every_sample_walked"] = {}
            for m, run in walked.items():
                roofw = roofline_of(m, run, True)
                line["every_sample_walked"][m] = dict(four(run, roofw), unit="Mpaths/s", steps=run["steps"], roofline=roofw)
                summary["every_sample_walked"][m] = four(run, roofw)
                line["roofline"][f"walked_{m}_frac"] = roofw["frac"]
                line["roofline"][f"walked_{m}_mpaths_per_s"] = paths_total / (run["elapsed_s"] / run["steps"]) / 1e6
            line["every_sample_walked"]["note"] = (
                "AEHIP_FLAG_WALK_ALL: no pixel is decided from bounds over its sample positions; exact build: the same "
                "image bit for bit (tests/test_gpu_parity.py)")
        line["device"] = device_facts(torch, dev_index)
        line["gather"] = {"ranks": world, "band_rows": gather.max_rows,
                          "kept_columns": kept_columns,       # per run: the columns a rank sends (plus one background pixel a row), None = whole rows
                          "bytes_per_peer_per_frame": {k: gather.max_rows * 4 * ((v[1] - v[0] + 1) if v else WIDTH) for k, v in kept_columns.items()}}
        if head["per_rank"]:
            ranks = head["per_rank"]
            line["rccl"] = {"rccl_ranks": len(ranks), "backend": dist.get_backend(), "distinct_devices": len({r["uuid"] for r in ranks}),
                            "kernel_ms_min": min(r["kernel_ms"] for r in ranks), "kernel_ms_max": max(r["kernel_ms"] for r in ranks),
                            "ranks": ranks}
            summary["rccl"] = {"rccl_ranks": len(ranks), "backend": dist.get_backend(), "distinct_devices": len({r["uuid"] for r in ranks}),
                               "kernel_ms_min": sig(min(r["kernel_ms"] for r in ranks)), "kernel_ms_max": sig(max(r["kernel_ms"] for r in ranks))}
        if world == 1 and not args.no_aux:
            line["aux"] = aux_measurements(rt, torch, stream, counters, args.prewarm_ms)
            summary["configs"] = {k: v["summary"] for k, v in line["aux"].items() if "summary" in v}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        # last, and small: the driver's record keeps the end of the line
        summary["samples_walked_per_frame"] = head["samples_walked"]
        summary["mpaths_walked_per_s"] = sig(head["samples_walked"] / per_step / 1e6, 5)
        summary["prewarm_ms"] = sig(head["prewarm_ms"])
        summary["gathered_frame_equals_single_gpu_frame"] = verified
        line["summary"] = summary
        os.write(result_fd, (json.dumps(line) + "\n").encode())

    if grouped:
        dist.destroy_process_group()


def settled_launches(back, img, torch, stream, prewarm_ms, launches=3):
    """Mean kernel time of `launches` back-to-back launches (the library's own events around each) after launches
    for prewarm_ms have brought the clocks up -- the rule the counters' launches follow too (a mean, not a best-of).
    Returns (mean ms, stats of the last launch)."""
    t0 = time.perf_counter()
    while True:
        back.trace_device(img.data_ptr(), None, stream)
        torch.cuda.synchronize()
        if (time.perf_counter() - t0) * 1e3 >= prewarm_ms:
            break
    total, st = 0.0, None
    for _ in range(launches):
        back.trace_device(img.data_ptr(), None, stream)
        torch.cuda.synchronize()
        st = back.stats()
        total += st.kernel_ms
    return total / launches, st


def aux_measurements(rt, torch, stream, counters, prewarm_ms):
    """Secondary kernels on one GPU, each against the roofline that bounds it."""
    from raytracer_amd import scenes

    out = {}
    # the reference-pinned pass: 1 centred ray per pixel; HBM-store / launch bound (4 B per pixel)
    back = rt.HipRaytracer(None)
    back.setup(WIDTH, HEIGHT, pass_=rt.PASS_PRIMARY)
    argb = torch.zeros((HEIGHT, WIDTH), dtype=torch.int32, device="cuda")
    reps = 50
    for _ in range(200):
        back.trace_device(argb.data_ptr(), None, stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        back.trace_device(argb.data_ptr(), None, stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    back.close()
    nbytes = WIDTH * HEIGHT * 4
    out["primary_pass_4k_exact"] = {
        "ms_per_frame": ms, "mpaths_per_s": WIDTH * HEIGHT / ms / 1e3,
        "roofline": {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": nbytes,
                     "note": "33 MB store per launch; launches back to back, so the figure includes launch gaps"},
    }
    # the drop-in call itself: setup once, trace() into host memory (render in parts, device ->
    # pinned staging -> the caller's buffer overlapped with it); PCIe-inclusive, never `value`
    import numpy as np
    for mode in ("fast", "exact"):
        fb = np.zeros((HEIGHT, WIDTH), dtype=np.uint32)
        back = rt.HipRaytracer(fb)
        back.setup(WIDTH, HEIGHT, spp=1024, max_depth=8, seed=1, flags=rt.FLAG_JITTER | (rt.FLAG_FAST if mode == "fast" else 0),
                   pass_=rt.PASS_PATH)
        for _ in range(3):
            back.trace()
        calls = 10 if mode == "fast" else 5
        t0 = time.perf_counter()
        for _ in range(calls):
            back.trace()
        wall = (time.perf_counter() - t0) / calls * 1e3
        st = back.stats()
        back.close()
        out[f"host_buffer_call_4k_1024spp_{mode}"] = {
            "ms_per_call": wall, "mpaths_per_s": WIDTH * HEIGHT * 1024 / wall / 1e3, "kernel_ms": st.kernel_ms,
            "copy_tail_ms": st.copy_ms, "note": "aehip_backend_trace: wall time seen by the caller, frame delivered in host memory",
        }
    # the benchmark's other configurations at the sizes it names.  The brute-force list: every ray tests every sphere,
    # 18 flop per test as the reference writes it (depth 1: camera rays only, the table of per-sphere camera
    # constants, 4 operations a test; depth 4: the streamed pass, rays from anywhere, 10 operations a test).
    # The divergence scene: lane utilisation of the bounce loop with and without path regeneration.
    made = {"default": None, "stress10k": scenes.stress_scene(10000, 42), "divergence": scenes.divergence_scene(300, 7)}
    for name, (w, h, spp, depth, seed, scene) in CONFIG_CASES.items():
        sph = made[scene]
        variants = [("fast", True), ("exact", True)] + ([("fast", False)] if scene == "divergence" else [])
        for mode, regen in variants:
            back = rt.HipRaytracer(None)
            flags = rt.FLAG_JITTER | rt.FLAG_COUNT_RAYS | (rt.FLAG_FAST if mode == "fast" else 0) | (0 if regen else rt.FLAG_NO_REGEN)
            if not back.setup(w, h, spp=spp, max_depth=depth, seed=seed, flags=flags, pass_=rt.PASS_PATH, spheres=sph):
                out[f"{name}_{mode}"] = {"error": back.last_error()}
                back.close()
                continue
            img = torch.zeros((h, w), dtype=torch.int32, device="cuda")
            ms, s = settled_launches(back, img, torch, stream, prewarm_ms, launches=3 if regen else 2)
            if ms < 1.0:
                # A launch of tens of microseconds: the library's pair of timing events adds several microseconds of its
                # own (the profiler's trace of the 1920x1080 64-sample frame says 24-26 us where the pair says 33).  Such
                # frames are queued back to back without library events (AEHIP_FLAG_NO_TIMING) between ONE pair of
                # events and the span is divided: a frame's time as a caller that queues frames sees it.
                quick = rt.HipRaytracer(None)
                if quick.setup(w, h, spp=spp, max_depth=depth, seed=seed, flags=flags | rt.FLAG_NO_TIMING, pass_=rt.PASS_PATH, spheres=sph):
                    batch = 200
                    for _ in range(50):
                        quick.trace_device(img.data_ptr(), None, stream)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(batch):
                        quick.trace_device(img.data_ptr(), None, stream)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / batch
                quick.close()
            n_spheres = 1 if sph is None else len(sph)
            entry = {"kernel_ms": ms, "mpaths_per_s": s.paths / ms / 1e3, "rays_per_s": s.rays / (ms * 1e-3),
                     "rays_per_path": s.rays / s.paths, "lane_utilisation": s.rays / s.lane_slots if s.lane_slots else None,
                     "samples_walked": int(s.samples_walked), "spheres": n_spheres}
            if scene == "stress10k":
                tests = s.rays * n_spheres
                entry["sphere_tests_per_s"] = tests / (ms * 1e-3)
                entry["algorithmic_tflops"] = tests * 18.0 / (ms * 1e-3) / 1e12        # the reference's quadratic: 18 flop a test
            ex = executed_work(counters, f"{name}_{mode}", ms * 1e-3) if regen else None
            if ex:
                entry["roofline"] = {"bound": "valu_fp32", "achieved": ex["tflops"], "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": ex["tflops"] / FP32_VECTOR_PEAK_TFLOPS, "executed": ex}
            entry["summary"] = {"mpaths_per_s": sig(entry["mpaths_per_s"], 5), "kernel_ms": sig(ms),
                                "frac": sig(ex["tflops"] / FP32_VECTOR_PEAK_TFLOPS) if ex else None,
                                "lane_util": sig(entry["lane_utilisation"], 3)}
            out[f"{name}_{mode}" + ("" if regen else "_no_regeneration")] = entry
            back.close()
    return out


if __name__ == "__main__":
    main()
