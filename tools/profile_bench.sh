#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box:
#   1. --kernel-trace --stats      per-kernel durations (no counters)
#   2..5 --pmc passes, one run each (counters never share a run with tracing domains):
#        issue counters, executed FP32 operation counters, HBM read bytes, HBM write bytes
# and folds the counters into profiles/r03_counters.json (tools/collect_counters.py).
# Usage (from the repo root on the GPU box): tools/profile_bench.sh <mode> <outdir-under-gpurun_out> [walked]
# "walked": the run with AEHIP_FLAG_WALK_ALL (every sample of every pixel walked), counters under headline_<mode>_walked.
set -o pipefail
MODE=${1:-exact}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/${2:-prof_$MODE}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
STEPS=20; WARMUP=10
KEY=headline_$MODE; EXTRA=
if [ "$3" = walked ]; then KEY=${KEY}_walked; EXTRA=--walk-all; fi
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --mode $MODE --steps $STEPS --warmup $WARMUP --prewarm-ms 0 --no-cpu-baseline --no-aux --no-second-mode $EXTRA"
# the kernel trace runs bench.py as the driver does (its time-based warm-up, then the counted frames); the counter
# passes skip the time-based warm-up: they serialise launches anyway and the counts do not depend on the clock.
TRACED="python3 $ROOT/bench.py --mode $MODE --no-cpu-baseline --no-aux --no-second-mode $EXTRA"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $TRACED > "$OUT/trace.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o pmc -- $BENCH > "$OUT/pmc_sq.log" 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d "$OUT/pmc_flops" -o pmc -- $BENCH > "$OUT/pmc_flops.log" 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc -- $BENCH > "$OUT/pmc_write.log" 2>&1 || exit 4
cd "$ROOT" && mkdir -p gpurun_out/profiles_out && python3 tools/collect_counters.py $KEY auto "$OUT" || exit 6
cp profiles/r03_counters.json gpurun_out/profiles_out/r03_counters.json
python3 tools/summarize_profile.py "$OUT" "$OUT/summary.txt" > /dev/null || exit 7
# the rocpd databases are tens of MiB a run: keep the summaries, drop the databases unless asked
[ -n "$KEEP_DB" ] || rm -rf "$OUT"/trace "$OUT"/pmc_*
echo profile_ok
