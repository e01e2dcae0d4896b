#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box:
#   1. --kernel-trace --stats      per-kernel durations (no counters)
#   2..4 --pmc passes, one run each (counters never share a run with tracing domains)
# Usage (from the repo root on the GPU box): tools/profile_bench.sh <mode> <outdir-under-gpurun_out>
set -o pipefail
MODE=${1:-exact}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/${2:-prof_$MODE}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --mode $MODE --steps 20 --warmup 10 --no-cpu-baseline --no-aux --no-second-mode"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o pmc -- $BENCH > "$OUT/pmc_sq.log" 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d "$OUT/pmc_write" -o pmc -- $BENCH > "$OUT/pmc_write.log" 2>&1 || exit 4
echo profile_ok
