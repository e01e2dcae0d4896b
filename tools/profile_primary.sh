#!/bin/bash
# rocprofv3 evidence for the reference-pinned primary pass (tools/bench_primary.py): kernel trace,
# issue counters, HBM write bytes; each counter pass a run of its own.
# Usage (from the repo root on the GPU box): tools/profile_primary.sh <outdir-under-gpurun_out>
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-prof_primary}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 "$ROOT/tools/bench_primary.py" > "$OUT/trace.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -o pmc -- python3 "$ROOT/tools/bench_primary.py" --only-4k > "$OUT/pmc_sq.log" 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc -- python3 "$ROOT/tools/bench_primary.py" --only-4k > "$OUT/pmc_write.log" 2>&1 || exit 3
cd "$ROOT" && python3 tools/summarize_profile.py "$OUT" "$OUT/summary.txt" > /dev/null || exit 4
[ -n "$KEEP_DB" ] || rm -rf "$OUT"/trace "$OUT"/pmc_*
echo profile_ok
