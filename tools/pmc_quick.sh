#!/bin/bash
# One SQ counter pass over a short bench.py run: tools/pmc_quick.sh <mode> <outdir-under-gpurun_out> [lib]
set -o pipefail
MODE=${1:-fast}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${2:-pmcq_$MODE}
export RAYTRACER_AMD_LIB=$3
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -o pmc -- python3 $ROOT/bench.py --mode $MODE --steps 10 --warmup 5 --no-cpu-baseline --no-aux --no-second-mode > "$OUT/pmc_sq.log" 2>&1 || exit 2
cd $ROOT && python3 tools/summarize_profile.py "$OUT" "$OUT/summary.txt" > /dev/null
cat "$OUT/summary.txt"
