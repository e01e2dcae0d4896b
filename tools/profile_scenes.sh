#!/bin/bash
# rocprofv3 kernel trace of tools/bench_scenes.py (the shape-list scenes) on the GPU box.
# Usage (from the repo root on the GPU box): tools/profile_scenes.sh <mode> <outdir-under-gpurun_out>
set -o pipefail
MODE=${1:-fast}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${2:-prof_scenes_$MODE}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 "$ROOT/tools/bench_scenes.py" --mode "$MODE" --reps 4 \
    --case stress10k_1080p_2spp_d1 stress10k_1080p_4spp_d4 stress2k_1080p_16spp_d4 divergence_1080p_16spp_d50 showcase_1080p_64spp_d8 default_1080p_64spp_d8 \
    > "$OUT/trace.log" 2>&1 || exit 1
# counters in a run of their own (never together with tracing domains)
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o pmc -- python3 "$ROOT/tools/bench_scenes.py" --mode "$MODE" --reps 2 \
    --case stress10k_1080p_2spp_d1 stress10k_1080p_4spp_d4 divergence_1080p_16spp_d50 \
    > "$OUT/pmc_sq.log" 2>&1 || exit 2
echo profile_ok
