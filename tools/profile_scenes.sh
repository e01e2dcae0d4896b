#!/bin/bash
# rocprofv3 evidence for the shape-list scenes (tools/bench_scenes.py) on the GPU box: one kernel
# trace over all cases, then per case two counter passes (issue counters, executed FP32 operation
# counters), each a run of its own, folded into profiles/r03_counters.json.
# Usage (from the repo root on the GPU box): tools/profile_scenes.sh <mode> <outdir-under-gpurun_out>
set -o pipefail
MODE=${1:-fast}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${2:-prof_scenes_$MODE}
REPS=2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 "$ROOT/tools/bench_scenes.py" --mode "$MODE" --reps 4 \
    --case config2_1080p_64spp_d8 stress10k_4k_2spp_d1 stress10k_4k_4spp_d4 divergence_4k_16spp_d50 stress10k_1080p_2spp_d1 stress10k_1080p_4spp_d4 stress2k_1080p_16spp_d4 divergence_1080p_16spp_d50 showcase_1080p_64spp_d8 \
    > "$OUT/trace.log" 2>&1 || exit 1
for CASE in ${CASES:-config2_1080p_64spp_d8 stress10k_4k_2spp_d1 stress10k_4k_4spp_d4 divergence_4k_16spp_d50}; do
    mkdir -p "$OUT/$CASE"
    timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY -d "$OUT/$CASE/pmc_sq" -o pmc -- python3 "$ROOT/tools/bench_scenes.py" --mode "$MODE" --reps $REPS --case $CASE \
        > "$OUT/$CASE/pmc_sq.log" 2>&1 || exit 2
    timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE -d "$OUT/$CASE/pmc_flops" -o pmc -- python3 "$ROOT/tools/bench_scenes.py" --mode "$MODE" --reps $REPS --case $CASE \
        > "$OUT/$CASE/pmc_flops.log" 2>&1 || exit 3
    (cd "$ROOT" && python3 tools/collect_counters.py ${CASE}_$MODE $REPS "$OUT/$CASE" && python3 tools/summarize_profile.py "$OUT/$CASE" "$OUT/$CASE.txt" > /dev/null) || exit 4
    [ -n "$KEEP_DB" ] || rm -rf "$OUT/$CASE"
done
(cd "$ROOT" && python3 tools/summarize_profile.py "$OUT" "$OUT/trace.txt" > /dev/null) || exit 5
[ -n "$KEEP_DB" ] || rm -rf "$OUT/trace"
mkdir -p "$ROOT/gpurun_out/profiles_out" && cp "$ROOT/profiles/r03_counters.json" "$ROOT/gpurun_out/profiles_out/r03_counters.json"
echo profile_ok
