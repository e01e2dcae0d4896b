#!/bin/bash
# Every profile the benchmark's figures rest on, one after the other, on the GPU box (from the repo root):
#   tools/profile_all.sh <tag>      -> gpurun_out/<tag>_{fast,exact,fast_walked,exact_walked,scenes_fast,scenes_exact,primary}, <tag>_wave_timeline.txt
# and the folded counters in gpurun_out/profiles_out/r03_counters.json.  Stops at the first step that fails.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
rm -f profiles/r03_counters.json
for spec in "fast ${TAG}_fast" "exact ${TAG}_exact" "fast ${TAG}_fast_walked walked" "exact ${TAG}_exact_walked walked"; do
    echo "== profile_bench $spec"
    tools/profile_bench.sh $spec || { echo "profile_bench $spec failed: $?"; exit 1; }
done
for mode in fast exact; do
    echo "== profile_scenes $mode"
    tools/profile_scenes.sh $mode ${TAG}_scenes_$mode || { echo "profile_scenes $mode failed: $?"; exit 2; }
done
echo "== profile_primary"
tools/profile_primary.sh ${TAG}_primary || { echo "profile_primary failed: $?"; exit 3; }
# the headline launch wave by wave on a settled chip (the measurement build of the library: `make timeline` beforehand)
if [ -f tools/bin/libhip_raytracer_timeline.so ]; then
    echo "== wave timeline"
    RAYTRACER_AMD_LIB=$ROOT/tools/bin/libhip_raytracer_timeline.so TIMELINE_LAUNCHES=800 timeout -k 10 120 python3 tools/wave_timeline.py > gpurun_out/${TAG}_wave_timeline.txt 2>&1 || { echo "wave timeline failed: $?"; exit 4; }
fi
echo all_profiles_ok
