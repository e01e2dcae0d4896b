#!/bin/bash
# Every profile the benchmark's figures rest on, one after the other, on the GPU box (from the repo root):
#   tools/profile_all.sh <tag>      -> gpurun_out/<tag>_{fast,exact,fast_walked,exact_walked,scenes_fast,scenes_exact,primary}
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
echo all_profiles_ok
