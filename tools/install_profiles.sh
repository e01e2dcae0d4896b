#!/bin/bash
# Copies what tools/profile_all.sh <tag> left under gpurun_out/ into profiles/ under the names the committed summaries carry:
#   tools/install_profiles.sh <tag in gpurun_out> <name under profiles/>      e.g.  tools/install_profiles.sh r03b r03_b
set -e
TAG=$1; NAME=$2
cp gpurun_out/profiles_out/r03_counters.json profiles/r03_counters.json
for m in fast exact fast_walked exact_walked; do cp gpurun_out/${TAG}_$m/summary.txt profiles/${NAME}_bench_$m.txt; done
for m in fast exact; do
    cat gpurun_out/${TAG}_scenes_$m/trace.txt gpurun_out/${TAG}_scenes_$m/config2_1080p_64spp_d8.txt gpurun_out/${TAG}_scenes_$m/stress10k_4k_2spp_d1.txt \
        gpurun_out/${TAG}_scenes_$m/stress10k_4k_4spp_d4.txt gpurun_out/${TAG}_scenes_$m/divergence_4k_16spp_d50.txt > profiles/${NAME}_scenes_$m.txt
done
cp gpurun_out/${TAG}_primary/summary.txt profiles/${NAME}_primary.txt
[ -f gpurun_out/${TAG}_wave_timeline.txt ] && grep -v amdgpu.ids gpurun_out/${TAG}_wave_timeline.txt > profiles/${NAME}_wave_timeline.txt
[ -f gpurun_out/${TAG}_bench_line.json ] && cp gpurun_out/${TAG}_bench_line.json profiles/${NAME}_bench_line.json
[ -f gpurun_out/${TAG}_bench_line_20_5.json ] && cp gpurun_out/${TAG}_bench_line_20_5.json profiles/${NAME}_bench_line_steps20_warmup5.json
[ -f gpurun_out/${TAG}_bench_line_collective.json ] && cp gpurun_out/${TAG}_bench_line_collective.json profiles/${NAME}_bench_line_forced_collective.json
python3 tools/resource_usage.py profiles/${NAME}_resource_usage.txt > /dev/null
echo installed
