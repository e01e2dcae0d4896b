#!/bin/bash
# A/B timing of two builds of the library on one GPU box: tools/ab_bench.sh <other.so> [bench args]
# alternates bench.py runs with the in-tree library and with <other.so>, twice each.
set -o pipefail
OTHER=$1; shift
for round in 1 2; do
  for lib in "" "$OTHER"; do
    RAYTRACER_AMD_LIB=$lib timeout -k 10 200 python3 bench.py --no-aux --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('${lib:-in-tree}', 'fast ms', round(d['ms_per_step'],4), 'kernel', round(d['kernel_ms'],4), 'exact ms', round(d['exact']['ms_per_step'],4) if 'exact' in d else None, flush=True)" || exit 1
  done
done
