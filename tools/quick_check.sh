#!/bin/bash
# Parity tests of the kernels, then one short bench line: tools/quick_check.sh <tag>
set -o pipefail
TAG=${1:-qc}
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/${TAG}_parity.log 2>&1
rc=$?
tail -4 gpurun_out/${TAG}_parity.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 bench.py --no-aux --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1 || { tail -5 gpurun_out/${TAG}_bench.log; exit 1; }
tail -1 gpurun_out/${TAG}_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fast ms', round(d['ms_per_step'],4), 'kernel', d.get('kernel_ms'), 'exact', round(d['exact']['ms_per_step'],4))"
