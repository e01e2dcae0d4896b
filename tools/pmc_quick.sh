#!/bin/bash
# Quick counter pass over bench.py's headline launch for a library variant:
#   tools/pmc_quick.sh <lib.so> <mode> <tag>   -> gpurun_out/<tag>.txt (instruction-class counters and launch time)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
LIB=$1; MODE=${2:-fast}; TAG=${3:-pmcq}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RAYTRACER_AMD_LIB=$LIB
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES -d "$OUT/pmc" -o pmc --output-format csv -- python3 $ROOT/bench.py --mode $MODE --steps 10 --warmup 3 --no-cpu-baseline --no-aux --no-second-mode > "$OUT/log.txt" 2>&1 || exit 2
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if "camera" not in k and "primary" not in k and "path" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
rm -rf "$OUT/pmc"
