#!/bin/bash
# per-kernel times of the local predictor bench (rocprofv3 kernel trace); usage: scripts/prof_local.sh <tag> <bench_local args...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_local_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lt -- python3 $ROOT/scripts/bench_local.py "$@" > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:12]:
    print(row["Name"][:48], row["Calls"], round(float(row["TotalDurationNs"]) / 1e6, 2), "ms")
PY
