#!/bin/bash
# per-kernel times of the 1 M-sounding variogram bench (rocprofv3 kernel trace); usage: scripts/prof_variogram.sh <tag> [cross]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_vario_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o vg -- python3 $ROOT/scripts/bench_variogram.py 1000000 "$@" > $OUT/run.log 2>&1
cat $OUT/vg_kernel_stats.csv
