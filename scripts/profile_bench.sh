#!/bin/bash
# Profile the headline bench on the GPU box: kernel-trace stats + three PMC passes
# (SQ counters, FETCH_SIZE, WRITE_SIZE kept in separate passes as the microarch guide prescribes).
# usage: scripts/profile_bench.sh <tag> [bench args...]     (run from the repo root via gpurun)
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# Round 4: the product path's dominant kernel is k_tall_group_d (every update of the tall matrix [Sigma; c0^T; z^T]), which only
# the DEFAULT form of the step runs -- so the stats pass and the three PMC passes profile the default command (rocprofv3
# serialises the dispatches of a counter pass, so a launch's counters are its own even though its launches overlap each other
# in the product); one more trace of --sweeps sequential shows round 3's two kernels with the chip to themselves.
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-config3 $@"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config3 $@ > $OUT/trace.log 2>&1 || echo "trace pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sequential -- python3 $ROOT/bench.py $ARGS --sweeps sequential > $OUT/trace_sequential.log 2>&1 || echo "sequential trace pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || echo "pmc sq pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || echo "pmc fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1 || echo "pmc write pass failed"
python3 $ROOT/scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
