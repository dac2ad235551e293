#!/bin/bash
# Everything profiles/ holds for one round, on the GPU box:  scripts/profile_round.sh <tag>
#   1. bench.py: rocprofv3 --kernel-trace --stats, then three separate --pmc passes (SQ, FETCH_SIZE, WRITE_SIZE)
#   2. variogram (1 M soundings): kernel stats + SQ counters of the pair kernels
#   3. local predictor (400 km): kernel stats + SQ counters of its matrix-core kernels
# Outputs under gpurun_out/prof_<tag>*/ ; copy the summaries into profiles/ (scripts/collect_profiles.py).
TAG=${1:-r04}
PART=${2:-all}      # part1: the rocprofv3 passes | part2: the bench lines and the A/B scripts (a gpurun call is capped at 20 minutes)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
if [ "$PART" != "part2" ]; then
scripts/profile_bench.sh $TAG > gpurun_out/prof_${TAG}_bench.txt 2>&1
echo "bench profile done"
scripts/prof_variogram.sh $TAG > gpurun_out/prof_${TAG}_vario_stats.txt 2>&1
scripts/pmc_kernel.sh ${TAG}_vario "k_vario" scripts/bench_variogram.py 1000000 > gpurun_out/prof_${TAG}_vario_pmc.txt 2>&1
echo "variogram profile done"
scripts/prof_local.sh $TAG 20000 400 400 > gpurun_out/prof_${TAG}_local_stats.txt 2>&1
scripts/pmc_kernel.sh ${TAG}_local "k_lt_|k_local" scripts/bench_local.py 20000 400 400 > gpurun_out/prof_${TAG}_local_pmc.txt 2>&1
echo "local profile done"
python3 scripts/time_assembly.py > gpurun_out/prof_${TAG}_assembly.txt 2>/dev/null
scripts/pmc_kernel.sh ${TAG}_asm "k_assemble" scripts/time_assembly.py >> gpurun_out/prof_${TAG}_assembly.txt 2>&1
echo "assembly profile done"
fi
if [ "$PART" = "part1" ]; then exit 0; fi
python3 bench.py --steps 10 --warmup 3 > gpurun_out/prof_${TAG}_bench_line.json 2> gpurun_out/prof_${TAG}_bench_line.err
python3 bench.py --config 1 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_${TAG}_bench_config1.json 2>/dev/null
python3 bench.py --n-obs 50000 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_${TAG}_bench_n50k.json 2>/dev/null
python3 scripts/ab_tall.py 20000 3 plain,two,tall,tallG2,tallG4,tallG4F2,tallF1T 2>/dev/null | grep "^N=" > gpurun_out/prof_${TAG}_tall_ab.txt
python3 scripts/ab_tall.py 5000 3 plain,two,tall,tallG2 1 2>/dev/null | grep "^N=" >> gpurun_out/prof_${TAG}_tall_ab.txt
python3 scripts/bench_variogram.py 1000000 > gpurun_out/prof_${TAG}_vario_1M.json 2>/dev/null
python3 scripts/bench_variogram.py 1000000 cross cpu >> gpurun_out/prof_${TAG}_vario_1M.json 2>/dev/null
python3 scripts/bench_local.py 20000 50 100 200 400 600 > gpurun_out/prof_${TAG}_local.json 2>/dev/null
python3 scripts/bench_local.py 20000 --reserve 50 100 200 400 600 > gpurun_out/prof_${TAG}_local_reserved.json 2>/dev/null
python3 scripts/bench_loocv.py > gpurun_out/prof_${TAG}_loocv.json 2>/dev/null
python3 scripts/bench_variogram.py 1000000 euclid >> gpurun_out/prof_${TAG}_vario_1M.json 2>/dev/null
echo "bench lines done"
# round 3: assembly kernels K1 / K2 (timings + SQ counters), the panel step's variants, the potrf phase profile, and the
# two-rank rehearsal of the multi-GPU form on this one GPU (gloo; calibration and schedule tuning included)
python3 scripts/ab_panel_step.py 20000 3 launches,coop,coopLA 2>/dev/null | grep "^N=" > gpurun_out/prof_${TAG}_panel_step.txt
python3 scripts/ab_panel_step.py 5000 3 launches,coop,coopLA 2>/dev/null | grep "^N=" >> gpurun_out/prof_${TAG}_panel_step.txt
python3 scripts/diag_potrf.py 2>/dev/null >> gpurun_out/prof_${TAG}_panel_step.txt
python3 -m pytest tests/test_gpu_distributed.py -q -s -k rccl 2>/dev/null | grep -a "RCCL" > gpurun_out/prof_${TAG}_rccl_single_rank.txt
CK_DIST_BACKEND=gloo CK_BENCH_CONFIG3=1 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_${TAG}_bench_2rank_gloo.json 2>/dev/null
echo "round-3 extras done"
