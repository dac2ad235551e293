"""Copy the summaries of scripts/profile_round.sh from gpurun_out/ into profiles/ (tracked) and derive
profiles/<tag>_traffic.json: HBM-side bytes per launch of the dominant kernel from the FETCH_SIZE / WRITE_SIZE passes,
corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes, next to the algorithmic bytes of the same launches.
usage: python scripts/collect_profiles.py [tag]"""
import glob
import json
import os
import shutil
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, dst))
        print("copied", dst)
    else:
        print("MISSING", src)


cp(f"{G}/prof_{tag}_bench.txt", f"{tag}_bench_n20k_rocprof_summary.txt")
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    cp(f"{G}/prof_{tag}/{sub}_per_kernel_mean.csv", f"{tag}_{sub}_per_kernel_mean.csv")
# (gpurun merges every call's output into the same directories: take the NEWEST file, not the first)
newest = lambda files: max(files, key=os.path.getmtime)
st = glob.glob(f"{G}/prof_{tag}/trace/**/*kernel_stats.csv", recursive=True)
if st:
    cp(newest(st), f"{tag}_bench_n20k_kernel_stats.csv")
so = glob.glob(f"{G}/prof_{tag}/trace_sequential/**/*kernel_stats.csv", recursive=True)
if so:
    cp(newest(so), f"{tag}_trace_sequential_kernel_stats.csv")   # --sweeps sequential: round 3's two kernels, each with the chip to itself
cp(f"{G}/prof_{tag}_bench_line.json", f"{tag}_bench_n20k.json")
cp(f"{G}/prof_{tag}_bench_config1.json", f"{tag}_bench_config1_n5k.json")
cp(f"{G}/prof_{tag}_bench_n50k.json", f"{tag}_bench_n50k_1gpu.json")
cp(f"{G}/prof_{tag}_vario_1M.json", f"{tag}_variogram_1M.json")
cp(f"{G}/prof_{tag}_vario_pmc.txt", f"{tag}_variogram_1M_pmc.txt")
cp(f"{G}/prof_{tag}_local.json", f"{tag}_local_predictor.json")
cp(f"{G}/prof_{tag}_local_pmc.txt", f"{tag}_local_400km_pmc.txt")
cp(f"{G}/prof_{tag}_loocv.json", f"{tag}_loocv.json")
cp(f"{G}/prof_{tag}_assembly.txt", f"{tag}_assembly_k1_k2.txt")
cp(f"{G}/prof_{tag}_panel_step.txt", f"{tag}_panel_step.txt")
cp(f"{G}/prof_{tag}_tall_ab.txt", f"{tag}_tall_sweep_ab.txt")
cp(f"{G}/prof_{tag}_local_reserved.json", f"{tag}_local_predictor_reserved.json")
cp(f"{G}/prof_{tag}_rccl_single_rank.txt", f"{tag}_rccl_single_rank.txt")
cp(f"{G}/prof_{tag}_bench_2rank_gloo.json", f"{tag}_bench_2rank_gloo_one_gpu.json")
for name, out in ((f"prof_vario_{tag}", f"{tag}_variogram_1M_kernel_stats.csv"), (f"prof_local_{tag}", f"{tag}_local_400km_kernel_stats.csv")):
    f = glob.glob(f"{G}/{name}/**/*kernel_stats.csv", recursive=True)
    if f:
        cp(newest(f), out)

# ---- traffic of k_tall_group_d -------------------------------------------------------------------
fe, wr = f"{P}/{tag}_pmc_fetch_per_kernel_mean.csv", f"{P}/{tag}_pmc_write_per_kernel_mean.csv"
if os.path.exists(fe) and os.path.exists(wr):
    f = pd.read_csv(fe).set_index("Kernel_Name")
    w = pd.read_csv(wr).set_index("Kernel_Name")
    k = [x for x in f.index if x.startswith("k_tall_group_d")][0]
    fetch_kb, write_kb = float(f.loc[k, "FETCH_SIZE"]), float(w.loc[k, "WRITE_SIZE"])
    # algorithmic bytes of the launches of one pass at N = 40 000, m = 8 833 (Npad = 40 448, 79 panels, a first group of 2 panels then groups of 4, mpad = 8 960;
    # the schedule of ck_api.hip: tall_sweeps -- per group two in-group launches on one block column, then A / B1 / B2):
    # every C tile of a launch (triangle + right-hand-side block) read once and written once, every operand row (panel rows of
    # the target columns' row range + the right-hand-side rows) read once per source panel
    NB, N, mpad, G = 512, 40000, 8960, 4
    nK = -(-N // NB)
    Np = nK * NB
    starts = [0] + list(range(G // 2, nK, G)) + [nK]      # group_plan(): a first group of G / 2 panels, then groups of G
    ng = len(starts) - 1

    def col_c(J):          # doubles of C in block column J: lower 128-tiles + right-hand-side block
        return (Np - J * NB) * NB - NB * (NB - 128) // 2 + mpad * NB

    def launch(K0, npan, J0, nJ):
        c = sum(col_c(J) for J in range(J0, J0 + nJ))
        rows = (Np - J0 * NB) + mpad          # operand rows read per source panel (A rows of the range + right-hand-side rows)
        return 8 * (2 * c + npan * rows * NB)
    launches = []
    for g in range(ng):
        K0, Gc = starts[g], starts[g + 1] - starts[g]
        for q in range(1, Gc):
            launches.append(launch(K0, q, K0 + q, 1))
        first = lambda x: starts[x]
        count = lambda x: starts[x + 1] - starts[x]
        if g + 1 < ng:
            launches.append(launch(K0, Gc, first(g + 1), count(g + 1)))
        if g + 2 < ng:
            launches.append(launch(K0, Gc, first(g + 2), count(g + 2)))
        if g + 3 < ng:
            launches.append(launch(K0, Gc, first(g + 3), nK - first(g + 3)))
    alg = sum(launches) / len(launches)
    out = {
        "kernel": k, "launches_per_pass": len(launches),
        "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb,
        "fetch_correction": "x2: on gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streams at 64 bytes "
                            "(MI355X_MICROARCH.md, HBM section); the operand panels arrive through buffer_load_dwordx4 ... lds "
                            "(16 B per lane).  The C tile is read with 8-byte-per-lane loads, a width the guide calls uncalibrated: "
                            "doubling everything is the upper bound, the raw figure the lower.",
        "k_tall_group_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
        "k_tall_group_bytes_per_launch_uncorrected": fetch_kb * 1024 + write_kb * 1024,
        "algorithmic_bytes_per_launch": alg,
        "ratio_upper": (2 * fetch_kb * 1024 + write_kb * 1024) / alg,
        "ratio_lower": (fetch_kb * 1024 + write_kb * 1024) / alg,
        "note": f"mean over the {3 * len(launches)} launches of a PMC pass (the default command with steps 1, warm-up 1 and its cold PCIe-inclusive pass "
                f"= three passes of {len(launches)} launches each, dispatches serialised by the profiler); algorithmic = every C tile read and written once + every operand row read once per source panel",
    }
    json.dump(out, open(f"{P}/{tag}_traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))
