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
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
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
st = glob.glob(f"{G}/prof_{tag}/trace/**/*kernel_stats.csv", recursive=True)
if st:
    cp(st[0], f"{tag}_bench_n20k_kernel_stats.csv")
so = glob.glob(f"{G}/prof_{tag}/trace_overlapped/**/*kernel_stats.csv", recursive=True)
if so:
    cp(so[0], f"{tag}_trace_overlapped_kernel_stats.csv")   # the default (overlapped) form of the same command
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
cp(f"{G}/prof_{tag}_bench_2rank_gloo.json", f"{tag}_bench_2rank_gloo_one_gpu.json")
for name, out in ((f"prof_vario_{tag}", f"{tag}_variogram_1M_kernel_stats.csv"), (f"prof_local_{tag}", f"{tag}_local_400km_kernel_stats.csv")):
    f = glob.glob(f"{G}/{name}/**/*kernel_stats.csv", recursive=True)
    if f:
        cp(f[0], out)

# ---- traffic of k_syrk_group_d -------------------------------------------------------------------
fe, wr = f"{P}/{tag}_pmc_fetch_per_kernel_mean.csv", f"{P}/{tag}_pmc_write_per_kernel_mean.csv"
if os.path.exists(fe) and os.path.exists(wr):
    f = pd.read_csv(fe).set_index("Kernel_Name")
    w = pd.read_csv(wr).set_index("Kernel_Name")
    k = [x for x in f.index if x.startswith("k_syrk_group_d")][0]
    fetch_kb, write_kb = float(f.loc[k, "FETCH_SIZE"]), float(w.loc[k, "WRITE_SIZE"])
    # algorithmic bytes of the 78 launches of one factorisation at N = 40 000 (Npad = 40 448, 79 panels, groups of 3):
    # every C tile of a launch read once and written once, every operand panel row read once
    NB, N = 512, 40000
    nK = -(-N // NB)
    Np = nK * NB
    launches = []
    for K0 in range(0, nK, 3):   # (79 panels: groups of three, no look-ahead -- the automatic schedule from 64 panels on)
        Gc = min(3, nK - K0)
        for g in range(1, Gc):
            rows = Np - (K0 + g) * NB
            c = rows * NB - NB * (NB - 128) // 2          # lower tiles of one block column (128-tiles on the diagonal kept whole)
            launches.append(8 * (2 * c + g * rows * NB))
        if K0 + Gc < nK:
            c = sum((Np - J * NB) * NB - NB * (NB - 128) // 2 for J in range(K0 + Gc, nK))
            rows = Np - (K0 + Gc) * NB
            launches.append(8 * (2 * c + Gc * rows * NB))
    alg = sum(launches) / len(launches)
    out = {
        "kernel": k, "launches_per_factorisation": len(launches),
        "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb,
        "fetch_correction": "x2: on gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streams at 64 bytes "
                            "(MI355X_MICROARCH.md, HBM section); the operand panels arrive through global_load_lds_dwordx4 "
                            "(16 B per lane).  The C tile is read with 8-byte-per-lane loads, a width the guide calls uncalibrated: "
                            "doubling everything is the upper bound, the raw figure the lower.",
        "fetch_bytes_corrected": 2 * fetch_kb * 1024, "write_bytes": write_kb * 1024,
        "k_syrk_group_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
        "k_syrk_group_bytes_per_launch_uncorrected": fetch_kb * 1024 + write_kb * 1024,
        "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": (2 * fetch_kb * 1024 + write_kb * 1024) / alg,
        "note": "mean per dispatch over the launches of k_syrk_group_d in the profiled bench run (warm-up, timed and cold pass; "
                "N = 40 000, panel_group = 3); FETCH_SIZE and WRITE_SIZE from separate rocprofv3 --pmc passes "
                "(scripts/profile_bench.sh), KB -> bytes.  Algorithmic bytes: each C tile of a launch read and written once, "
                "each operand panel row read once, averaged over the same launches.",
    }
    json.dump(out, open(f"{P}/{tag}_traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))
