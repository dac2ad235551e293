"""BASELINE config 5: empirical (cross-)semivariogram on synthetic soundings, pairwise lag-binning
kernel on one MI355X.  VarioConfig(1500 km, 30 bins) as in the reference's notebooks
(research/variography_compare_tlag.ipynb:86).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native
from sif_xco2_cokriging_amd.variogram import variogram_arrays

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cross = "cross" in sys.argv[2:]
euclid = "euclid" in sys.argv[2:]      # the same soundings on a plane (km): the Euclidean instantiation of the kernels
rng = np.random.default_rng(20005)
c0 = np.column_stack([rng.uniform(22, 58, n), rng.uniform(-125, -65, n)])
v0 = rng.standard_normal(n)
c1 = np.column_stack([rng.uniform(22, 58, n), rng.uniform(-125, -65, n)])
v1 = rng.standard_normal(n)
if euclid:
    c0 = np.column_stack([c0[:, 0] * 111.2, c0[:, 1] * 85.0])
    c1 = np.column_stack([c1[:, 0] * 111.2, c1[:, 1] * 85.0])
h = native.Handle(0)
h.set_metric(1 if euclid else 0)
variogram_arrays(h, c0[:4096], v0[:4096], None, None, True, 1500.0, 30)      # code objects loaded before the clock starts
t0 = time.perf_counter()
if cross:
    centers, edges, means, counts = variogram_arrays(h, c0, v0, c1, v1, False, 1500.0, 30)
    pairs = n * n
else:
    centers, edges, means, counts = variogram_arrays(h, c0, v0, None, None, True, 1500.0, 30)
    pairs = n * (n - 1) // 2
dt = time.perf_counter() - t0
tb = h.timings()["vario_bin_ms"]
st = h.vario_stats()
# SURVEY 8(d) for K5: FP64 vector work.  Per VISITED pair: x = -(u_i . u_j) 5 flop (3 for the Euclidean q), the cloud value
# 2, one subtraction per compared level of the sub-chunk's window (3.5 on average at these settings) = 10.5 flop against
# the 78.6 TFLOP/s vector peak.  Most of the loop's vector instructions are not flops (sign-bit shifts, integer minima,
# popcount, one address): the instruction-issue fraction (16 vector instructions per 64-pair step, 4 cycles each, 1 024
# SIMDs at the measured clock) is the figure that says how far the kernel is from ITS bound; both are given.
flop_per_pair, valu_per_step, simds, clk = (10.5 if not euclid else 8.5), 16.0, 1024, 2.4e9
vp = st["bin_visited_pairs"]
roof = {"kernel": "k_vario_bin", "bound": "fp64 vector ALU (issue)", "visited_pairs": vp, "flop_per_visited_pair": flop_per_pair,
        "achieved_TFLOPs": flop_per_pair * vp / (tb / 1e3) / 1e12, "peak_TFLOPs": 78.6,
        "frac_of_fp64_vector_peak": flop_per_pair * vp / (tb / 1e3) / 1e12 / 78.6,
        "valu_instr_per_64_pairs": valu_per_step,
        "valu_issue_frac": (vp / 64.0) * valu_per_step * 4.0 / (simds * clk) / (tb / 1e3),
        "valu_issue_note": "16 vector instructions per 64-pair step from the SQ counters of profiles/r02_variogram_1M_pmc.txt "
                           "(the kernel's loop has not changed since), 4 cycles each on 1 024 SIMDs at 2.4 GHz"}
print(json.dumps({"workload": f"config 5: {'cross-' if cross else ''}semivariogram, {n} soundings, max_dist 1500 km, 30 bins"
                              + (", Euclidean on a plane" if euclid else ""), "roofline": roof,
                  "pairs": pairs, "retained_pairs": int(counts.sum()), "wall_s": dt, "bin_pass_ms": tb,
                  "pairs_per_s_wall": pairs / dt, "pairs_per_s_bin_pass": pairs / (tb / 1e3),
                  "visited_pairs": st["bin_visited_pairs"], "visited_pairs_per_s_bin_pass": st["bin_visited_pairs"] / (tb / 1e3),
                  "host_decided_pairs": [st["extent_host_pairs"], st["bin_host_pairs"]], "extent_extra_rounds": st["extent_extra_rounds"],
                  "bin_mean_first3": means[:3].tolist(), "bin_count_first3": counts[:3].tolist()}))
if "cpu" in sys.argv[2:]:
    # the reference's dense-matrix path (oracle restatement) on this box's host cores, bounded sample
    from oracle import cokrige_oracle as orc
    nc = 8000
    t0 = time.perf_counter()
    orc.variogram(c0[:nc], v0[:nc], c0[:nc], v0[:nc], True, 0, 1500.0, 30)
    dtc = time.perf_counter() - t0
    print(json.dumps({"cpu_baseline": "oracle variogram (dense n x n distance / cloud matrices, numpy)", "n": nc,
                      "pairs": nc * (nc - 1) // 2, "seconds": dtc, "pairs_per_s": nc * (nc - 1) / 2 / dtc,
                      "cores": os.cpu_count()}))
