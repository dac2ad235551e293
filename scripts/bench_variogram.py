"""BASELINE config 5: empirical (cross-)semivariogram on synthetic soundings, pairwise lag-binning
kernel on one MI355X.  VarioConfig(1500 km, 30 bins) as in the reference's notebooks
(research/variography_compare_tlag.ipynb:86).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native
from sif_xco2_cokriging_amd.variogram import variogram_arrays

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cross = len(sys.argv) > 2 and sys.argv[2] == "cross"
rng = np.random.default_rng(20005)
c0 = np.column_stack([rng.uniform(22, 58, n), rng.uniform(-125, -65, n)])
v0 = rng.standard_normal(n)
c1 = np.column_stack([rng.uniform(22, 58, n), rng.uniform(-125, -65, n)])
v1 = rng.standard_normal(n)
h = native.Handle(0)
h.set_metric(0)
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
print(json.dumps({"workload": f"config 5: {'cross-' if cross else ''}semivariogram, {n} soundings, max_dist 1500 km, 30 bins",
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
