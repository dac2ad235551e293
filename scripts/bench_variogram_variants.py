"""Bin-pass times of the four k_vario_bin instantiations (haversine / Euclidean x semivariogram / covariogram) and of
the cross form at n soundings; Euclidean coordinates are the same points on a plane, max_dist scaled alike."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native
from sif_xco2_cokriging_amd.variogram import variogram_arrays

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(20005)
c0 = np.column_stack([rng.uniform(22, 58, n), rng.uniform(-125, -65, n)])
v0 = rng.standard_normal(n)
c1 = np.column_stack([rng.uniform(22, 58, n), rng.uniform(-125, -65, n)])
v1 = rng.standard_normal(n)
for metric, md in ((0, 1500.0), (1, 13.5)):          # 13.5 degrees ~ 1500 km
    for cov in (False, True):
        for cross in (False, True):
            h = native.Handle(0)
            h.set_metric(metric)
            t0 = time.perf_counter()
            if cross:
                out = variogram_arrays(h, c0, v0, c1, v1, False, md, 30, covariogram=cov)
            else:
                out = variogram_arrays(h, c0, v0, None, None, True, md, 30, covariogram=cov)
            dt = time.perf_counter() - t0
            st = h.vario_stats()
            print(json.dumps({"metric": "haversine" if metric == 0 else "euclid", "covariogram": cov, "cross": cross, "n": n,
                              "wall_s": round(dt, 4), "bin_pass_ms": round(h.timings()["vario_bin_ms"], 2),
                              "retained_pairs": int(out[3].sum()), "visited_pairs": st["bin_visited_pairs"],
                              "host_decided": [st["extent_host_pairs"], st["bin_host_pairs"]]}), flush=True)
            h.close()
