"""A/B, interleaved in one process: the local predictor's tiled path right-looking (a K = 256 trailing update behind every group of
four 64-column blocks) against left-looking (one pass with K = g0 in front of every group, option local_left).

    python scripts/ab_local.py [n_obs=20000] [max_dist ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dists = [float(x) for x in (sys.argv[2:] or ["200", "400", "600"])]
pb = synth.conus_problem(n)
pv = pb["params"]
h = native.Handle(0)
h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
h.set_metric(0)
for k in range(2):
    h.set_data(k, pb["coords"][k], pb["values"][k])
h.local_reserve(0)
for md in dists:
    ref = None
    for rep in range(3):
        for name, left in (("right-looking", 0), ("left-looking", 1)):
            h.set_option("local_left", left)
            t0 = time.perf_counter()
            pred, err, info = h.predict_local(0, pb["pcoords"], md)
            wall = (time.perf_counter() - t0) * 1e3
            if ref is None:
                ref = (pred, err)
            same = np.array_equal(pred, ref[0], equal_nan=True) and np.array_equal(err, ref[1], equal_nan=True)
            print(f"max_dist {md:6.0f} km  k_max {info['k_max']:5d}  {name:14s} device {h.timings()['local_ms']:8.2f} ms  wall {wall:8.2f} ms  "
                  f"same bits as the first run: {same}", flush=True)
