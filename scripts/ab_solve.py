"""A/B, interleaved in one process: ck_predict on a resident factor (the second field of a Predictor, a new grid) with the
sequential grouped sweep (solve_la = 0) and with the chain of the next group under the bulk of the current one (solve_la = 1).

    python scripts/ab_solve.py [n_obs=20000] [reps=4]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pb = synth.conus_problem(n, seed=20003)
pv, pc = pb["params"], pb["pcoords"]
h = native.Handle(0)
h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
h.set_metric(pb["metric"])
for k in range(2):
    h.set_data(k, pb["coords"][k], pb["values"][k])
h.assemble_joint()
assert h.factor() == 0
ref = None
for r in range(reps):
    for la in (0, 1):
        h.set_option("solve_la", la)
        h.synchronize()
        t0 = time.perf_counter()
        pred, err = h.predict(r % 2, pc)
        wall = (time.perf_counter() - t0) * 1e3
        t = h.timings()
        key = r % 2
        if ref is None:
            ref = {}
        if key not in ref:
            ref[key] = (pred, err)
        same = np.array_equal(pred, ref[key][0]) and np.array_equal(err, ref[key][1])
        print(f"N={2 * n} m={len(pc)} field {key} solve_la={la}: wall {wall:7.2f} ms, sweep {t['solve_ms']:7.2f} ms, K2 {t['assemble_aux_ms']:.2f} ms | same bits {same}", flush=True)
