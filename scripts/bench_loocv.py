"""Leave-one-out cross-validation of the joint predictor from ONE factorisation (ck_loocv) at the headline size."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
pb = synth.conus_problem(n)
pv = pb["params"]
h = native.Handle(0)
h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
h.set_metric(0)
for k in range(2):
    h.set_data(k, pb["coords"][k], pb["values"][k])
h.assemble_joint()
h.factor()
for rep in range(2):
    for i in (0, 1):
        t0 = time.perf_counter()
        pred, err = h.loocv(i, n)
        dt = time.perf_counter() - t0
        t = h.timings()
        print(json.dumps({"n_obs": n, "process": i, "rep": rep, "seconds": dt, "solve_ms": t["solve_ms"],
                          "loo_predictions_per_s": n / dt, "rmse": float(np.sqrt(np.mean((pred - pb["values"][i]) ** 2))),
                          "checksum": float(pred.sum())}), flush=True)
