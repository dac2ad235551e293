"""A/B of the panel step inside one process: option panel_fused 0 (potrf64 / trsm64m / K = 64 updates, right-looking)
against its fused left-looking forms (bit 0: factorisation, bit 1: right-hand-side rows).
usage: ab_panel.py [n_obs] [values, default 0,2] [option name, default panel_fused]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
pb = synth.conus_problem(n)
pv = pb["params"]
res = {}
for rep in range(2):
    for fused in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,2").split(",")]:
        h = native.Handle(0)
        h.set_option(sys.argv[3] if len(sys.argv) > 3 else "panel_fused", fused)
        h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        h.set_metric(0)
        for k in range(2):
            h.set_data(k, pb["coords"][k], pb["values"][k])
        for it in range(2):
            h.assemble_joint()
            t0 = time.perf_counter()
            h.factor()
            pred, err = h.predict(0, pb["pcoords"])
            dt = time.perf_counter() - t0
        t = h.timings()
        res[fused] = (pred, err)
        print(json.dumps({"panel_fused": fused, "rep": rep, "factor_ms": t["factor_ms"], "solve_ms": t["solve_ms"],
                          "wall_ms": 1e3 * dt}), flush=True)
        del h
ks = sorted(res)
print("max |pred diff|", float(np.max(np.abs(res[ks[0]][0] - res[ks[-1]][0]))), "max |err diff|", float(np.max(np.abs(res[ks[0]][1] - res[ks[-1]][1]))))
