"""PCIe-inclusive rate of the headline workload: everything from host arrays to host results in one timed
region -- new handle, model, site upload (with the Hilbert sort on the host), assembly, factorisation, the
prediction sweep, results back -- against bench.py's resident-input figure."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
pb = synth.conus_problem(n)
pv = pb["params"]

def run():
    h = native.Handle(0)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(0)
    for k in range(2):
        h.set_data(k, pb["coords"][k], pb["values"][k])
    h.assemble_joint()
    h.factor()
    out = h.predict(0, pb["pcoords"])
    t = h.timings()
    h.close() if hasattr(h, "close") else None
    return out, t

run()   # library load, first-touch of the device
rows = []
for rep in range(3):
    t0 = time.perf_counter()
    out, t = run()
    dt = time.perf_counter() - t0
    dev = t["assemble_sigma_ms"] + t["factor_ms"] + t["assemble_aux_ms"] + t["solve_ms"] + t["reduce_ms"]
    rows.append({"rep": rep, "host_to_host_ms": 1e3 * dt, "device_kernels_ms": dev,
                 "grid_points_per_s_host_to_host": len(out[0]) / dt})
    print(json.dumps(rows[-1]), flush=True)
