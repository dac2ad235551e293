"""assembly time of Sigma at the headline size, several repetitions in one process (HIP events inside the library)"""
import os, sys
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
ts = []
for it in range(12):
    h.assemble_joint()
    ts.append(h.timings()["assemble_sigma_ms"])
N = 2 * n
print("assemble_sigma_ms", " ".join(f"{t:.3f}" for t in ts), "| median", f"{np.median(ts[2:]):.3f}", "ms ->", f"{8 * N * (N + 1) / 2 / np.median(ts[2:]) / 1e9:.2f} TB/s")
