"""assembly time of Sigma (K1) and of the right-hand-side rows c0^T | z^T (K2) at the headline size, several repetitions in
one process (HIP events inside the library); SURVEY 8(d): algorithmic bytes 8 [N (N + 1) / 2 + N m] over K1 + K2"""
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
ts, ta = [], []
for it in range(12):
    h.assemble_joint()
    ts.append(h.timings()["assemble_sigma_ms"])
    h.aux_begin(0, pb["pcoords"])
    ta.append(h.timings()["assemble_aux_ms"])
print("table errors, gate 2e-13:", [(h.table_info(b)["enabled"], f"{h.table_info(b)['max_rel_err']:.2e}") for b in range(3)])
N, m = 2 * n, len(pb["pcoords"])
t1, t2 = np.median(ts[2:]), np.median(ta[2:])
b1, b2 = 8 * N * (N + 1) / 2, 8 * N * m
print("assemble_sigma_ms", " ".join(f"{t:.3f}" for t in ts), "| median", f"{t1:.3f}", "ms ->", f"{b1 / t1 / 1e9:.2f} TB/s")
print("assemble_aux_ms  ", " ".join(f"{t:.3f}" for t in ta), "| median", f"{t2:.3f}", "ms ->", f"{b2 / t2 / 1e9:.2f} TB/s",
      "(device work of K2: round 4 one launch + the exact pass)")
print(f"K1 + K2: {(b1 + b2) / 1e9:.2f} GB in {t1 + t2:.3f} ms -> {(b1 + b2) / (t1 + t2) / 1e9:.2f} TB/s = {(b1 + b2) / (t1 + t2) / 1e9 / 8:.3f} of 8 TB/s")
