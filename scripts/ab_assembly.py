"""A/B of the table-path assembly kernels: one 64 x 512 strip per workgroup against a resident set of workgroups on a work
queue (option assemble_queue = number of workgroups).  K1 / K2 times (HIP events inside the library) and bit-identity of the results.

    python scripts/ab_assembly.py [n_obs=20000] [config=2|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pb = synth.conus_problem(n) if cfg == 2 else synth.unit_square_problem(n, grid_side=100)
pv = pb["params"]
N, m = 2 * n, len(pb["pcoords"])
b1, b2 = 8 * N * (N + 1) / 2, 8 * N * m
ref = None
for q in (0, -1, 768, 1024, 0, -1):
    h = native.Handle(0)
    h.set_option("assemble_queue", q)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(pb["metric"])
    for k in range(2):
        h.set_data(k, pb["coords"][k], pb["values"][k])
    ts, ta = [], []
    for it in range(10):
        h.assemble_joint()
        ts.append(h.timings()["assemble_sigma_ms"])
        h.aux_begin(0, pb["pcoords"])
        ta.append(h.timings()["assemble_aux_ms"])
    t1, t2 = np.median(ts[2:]), np.median(ta[2:])
    h.assemble_joint()
    info, pred, err = h.factor_predict(0, pb["pcoords"])
    if ref is None:
        ref = (pred, err)
    same = np.array_equal(pred, ref[0]) and np.array_equal(err, ref[1])
    print(f"N={N} m={m} queue {q:5d}: K1 {t1:.3f} ms = {b1 / t1 / 1e9 / 8:.3f}  K2 {t2:.3f} ms = {b2 / t2 / 1e9 / 8:.3f}  K1+K2 {(b1 + b2) / (t1 + t2) / 1e9 / 8:.3f} of 8 TB/s"
          f" | fallbacks {h.table_fallbacks()} | predictions same bits: {same}", flush=True)
    h.close()
