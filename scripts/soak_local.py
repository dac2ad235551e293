"""Randomised soak of ck_predict_local against the oracle: random data-set sizes, radii, metrics, cross-validation
mode, size-class boundaries and batch budgets.  Prints the worst deviation; exits non-zero on a mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth
from oracle import cokrige_oracle as orc

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ntrial = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = 0.0
for trial in range(ntrial):
    metric = int(rng.integers(0, 2))
    n0, n1 = int(rng.integers(1, 420)), int(rng.integers(1, 420))
    uni = rng.random() < 0.2
    if metric == 0:
        pb = synth.conus_problem(max(n0, n1), seed=int(rng.integers(1, 10 ** 6)))
        md = float(rng.choice([150.0, 400.0, 900.0, 2500.0, 1e9]))
    else:
        pb = synth.unit_square_problem(max(n0, n1), grid_side=7, seed=int(rng.integers(1, 10 ** 6)))
        md = float(rng.choice([0.05, 0.15, 0.4, 2.0]))
    pv = list(pb["params"])
    coords = [pb["coords"][0][:n0], pb["coords"][1][:n1]]
    values = [pb["values"][0][:n0], pb["values"][1][:n1]]
    h = native.Handle(0)
    h.set_option("site_order", int(rng.integers(0, 2)))
    if uni:
        coords, values = coords[:1], values[:1]
        h.set_model(1, pv[0:1], pv[2:3], pv[5:6], pv[8:9])
        op = orc.Params.from_flat([pv[0], pv[2], pv[5], pv[8]])
    else:
        h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        op = orc.Params.from_flat(pv)
    h.set_metric(metric)
    for k in range(len(coords)):
        h.set_data(k, coords[k], values[k])
    tile_min = int(rng.choice([0, 20, 64, 100, 10 ** 6]))
    h.set_option("local_tile_min", tile_min)
    h.set_option("local_group", int(rng.integers(1, 6)))
    if rng.random() < 0.5:
        h.set_option("local_slab_mb", int(rng.integers(1, 6)))
    i = 0 if uni else int(rng.integers(0, 2))
    cv = bool(rng.random() < 0.4)
    pc = coords[i][:: max(1, len(coords[i]) // 25)] if cv else pb["pcoords"][rng.permutation(len(pb["pcoords"]))[:25]]
    pred, err, info = h.predict_local(i, pc, max_dist=md, cv=cv)
    rp, re = orc.local_predict(op, coords, values, pc, i, metric, md, cv)[:2]
    h.close()
    ok = np.array_equal(np.isnan(pred), np.isnan(rp))
    fin = ~np.isnan(rp)
    dev = 0.0
    if fin.any():
        dev = max(float(np.max(np.abs(pred[fin] - rp[fin]) / np.maximum(1.0, np.abs(rp[fin])))),
                  float(np.max(np.abs(err[fin] ** 2 - re[fin] ** 2))))
    worst = max(worst, dev)
    tag = f"trial {trial}: metric {metric} n=({n0},{n1 if not uni else 0}) md {md} cv {cv} i {i} tile_min {tile_min} k_max {info['k_max']} empty {info['n_empty']} -> dev {dev:.2e}"
    print(tag, flush=True)
    if not ok or not (dev < 1e-8):
        print("MISMATCH", tag)
        sys.exit(1)
print("worst deviation", worst)
