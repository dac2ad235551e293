"""Randomised soak of the joint predictor (assemble -> factor -> predict, LOOCV) against the oracle: random sizes
around the panel / tile boundaries, both metrics, both parameter sets with random perturbations, univariate
models, random options (panel_group, panel_fused, site_order, exact_cov)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth
from oracle import cokrige_oracle as orc

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ntrial = int(sys.argv[2]) if len(sys.argv) > 2 else 20
worst = 0.0
for trial in range(ntrial):
    metric = int(rng.integers(0, 2))
    n0 = int(rng.choice([1, 17, 63, 64, 65, 255, 256, 257, 511, 512, 513, 700, 1023, 1025, 1400, 2300]))
    n1 = int(rng.choice([1, 30, 64, 200, 448, 449, 512, 600, 1100]))
    uni = rng.random() < 0.2
    if metric == 0:
        pb = synth.conus_problem(max(n0, n1), seed=int(rng.integers(1, 10 ** 6)))
    else:
        pb = synth.unit_square_problem(max(n0, n1), grid_side=12, seed=int(rng.integers(1, 10 ** 6)))
    pv = np.array(pb["params"], dtype=float)
    pv[2:5] = np.clip(pv[2:5] * rng.uniform(0.8, 1.3, 3), 0.25, 3.4)      # nu
    pv[5:8] *= rng.uniform(0.7, 1.4)                                        # length scales (kept equal: valid model)
    pv[8:10] = rng.uniform(0.005, 0.05, 2)                                  # nuggets
    pv[10] *= rng.uniform(0.2, 1.0)
    coords = [pb["coords"][0][:n0], pb["coords"][1][:n1]]
    values = [pb["values"][0][:n0], pb["values"][1][:n1]]
    m = int(rng.choice([1, 5, 127, 128, 140, 255, 256, 300, 700]))
    pc = pb["pcoords"][rng.permutation(len(pb["pcoords"]))[:m]]
    h = native.Handle(0)
    opts = {"site_order": int(rng.integers(0, 2)), "panel_group": int(rng.integers(0, 5)),
            "panel_fused": int(rng.choice([0, 1, 2, 3, 16, 18, 18, 18, 18])), "exact_cov": int(rng.random() < 0.2)}
    for k, v in opts.items():
        h.set_option(k, v)
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
    h.assemble_joint()
    i = 0 if uni else int(rng.integers(0, 2))
    fused = rng.random() < 0.6    # ck_factor_predict: the two sweeps overlapped, random stream assignment / grouping / look-ahead
    if fused:
        # (round 4: with the cooperative panel step -- panel_fused bit 4 -- the call is ONE sweep over the tall matrix unless tall_sweep = 0;
        # group plans with a short first group / small groups at the tail)
        fo = {"fused_prio": int(rng.integers(0, 3)), "fused_group": int(rng.integers(0, 5)), "fused_la": int(rng.integers(-1, 2)),
              "tall_sweep": int(rng.random() < 0.75), "group_first": int(rng.integers(0, 3)), "group_tail": int(rng.integers(0, 3)),
              "group_tail_panels": int(rng.integers(0, 6)), "tall_split": int(rng.integers(0, 3)),
              "tall_split_rows": int(rng.choice([0, 1024, 12288])), "tall_thin": int(rng.random() < 0.8)}
        for k, v in fo.items():
            h.set_option(k, v)
        opts.update(fo)
        info, pred, err = h.factor_predict(i, pc)
    else:
        info = h.factor()
    try:
        rp, re = orc.joint_predict(op, coords, values, pc, i, metric)
    except np.linalg.LinAlgError:
        print(f"trial {trial}: oracle not positive definite, info {info}")
        assert info != 0
        h.close()
        continue
    assert info == 0, info
    if not fused:
        pred, err = h.predict(i, pc)
    dev = max(float(np.max(np.abs(pred - rp)) / max(1.0, float(np.max(np.abs(rp))))), float(np.max(np.abs(err ** 2 - re ** 2))))
    ni = len(coords[i])
    if ni >= 3:
        cp, ce = h.loocv(i, ni)
        for ix in {0, ni // 2, ni - 1}:
            o1, o2 = orc.joint_predict(op, coords, values, coords[i][ix], i, metric, cv_ix=ix)
            dev = max(dev, abs(cp[ix] - o1[0]) / max(1.0, abs(o1[0])), abs(ce[ix] ** 2 - o2[0] ** 2))
    h.close()
    worst = max(worst, dev)
    tag = f"trial {trial}: metric {metric} n=({n0},{0 if uni else n1}) m {m} i {i} {opts} -> dev {dev:.2e}"
    print(tag, flush=True)
    if not (dev < 1e-7):
        print("MISMATCH", tag)
        sys.exit(1)
print("worst deviation", worst)
