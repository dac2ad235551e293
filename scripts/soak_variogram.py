"""Randomised soak of the lag-binning kernels against the oracle's dense computation: random sizes (below and
above the 2 048-point sorting threshold), radii, bin counts, metric, marginal / cross, semivariogram / covariogram,
clustered points (compact tiles, many culled) and duplicated sites (zero lags)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native
from sif_xco2_cokriging_amd.variogram import variogram_arrays
from oracle import cokrige_oracle as orc

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ntrial = int(sys.argv[2]) if len(sys.argv) > 2 else 20
worst = 0.0
for trial in range(ntrial):
    metric = int(rng.integers(0, 2))
    ni, nj = int(rng.integers(300, 5200)), int(rng.integers(300, 5200))
    def pts(n):
        if rng.random() < 0.4:   # clusters
            c = rng.uniform(0.1, 0.9, (6, 2))
            p = c[rng.integers(0, 6, n)] + 0.03 * rng.standard_normal((n, 2))
        else:
            p = rng.random((n, 2))
        if rng.random() < 0.3:
            p[: n // 10] = p[n // 2: n // 2 + n // 10]   # duplicated sites
        if metric == 0:
            p = np.column_stack([25 + 25 * p[:, 0], -120 + 50 * p[:, 1]])
        return p
    ci, cj = pts(ni), pts(nj)
    vi, vj = rng.standard_normal(ni), rng.standard_normal(nj)
    same = bool(rng.random() < 0.5)
    cov = bool(rng.random() < 0.3)
    md = float(rng.choice([200.0, 600.0, 1500.0, 6000.0])) if metric == 0 else float(rng.choice([0.04, 0.15, 0.5, 3.0]))
    nb = int(rng.choice([5, 12, 30, 36]))
    h = native.Handle(0)
    h.set_option("site_order", int(rng.integers(0, 2)))
    h.set_metric(metric)
    try:
        got = variogram_arrays(h, ci, vi, None if same else cj, None if same else vj, same, md, nb, covariogram=cov)
    except ValueError as e:
        print(f"trial {trial}: skipped ({e})")
        continue
    ref = orc.variogram(ci, vi, ci if same else cj, vi if same else vj, same, metric, md, nb, cov)
    ok = np.array_equal(got[3], ref[3])
    has = ref[3] > 0
    dev = float(np.max(np.abs(got[2][has] - ref[2][has]) / np.maximum(1e-3, np.abs(ref[2][has])))) if has.any() else 0.0
    dev = max(dev, float(np.max(np.abs(got[1] - ref[1]) / np.maximum(1e-12, np.abs(ref[1]).max()))))
    worst = max(worst, dev)
    tag = f"trial {trial}: metric {metric} n=({ni},{nj}) same {same} cov {cov} md {md} nb {nb} pairs {int(ref[3].sum())} -> dev {dev:.2e}"
    print(tag, flush=True)
    if not ok or not (dev < 1e-9):
        print("MISMATCH", tag, got[3], ref[3])
        sys.exit(1)
print("worst deviation", worst)
