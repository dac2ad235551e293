"""GPU parity of the pairwise lag-binning kernel (K5) against the fixtures produced by the
reference's MultiField.get_variogram (src/fields.py:208-232)."""
import numpy as np
import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def _mf(c0, v0, c1, v1):
    from sif_xco2_cokriging_amd import fields
    return fields.MultiField([fields.Field(c0, v0), fields.Field(c1, v1)])


@pytest.mark.parametrize("kind", ["Semivariogram", "Covariogram"])
@pytest.mark.parametrize("md,nb", [(1500, 30), (600, 12)])
def test_variogram_haversine(kind, md, nb):
    import warnings
    from sif_xco2_cokriging_amd import fields
    g = load_golden("variogram")
    mf = _mf(g["coords0"], g["values0"], g["coords1"], g["values1"])
    cfg = fields.VarioConfig(float(md), nb, kind=kind)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev = mf.empirical_variograms(cfg)
    for (i, j) in ((0, 0), (0, 1), (1, 1)):
        df = ev.df.loc[(i, j)]
        key = f"{kind[:4].lower()}_{md}_{nb}_{i}{j}"
        assert np.array_equal(df["bin_count"].values, g[key + "_counts"])          # counts exact
        np.testing.assert_allclose(df["bin_center"].values, g[key + "_centers"], rtol=1e-12)
        np.testing.assert_allclose(df["bin_mean"].values, g[key + "_means"], rtol=1e-11, atol=1e-14)


def test_variogram_euclid():
    import warnings
    from sif_xco2_cokriging_amd import fields
    g = load_golden("variogram")
    mf = _mf(g["e0"], g["w0"], g["e1"], g["w1"])
    cfg = fields.VarioConfig(0.6, 15, dist_units=None, fast_dist=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev = mf.empirical_variograms(cfg)
    for (i, j) in ((0, 0), (0, 1), (1, 1)):
        df = ev.df.loc[(i, j)]
        key = f"euc_{i}{j}"
        assert np.array_equal(df["bin_count"].values, g[key + "_counts"])
        np.testing.assert_allclose(df["bin_mean"].values, g[key + "_means"], rtol=1e-11)
        np.testing.assert_allclose(df["bin_center"].values, g[key + "_centers"], rtol=1e-12)


def test_variogram_pair_count_conservation():
    """size-independent property at a larger n: the bins partition the retained pairs, and the
    marginal variogram is invariant under a permutation of the sites."""
    from sif_xco2_cokriging_amd import native
    from sif_xco2_cokriging_amd.variogram import variogram_arrays
    rng = np.random.default_rng(5)
    n = 6000
    c = np.column_stack([rng.uniform(25, 50, n), rng.uniform(-120, -70, n)])
    v = rng.standard_normal(n)
    h = native.Handle(0)
    h.set_metric(0)
    c1, e1, m1, k1 = variogram_arrays(h, c, v, None, None, True, 1e9, 30)
    assert k1.sum() == n * (n - 1) // 2          # max_dist = inf keeps every pair
    perm = rng.permutation(n)
    c2, e2, m2, k2 = variogram_arrays(h, c[perm], v[perm], None, None, True, 1e9, 30)
    assert np.array_equal(k1, k2)
    np.testing.assert_allclose(m1, m2, rtol=1e-10)
    # cross-variogram of a field with itself counts every ordered pair incl. the n zero lags
    c3, e3, m3, k3 = variogram_arrays(h, c, v, c, v, False, 1e9, 30)
    assert k3.sum() == n * n


@pytest.mark.parametrize("metric,md", [(0, 700.0), (0, 2500.0), (1, 0.3)])
def test_variogram_tile_culling_and_point_order(metric, md):
    """From 2 048 points on the library lays the points out along a Hilbert curve and skips pair tiles whose
    bounding balls are farther apart than max_dist.  Against the oracle's dense computation, and with the
    sorting switched off (site_order = 0: the tiles are then not compact and hardly any is skipped): counts
    exact, means to rounding -- marginal and cross variograms, semivariogram and covariogram."""
    from sif_xco2_cokriging_amd import native
    from sif_xco2_cokriging_amd.variogram import variogram_arrays
    from oracle import cokrige_oracle as orc
    rng = np.random.default_rng(17)
    n0, n1 = 3300, 2600
    if metric == 0:
        c0 = np.column_stack([rng.uniform(25, 50, n0), rng.uniform(-120, -70, n0)])
        c1 = np.column_stack([rng.uniform(25, 50, n1), rng.uniform(-120, -70, n1)])
    else:
        c0, c1 = rng.random((n0, 2)), rng.random((n1, 2))
    v0, v1 = rng.standard_normal(n0), rng.standard_normal(n1)
    res = {}
    for order in (1, 0):
        h = native.Handle(0)
        h.set_option("site_order", order)
        h.set_metric(metric)
        res[order] = [variogram_arrays(h, c0, v0, None, None, True, md, 20),
                      variogram_arrays(h, c0, v0, c1, v1, False, md, 20),
                      variogram_arrays(h, c0, v0, c1, v1, False, md, 20, covariogram=True)]
    for a, b in zip(res[1], res[0]):
        assert np.array_equal(a[3], b[3])                         # counts
        np.testing.assert_allclose(a[1], b[1], rtol=1e-13)        # edges (from the extreme pairs)
        np.testing.assert_allclose(a[2], b[2], rtol=1e-10, atol=1e-13)
    # oracle: dense distances, same binning rule
    d00 = orc.distance_matrix(c0, c0, metric)
    iu = np.triu_indices(n0, 1)
    d = d00[iu]
    keep = d <= md
    cloud = 0.5 * (v0[iu[0]] - v0[iu[1]]) ** 2
    edges = res[1][0][1]
    ids = np.searchsorted(edges, d[keep], side="left")
    ids[d[keep] == edges[0]] = 1
    cnt = np.bincount(ids - 1, minlength=20)[:20]
    assert np.array_equal(cnt, res[1][0][3])
    means = np.bincount(ids - 1, weights=cloud[keep], minlength=20)[:20] / np.maximum(cnt, 1)
    np.testing.assert_allclose(res[1][0][2][cnt > 0], means[cnt > 0], rtol=1e-10)
