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
