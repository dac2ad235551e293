"""BASELINE.json's two largest configurations at FULL size on one MI355X, through size-independent properties
(no oracle can run there): configs[3] n_obs = 50 000 per process (N = 100 000, Sigma = 40 GB) and configs[4]
1 000 000 soundings (5e11 pairs)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_joint_n100k_properties():
    from sif_xco2_cokriging_amd import native, synth
    pb = synth.conus_problem(50000, seed=20004)
    pv = pb["params"]
    h = native.Handle(0)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(pb["metric"])
    for k in range(2):
        h.set_data(k, pb["coords"][k], pb["values"][k])
    h.assemble_joint()
    assert h.factor() == 0                                    # positive definite
    pc = pb["pcoords"]
    for i in (0, 1):
        pred, err = h.predict(i, pc)                          # second call: resident factor, one sweep
        c0 = pv[i] ** 2 + pv[8 + i]                           # sigma_i^2 + nugget_i: the prior variance bounds the kriging variance
        assert np.all(np.isfinite(pred)) and np.all(np.isfinite(err))
        assert np.all(err >= 0.0) and np.all(err ** 2 <= c0 * (1 + 1e-12))
        assert err.min() < 0.9 * np.sqrt(c0)                  # data nearby do reduce the variance
        assert np.abs(pred).max() < 10.0 * max(np.abs(pb["values"][i]).max(), 1.0)
        # columns of the solve are independent: a subset of the points gives the same numbers, in either layout
        # (fewer than 256 points are not Hilbert-sorted by the library)
        sub = np.arange(0, len(pc), 41)
        p2, e2 = h.predict(i, pc[sub])
        np.testing.assert_allclose(p2, pred[sub], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(e2, err[sub], rtol=1e-9, atol=1e-11)
        p3, e3 = h.predict(i, pc)                             # repeatable bit for bit
        assert np.array_equal(p3, pred) and np.array_equal(e3, err)
    # prediction AT a datum of process 0 (nugget > 0: no exact interpolation, but closer to the datum than the prior)
    k = 12345
    pk, ek = h.predict(0, pb["coords"][0][k:k + 1])
    assert ek[0] ** 2 < pv[8] * 1.0001 + 1e-12               # never worse than the nugget: the datum itself is there
    t = h.timings()
    assert t["factor_ms"] > 0
    h.close()


def test_variogram_1M_soundings_properties():
    from sif_xco2_cokriging_amd import native
    from sif_xco2_cokriging_amd.variogram import variogram_arrays
    n = 1_000_000
    rng = np.random.default_rng(20005)
    c = np.column_stack([rng.uniform(22, 58, n), rng.uniform(-125, -65, n)])
    v = rng.standard_normal(n)
    h = native.Handle(0)
    h.set_metric(0)
    # (1) no cap: the bins partition ALL pairs
    cen, edg, mean, cnt = variogram_arrays(h, c, v, None, None, True, 1e9, 30)
    assert int(cnt.sum()) == n * (n - 1) // 2
    assert h.vario_stats()["bin_visited_pairs"] >= n * (n - 1) // 2
    # exact identity for centred values: sum over all pairs of 0.5 (a_i - a_j)^2 = n^2 var / 2 -- a checksum of all 5e11
    # cloud values through the binning (white noise: each bin's mean is the variance up to sampling noise)
    np.testing.assert_allclose(float((mean * cnt).sum()), 0.5 * n * n * v.var(), rtol=1e-9)
    np.testing.assert_allclose(mean, v.var(), rtol=5e-2)
    # (2) the headline case: culling on (Hilbert order) and off (caller's order) see the same pairs
    res = {}
    for order in (1, 0):
        hh = native.Handle(0)
        hh.set_option("site_order", order)
        hh.set_metric(0)
        res[order] = variogram_arrays(hh, c, v, None, None, True, 1500.0, 30)
        res[order] += (hh.vario_stats()["bin_visited_pairs"],)
        hh.close()
    assert np.array_equal(res[1][3], res[0][3])                                # counts
    assert np.array_equal(res[1][1], res[0][1])                                # edges: lo / hi are order-independent bits
    np.testing.assert_allclose(res[1][2], res[0][2], rtol=1e-10)
    assert res[1][4] < 0.4 * res[0][4]                                          # culling skipped most far tiles
    assert int(res[1][3].sum()) < n * (n - 1) // 2
    # (3) permutation invariance
    perm = rng.permutation(n)
    c2, e2, m2, k2 = variogram_arrays(h, c[perm], v[perm], None, None, True, 1500.0, 30)
    assert np.array_equal(k2, res[1][3]) and np.array_equal(e2, res[1][1])
    np.testing.assert_allclose(m2, res[1][2], rtol=1e-10)
    h.close()


def test_cross_variogram_1M_by_1M_soundings_properties():
    """BASELINE configs[4] AS WRITTEN: the empirical CROSS-semivariogram (i != j: all n_i n_j pairs,
    src/fields.py:201-204) of 1 000 000 x 1 000 000 soundings -- 1e12 pairs -- through what the domain offers at that size:
    the bins partition all pairs, a checksum of all cloud values, culling on / off, permutation of either point set."""
    from sif_xco2_cokriging_amd import native
    from sif_xco2_cokriging_amd.variogram import variogram_arrays
    n = 1_000_000
    rng = np.random.default_rng(20006)
    ci = np.column_stack([rng.uniform(22, 58, n), rng.uniform(-125, -65, n)])
    cj = np.column_stack([rng.uniform(22, 58, n), rng.uniform(-125, -65, n)])
    vi = rng.standard_normal(n)
    vj = 0.6 * rng.standard_normal(n) + 0.3
    h = native.Handle(0)
    h.set_metric(0)
    # (1) no cap: every one of the n_i n_j pairs lands in exactly one bin
    cen, edg, mean, cnt = variogram_arrays(h, ci, vi, cj, vj, False, 1e9, 30)
    assert int(cnt.sum()) == n * n
    assert h.vario_stats()["bin_visited_pairs"] >= n * n
    # sum over all pairs of 0.5 ((a_i - abar) - (b_j - bbar))^2 = 0.5 n_i n_j (var a + var b) exactly (src/fields.py:378-386
    # centres each field): a checksum of all 1e12 cloud values through the binning
    np.testing.assert_allclose(float((mean * cnt).sum()), 0.5 * n * n * (vi.var() + vj.var()), rtol=1e-9)
    np.testing.assert_allclose(mean, 0.5 * (vi.var() + vj.var()), rtol=5e-2)       # independent white noise: flat
    # (2) the headline case (1 500 km, 30 bins): culling on (Hilbert order) and off (caller's order) see the same pairs
    res = {}
    for order in (1, 0):
        hh = native.Handle(0)
        hh.set_option("site_order", order)
        hh.set_metric(0)
        res[order] = variogram_arrays(hh, ci, vi, cj, vj, False, 1500.0, 30)
        res[order] += (hh.vario_stats()["bin_visited_pairs"],)
        hh.close()
    assert np.array_equal(res[1][3], res[0][3]) and np.array_equal(res[1][1], res[0][1])
    np.testing.assert_allclose(res[1][2], res[0][2], rtol=1e-10)
    assert res[1][4] < 0.4 * res[0][4] and int(res[1][3].sum()) < n * n
    # (3) permutation of either point set; (4) swapping the two sets (the cloud value is symmetric in its arguments)
    pi, pj = rng.permutation(n), rng.permutation(n)
    c2, e2, m2, k2 = variogram_arrays(h, ci[pi], vi[pi], cj[pj], vj[pj], False, 1500.0, 30)
    assert np.array_equal(k2, res[1][3]) and np.array_equal(e2, res[1][1])
    np.testing.assert_allclose(m2, res[1][2], rtol=1e-10)
    c3, e3, m3, k3 = variogram_arrays(h, cj, vj, ci, vi, False, 1500.0, 30)
    assert np.array_equal(k3, res[1][3]) and np.array_equal(e3, res[1][1])
    np.testing.assert_allclose(m3, res[1][2], rtol=1e-10)
    h.close()
