"""Pin the CPU oracle (oracle/cokrige_oracle.py) against fixtures produced by the
imported reference (tests/golden/make_fixtures.py), including the reference's one
known-answer test (research/simulation_experiment.ipynb:762-763,1410-1411)."""
import numpy as np
import pytest
from scipy.linalg import LinAlgError

from oracle import cokrige_oracle as orc
from tests.conftest import load_golden

HAV, EUC = orc.METRIC_HAVERSINE, orc.METRIC_EUCLID


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def test_kat_notebook_digits():
    g = load_golden("kat_simulation_experiment")
    # digits printed in the notebook
    assert np.allclose(g["pred"][:4], [1.025, 1.129, 1.177, 1.106], atol=6e-4)
    assert np.allclose(g["pred_err"][-3:], [0.6993, 0.7249, 0.754], atol=6e-4)
    p = orc.Params.from_flat(g["params"])
    pred, err = orc.joint_predict(p, [g["coords0"], g["coords1"]], [g["values0"], g["values1"]],
                                  g["pcoords"], 1, EUC)
    # nugget-free, cond ~1e7: solver-order noise is ~1e-9 relative
    assert rel(pred, g["pred"]) < 1e-8
    assert np.max(np.abs(err - g["pred_err"])) < 1e-6
    pu = orc.Params.from_flat(g["params_uni"])
    pred, err = orc.joint_predict(pu, [g["coords1"]], [g["values1"]], g["pcoords"], 0, EUC)
    assert rel(pred, g["pred_uni"]) < 1e-8
    assert np.max(np.abs(err - g["pred_err_uni"])) < 1e-6


def test_distances():
    g = load_golden("cov_blocks")
    assert rel(orc.haversine_km(g["A"], g["B"]), g["hav_AB"]) < 1e-14
    d = orc.haversine_km(g["A"], g["A"])
    assert rel(d, g["hav_AA"]) < 1e-14
    assert np.array_equal(d == 0, g["hav_AA"] == 0)  # exact-zero pattern (nugget semantics)
    assert rel(orc.euclid(g["A"], g["B"]), g["euc_AB"]) < 1e-15


def test_matern_correlation_grid():
    g = load_golden("kv_grid")
    for k, nu in enumerate(g["nus"]):
        r = orc.matern_correlation(nu, 1.0, g["h"])
        np.testing.assert_allclose(r, g["rho"][k], rtol=1e-15, atol=0)
        r = orc.matern_correlation(nu, 460.0, g["h"] * 460.0)
        np.testing.assert_allclose(r, g["rho_len460"][k], rtol=1e-15, atol=0)


@pytest.mark.parametrize("tag", ["A", "B", "R", "S"])
def test_cov_blocks(tag):
    g = load_golden("cov_blocks")
    p = orc.Params.from_flat(g[f"params_{tag}"])
    S = orc.joint_cov(p, [g["A"], g["B"]], HAV)
    np.testing.assert_allclose(S, g[f"Sigma_{tag}"], rtol=1e-13, atol=1e-300)
    for i in (0, 1):
        c0 = orc.pred_cross_cov(p, [g["A"], g["B"]], g["G"], i, HAV)
        np.testing.assert_allclose(c0, g[f"c0_{tag}_{i}"], rtol=1e-13, atol=1e-300)


def test_cov_blocks_euclid():
    g = load_golden("cov_blocks")
    p = orc.Params.from_flat(g["params_U"])
    np.testing.assert_allclose(orc.joint_cov(p, [g["U0"], g["U1"]], EUC), g["Sigma_U"], rtol=1e-13)
    np.testing.assert_allclose(orc.pred_cross_cov(p, [g["U0"], g["U1"]], g["UG"], 0, EUC), g["c0_U_0"], rtol=1e-13)


@pytest.mark.parametrize("tag", ["A", "R", "B"])
def test_joint_solve(tag):
    g = load_golden("joint_solve")
    p = orc.Params.from_flat(g[f"params_{tag}"])
    coords = [g[f"coords0_{tag}"], g[f"coords1_{tag}"]]
    values = [g[f"values0_{tag}"], g[f"values1_{tag}"]]
    for i in (0, 1):
        pred, err = orc.joint_predict(p, coords, values, g[f"pcoords_{tag}"], i, HAV)
        assert rel(pred, g[f"pred_{tag}_{i}"]) < 1e-10
        # compare variances: at prediction sites on nugget-free data sites var = 0 +- 1e-16,
        # whose square root is rounding noise (src/joint_prediction.py:78 zeroes the NaNs)
        assert np.max(np.abs(err ** 2 - g[f"pred_err_{tag}_{i}"] ** 2)) < 1e-12


def test_joint_not_pd_raises():
    g = load_golden("joint_not_pd")
    p = orc.Params.from_flat(g["params"])
    with pytest.raises(LinAlgError) as e:
        orc.joint_predict(p, [g["coords0"], g["coords1"]], [np.zeros(260), np.zeros(260)],
                          g["coords0"][:3], 0, HAV)
    assert str(e.value).startswith(f"{int(g['minor'])}-th leading minor")


def test_joint_loocv():
    g = load_golden("joint_loocv")
    p = orc.Params.from_flat(g["params"])
    for i in (0, 1):
        pred, err = orc.joint_loocv(p, [g["coords0"], g["coords1"]], [g["values0"], g["values1"]], i, HAV)
        assert rel(pred, g[f"pred_{i}"]) < 1e-11
        assert rel(err, g[f"pred_err_{i}"]) < 1e-11


@pytest.mark.parametrize("tag", ["A", "R"])
def test_point_local(tag):
    g = load_golden("point_local")
    p = orc.Params.from_flat(g[f"params_{tag}"])
    coords = [g["coords0"], g["coords1"]]
    values = [g[f"values0_{tag}"], g[f"values1_{tag}"]]
    n_nan = 0
    for i in (0, 1):
        for md in (300, 1000):
            pred, err = orc.local_predict(p, coords, values, g[f"pcoords_{tag}"], i, HAV, max_dist=float(md))
            gp, ge = g[f"pred_{tag}_{i}_{md}"], g[f"pred_err_{tag}_{i}_{md}"]
            assert np.array_equal(np.isnan(pred), np.isnan(gp))
            n_nan += int(np.isnan(gp).sum())
            ok = ~np.isnan(gp)
            np.testing.assert_allclose(pred[ok], gp[ok], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(err[ok] ** 2, ge[ok] ** 2, rtol=1e-9, atol=1e-12)
    assert n_nan > 0  # the empty-neighbourhood rule is exercised
    pred, err = orc.local_predict(p, coords, values, coords[0][:60], 0, HAV, max_dist=700.0, cv=True)
    np.testing.assert_allclose(pred, g[f"cv_pred_{tag}"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(err ** 2, g[f"cv_pred_err_{tag}"] ** 2, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("kind", ["semi", "cova"])
@pytest.mark.parametrize("md,nb", [(1500, 30), (600, 12)])
def test_variogram(kind, md, nb):
    g = load_golden("variogram")
    c = [g["coords0"], g["coords1"]]
    v = [g["values0"], g["values1"]]
    for (i, j) in ((0, 0), (0, 1), (1, 1)):
        centers, edges, means, counts = orc.variogram(c[i], v[i], c[j], v[j], i == j, HAV, float(md), nb,
                                                      covariogram=(kind == "cova"))
        key = f"{kind}_{md}_{nb}_{i}{j}"
        assert len(edges) == nb + 1
        np.testing.assert_allclose(centers, g[key + "_centers"], rtol=1e-12)
        np.testing.assert_allclose(edges, g[key + "_edges"], rtol=1e-12, atol=1e-12)
        assert np.array_equal(counts, g[key + "_counts"])
        np.testing.assert_allclose(means, g[key + "_means"], rtol=1e-12, atol=1e-15)


def test_variogram_euclid():
    g = load_golden("variogram")
    c = [g["e0"], g["e1"]]
    v = [g["w0"], g["w1"]]
    for (i, j) in ((0, 0), (0, 1), (1, 1)):
        centers, edges, means, counts = orc.variogram(c[i], v[i], c[j], v[j], i == j, EUC, 0.6, 15)
        key = f"euc_{i}{j}"
        assert np.array_equal(counts, g[key + "_counts"])
        np.testing.assert_allclose(means, g[key + "_means"], rtol=1e-12)
        np.testing.assert_allclose(centers, g[key + "_centers"], rtol=1e-12)


def test_sim_field_generator():
    """sim.BivariateRandomField (src/sim.py:45-54): cmat and L @ noise."""
    g = load_golden("sim_field")
    p = orc.Params.from_flat(g["params"])
    S = orc.joint_cov(p, [g["coords"], g["coords"]], EUC)
    np.testing.assert_allclose(S, g["cmat"], rtol=1e-13, atol=1e-300)
    z = np.linalg.cholesky(S) @ g["noise"]
    n = len(g["coords"])
    np.testing.assert_allclose(z[:n], g["field0"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(z[n:], g["field1"], rtol=1e-7, atol=1e-9)


def _fit_groups(g):
    return {(i, j): (g[f"centers_{i}{j}"], g[f"means_{i}{j}"], g[f"counts_{i}{j}"]) for (i, j) in ((0, 0), (0, 1), (1, 1))}


def test_composite_wls_and_fit_vs_reference():
    """MultivariateMatern._composite_wls at probe vectors, then fit() from the default start and from
    a guess with narrowed bounds (src/model.py:277-317), against the reference's own results."""
    g = load_golden("model_fit")
    groups = _fit_groups(g)
    cost = np.array([orc.composite_wls(p, groups) for p in g["probes"]])
    np.testing.assert_allclose(cost, g["probe_cost"], rtol=1e-12)
    x, c, ok = orc.fit(groups)
    assert ok
    np.testing.assert_allclose(c, float(g["fit_cost"]), rtol=1e-8)
    np.testing.assert_allclose(x, g["fit_x"], rtol=1e-5, atol=1e-7)
    b = list(orc.PARAM_BOUNDS)
    b[2:5] = [(0.3, 2.5)] * 3
    b[5:8] = [(2e2, 1e3)] * 3
    x, c, ok = orc.fit(groups, x0=g["guess_x0"], bounds=b)
    np.testing.assert_allclose(c, float(g["guess_fit_cost"]), rtol=1e-8)
    np.testing.assert_allclose(x, g["guess_fit_x"], rtol=1e-5, atol=1e-7)
    # the theoretical variograms FittedVariogram tabulates (src/model.py:330-331)
    p = orc.Params.from_flat(g["fit_x"])
    for (i, j) in ((0, 0), (0, 1), (1, 1)):
        sel = (g["theo_i"] == i) & (g["theo_j"] == j)
        np.testing.assert_allclose(orc.model_variogram(p, i, j, g["theo_distance"][sel]), g["theo_variogram"][sel], rtol=1e-12)


def test_variogram_lattice_fixtures():
    """Lattice data with max_dist / bin edges ON lattice distances: the counts hinge on the last bit of the
    distances and edges (tests/golden/make_fixtures.py: fixture_vario_lattice); the oracle must be exact there."""
    g = load_golden("variogram_lattice")
    c, v = [g["e0"], g["e1"]], [g["w0"], g["w1"]]
    for tag in ("a", "b", "c"):
        md, nb = float(g[f"euc_{tag}_cfg"][0]), int(g[f"euc_{tag}_cfg"][1])
        for kind in ("semi", "cova"):
            for (i, j) in ((0, 0), (0, 1), (1, 1)):
                centers, edges, means, counts = orc.variogram(c[i], v[i], c[j], v[j], i == j, EUC, md, nb,
                                                              covariogram=(kind == "cova"))
                key = f"euc_{tag}_{kind}_{i}{j}"
                assert np.array_equal(counts, g[key + "_counts"]), key
                np.testing.assert_allclose(centers, g[key + "_centers"], rtol=1e-15, atol=0)
                ok = counts > 0
                np.testing.assert_allclose(means[ok], g[key + "_means"][ok], rtol=1e-12, atol=1e-15)
    c, v = [g["c0"], g["c1"]], [g["v0"], g["v1"]]
    for tag in ("tie", "plain"):
        md, nb = float(g[f"hav_{tag}_cfg"][0]), int(g[f"hav_{tag}_cfg"][1])
        for (i, j) in ((0, 0), (0, 1), (1, 1)):
            centers, edges, means, counts = orc.variogram(c[i], v[i], c[j], v[j], i == j, HAV, md, nb)
            key = f"hav_{tag}_{i}{j}"
            assert np.array_equal(counts, g[key + "_counts"]), key
            np.testing.assert_array_equal(centers, g[key + "_centers"])
    assert g["hav_tie_00_n_at_maxdist"][1] > g["hav_tie_00_n_at_maxdist"][0] > 0   # the tie case really has ties


def test_verify_model_fixture():
    """The oracle's restatement of _verify_model (stacked (m+N)^2 factorisation) warns exactly where the reference did
    (tests/golden/make_fixtures.py: fixture_verify); the indefinite case has a POSITIVE prediction-variance diagonal."""
    import warnings
    g = load_golden("verify_model")
    assert g["indef_min_schur_diag"] > 0 > g["indef_min_schur_eig"]
    cases = [("indef", "indef", 0)] + [(t, "A", 0) for t in ("plain", "dup", "ondata", "onother")]
    for tag, dset, i in cases:
        p = orc.Params.from_flat(g[f"{dset}_params"])
        c, v = [g[f"{dset}_c0"], g[f"{dset}_c1"]], [g[f"{dset}_v0"], g[f"{dset}_v1"]]
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            pred, err = orc.joint_predict(p, c, v, g[f"{tag}_pc"], i, HAV, verify=True)
        warned = any("not positive definte" in str(x.message) for x in w)
        assert warned == bool(g[f"{tag}_warned"]), tag
        np.testing.assert_allclose(pred, g[f"{tag}_pred"], rtol=1e-9, atol=1e-12)
