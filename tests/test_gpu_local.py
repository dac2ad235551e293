"""GPU parity of the local-neighbourhood predictor (ck_predict_local) against fixtures produced by
the reference's point_prediction.Predictor._predict_chunk (src/point_prediction.py:243-249)."""
import warnings

import numpy as np
import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def _predictor(tag):
    from sif_xco2_cokriging_amd import fields, model, point_prediction
    g = load_golden("point_local")
    mod = model.MultivariateMatern(params=model.MaternParams().set_values(g[f"params_{tag}"]))
    mf = fields.MultiField([fields.Field(g["coords0"], g[f"values0_{tag}"]), fields.Field(g["coords1"], g[f"values1_{tag}"])])
    return point_prediction.Predictor(mod, mf), g


@pytest.mark.parametrize("tile_min", [None, 0, 10 ** 6])   # default classes | everything tiled | LDS + slab kernels
@pytest.mark.parametrize("tag", ["A", "R"])
def test_local_prediction_fixture(tag, tile_min):
    P, g = _predictor(tag)
    if tile_min is not None:
        P._handle().set_option("local_tile_min", tile_min)
    n_nan = 0
    for i in (0, 1):
        for md in (300, 1000):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                pred, err = P.predict_arrays(i, g[f"pcoords_{tag}"], max_dist=float(md))
            gp, ge = g[f"pred_{tag}_{i}_{md}"], g[f"pred_err_{tag}_{i}_{md}"]
            assert np.array_equal(np.isnan(pred), np.isnan(gp))      # empty neighbourhoods -> NaN
            assert np.array_equal(np.isnan(err), np.isnan(ge))
            ok = ~np.isnan(gp)
            n_nan += int((~ok).sum())
            np.testing.assert_allclose(pred[ok], gp[ok], rtol=1e-8, atol=1e-11)
            np.testing.assert_allclose(err[ok] ** 2, ge[ok] ** 2, rtol=1e-8, atol=1e-11)
    assert n_nan > 0


@pytest.mark.parametrize("tag", ["A", "R"])
def test_local_cross_validation_rule(tag):
    P, g = _predictor(tag)
    P.cv = True
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pred, err = P.predict_arrays(0, g["coords0"][:60], max_dist=700.0)
    np.testing.assert_allclose(pred, g[f"cv_pred_{tag}"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(err ** 2, g[f"cv_pred_err_{tag}"] ** 2, rtol=1e-8, atol=1e-11)


# "local_tile_min": neighbourhoods above it take the tiled path (batched 64-column steps on the MFMA tiles),
# those below it (and above the LDS limit) the one-workgroup-per-point slab kernel
TILE_MIN = [0, 256, 10 ** 6]


@pytest.mark.parametrize("tile_min", TILE_MIN)
def test_local_large_neighbourhood_matches_joint(tile_min):
    """max_dist = infinity makes every neighbourhood the whole data set (k = 400 > the LDS limit,
    global-scratch paths): the local predictor must then equal the joint one."""
    from sif_xco2_cokriging_amd import native
    g = load_golden("joint_solve")
    pv = g["params_A"]
    h = native.Handle(0)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(0)
    h.set_data(0, g["coords0_A"], g["values0_A"])
    h.set_data(1, g["coords1_A"], g["values1_A"])
    h.set_option("local_tile_min", tile_min)
    pred, err, info = h.predict_local(1, g["pcoords_A"], max_dist=1e9)
    assert info["k_max"] == 400 and info["n_empty"] == 0 and info["n_not_pd"] == 0
    ref = g["pred_A_1"]
    assert np.max(np.abs(pred - ref)) / np.max(np.abs(ref)) < 1e-8
    assert np.max(np.abs(err ** 2 - g["pred_err_A_1"] ** 2)) < 1e-9


@pytest.mark.parametrize("tile_min", TILE_MIN)
def test_local_large_neighbourhood_not_positive_definite(tile_min):
    """The large-neighbourhood paths (k = 520 > the LDS limit) on an indefinite model: every
    local system fails like the joint one does -> (NaN, NaN) per point (src/point_prediction.py:218-222)."""
    from sif_xco2_cokriging_amd import native
    g = load_golden("joint_not_pd")
    pv = g["params"]
    h = native.Handle(0)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(0)
    h.set_data(0, g["coords0"], np.zeros(260))
    h.set_data(1, g["coords1"], np.zeros(260))
    pc = g["coords0"][:7] + 0.013
    h.set_option("local_tile_min", tile_min)
    pred, err, info = h.predict_local(0, pc, max_dist=1e9)
    assert info["k_max"] == 520 and info["n_not_pd"] == 7
    assert np.all(np.isnan(pred)) and np.all(np.isnan(err))


@pytest.mark.parametrize("tile_min,group", [(0, 1), (0, 2), (0, 4), (256, 3), (10 ** 6, 4)])
def test_local_mid_sized_neighbourhoods_vs_oracle(tile_min, group):
    """Neighbourhoods of a few hundred sites (sizes that are not multiples of the column blocks or the
    tiles; a mix of the three size classes) against the oracle's per-point solves."""
    from sif_xco2_cokriging_amd import native, synth
    from oracle import cokrige_oracle as orc
    pb = synth.conus_problem(1500, seed=9)
    pv = pb["params"]
    h = native.Handle(0)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(0)
    for k in range(2):
        h.set_data(k, pb["coords"][k], pb["values"][k])
    pc = pb["pcoords"][::173][:40]
    h.set_option("local_tile_min", tile_min)
    h.set_option("local_group", group)   # 64-column blocks per trailing update of the tiled path
    pred, err, info = h.predict_local(0, pc, max_dist=900.0)
    assert info["k_max"] > 300
    h.set_option("local_slab_mb", 3)   # force several point batches through one small scratch slab
    pred_b, err_b, _ = h.predict_local(0, pc, max_dist=900.0)
    assert np.array_equal(pred, pred_b) and np.array_equal(err, err_b)
    rp, re = orc.local_predict(orc.Params.from_flat(pv), pb["coords"], pb["values"], pc, 0, 0, 900.0)[:2]
    np.testing.assert_allclose(pred, rp, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(err ** 2, re ** 2, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("case", ["euclid", "euclid_cv", "univariate", "haversine_cv_i1"])
@pytest.mark.parametrize("tile_min", [0, 10 ** 6])
def test_local_large_paths_variants_vs_oracle(case, tile_min):
    """The two large-neighbourhood paths on the variants the other tests do not reach: Euclidean metric,
    cross-validation mode (prediction points = data sites, d > 0 rule for the predicted process only), a
    univariate model, process 1 as the target."""
    from sif_xco2_cokriging_amd import native, synth
    from oracle import cokrige_oracle as orc
    if case.startswith("euclid"):
        pb, md = synth.unit_square_problem(450, grid_side=6, seed=31), 0.45
    else:
        pb, md = synth.conus_problem(500, seed=32), 1500.0
    pv = list(pb["params"])
    cv = case.endswith("cv") or "_cv_" in case
    i_pred = 1 if case.endswith("i1") else 0
    coords, values = pb["coords"], pb["values"]
    h = native.Handle(0)
    if case == "univariate":
        coords, values = coords[:1], values[:1]
        h.set_model(1, pv[0:1], pv[2:3], pv[5:6], pv[8:9])
        op = orc.Params.from_flat([pv[0], pv[2], pv[5], pv[8]])
    else:
        h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        op = orc.Params.from_flat(pv)
    h.set_metric(pb["metric"])
    for k in range(len(coords)):
        h.set_data(k, coords[k], values[k])
    h.set_option("local_tile_min", tile_min)
    pc = coords[i_pred][::9] if cv else pb["pcoords"][::211][:36]
    pred, err, info = h.predict_local(i_pred, pc, max_dist=md, cv=cv)
    assert info["k_max"] > 124 and info["n_not_pd"] == 0
    rp, re = orc.local_predict(op, coords, values, pc, i_pred, pb["metric"], md, cv)[:2]
    np.testing.assert_allclose(pred, rp, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(err ** 2, re ** 2, rtol=1e-8, atol=1e-9)


def test_local_radius_search_culling_is_exact():
    """The radius search skips whole 256-site chunks by a bounding-ball test.  With the sites in Hilbert order
    (default) almost every chunk is skipped, in the caller's order (site_order = 0) hardly any: neighbour counts,
    empty neighbourhoods and predictions must agree, also in cross-validation mode (d > 0 rule) and for a
    radius that reaches across the whole domain."""
    from sif_xco2_cokriging_amd import native, synth
    pb = synth.conus_problem(3000, seed=4)
    pv = pb["params"]
    pc = np.vstack([pb["pcoords"][::37], pb["coords"][0][:50], [[10.0, -170.0]]])   # grid, data sites, far away
    out = {}
    for order in (0, 1):
        h = native.Handle(0)
        h.set_option("site_order", order)
        h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        h.set_metric(0)
        for k in range(2):
            h.set_data(k, pb["coords"][k], pb["values"][k])
        out[order] = [h.predict_local(0, pc, max_dist=md, cv=cv) for md, cv in ((60.0, False), (150.0, True), (9000.0, False))]
    for (p0, e0, i0), (p1, e1, i1) in zip(out[0], out[1]):
        assert i0 == i1
        assert np.array_equal(np.isnan(p0), np.isnan(p1))
        np.testing.assert_allclose(p0, p1, rtol=1e-9, atol=1e-11, equal_nan=True)
        np.testing.assert_allclose(e0 ** 2, e1 ** 2, rtol=1e-9, atol=1e-11, equal_nan=True)   # variances: at a data
        # site the kriging variance is 0 up to cancellation noise, its square root is not comparable
    assert out[1][0][2]["n_empty"] >= 1        # the far-away point at 60 km


def test_predictor_call_signature_and_warnings():
    import pandas as pd
    P, g = _predictor("A")
    pc = pd.DataFrame(g["pcoords_A"], columns=["lat", "lon"])
    with pytest.warns(UserWarning, match="No data within maximum distance"):
        out = P(0, pc, max_dist=300.0, postprocess=False)
    df = out.to_dataframe().reset_index() if hasattr(out, "to_dataframe") else out.reset_index()
    assert {"pred", "pred_err"} <= set(df.columns)
    cv = P.cross_validation(0, max_dist=700.0, postprocess=False)
    assert list(cv.columns) == ["d1", "d2", "data", "pred", "residual", "pred_err"]
    assert len(cv) == 200


@pytest.mark.parametrize("tile_min", [0, None])
def test_local_system_sizes_around_the_padding_boundaries(tile_min):
    """max_dist = infinity makes every neighbourhood the whole data set, so the local system has exactly
    k = n0 + n1 sites: sizes around the tiled path's padding boundaries (k + 2 a multiple of 64: no identity
    padding at all; k + 2 one more: a whole block of it), around the LDS kernel's limit of 64, and two- / three-site
    systems."""
    from sif_xco2_cokriging_amd import native, synth
    from oracle import cokrige_oracle as orc
    pb = synth.conus_problem(120, seed=41)
    pv = pb["params"]
    op = orc.Params.from_flat(pv)
    pc = pb["pcoords"][::797][:9]
    for n0, n1 in [(1, 1), (2, 1), (30, 31), (31, 31), (32, 31), (32, 32), (33, 32), (63, 63), (64, 63),
                   (64, 64), (65, 64), (100, 90), (100, 91), (120, 7)]:
        coords = [pb["coords"][0][:n0], pb["coords"][1][:n1]]
        values = [pb["values"][0][:n0], pb["values"][1][:n1]]
        h = native.Handle(0)
        h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        h.set_metric(0)
        for k in range(2):
            h.set_data(k, coords[k], values[k])
        if tile_min is not None:
            h.set_option("local_tile_min", tile_min)
        for i in (0, 1):
            pred, err, info = h.predict_local(i, pc, max_dist=1e9)
            assert info["k_max"] == n0 + n1 and info["n_not_pd"] == 0, (n0, n1, info)
            rp, re = orc.local_predict(op, coords, values, pc, i, 0, 1e9)[:2]
            np.testing.assert_allclose(pred, rp, rtol=1e-8, atol=1e-10, err_msg=f"n0={n0} n1={n1} i={i}")
            np.testing.assert_allclose(err ** 2, re ** 2, rtol=1e-8, atol=1e-10, err_msg=f"n0={n0} n1={n1} i={i}")
        h.close()
