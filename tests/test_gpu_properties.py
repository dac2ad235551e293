"""Size-independent properties of the joint predictor at the BASELINE sizes (configs 2 and 3),
plus oracle comparisons at the largest size the CPU oracle finishes in seconds."""
import numpy as np
import pytest

from oracle import cokrige_oracle as orc

pytestmark = pytest.mark.gpu


def _handle(pb, values=None, order=None, options=None):
    from sif_xco2_cokriging_amd import native
    h = native.Handle(0)
    for name, value in (options or {}).items():
        h.set_option(name, value)
    pv = pb["params"]
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(pb["metric"])
    for k in range(2):
        c = pb["coords"][k]
        v = (values if values is not None else pb["values"])[k]
        if order is not None:
            c, v = c[order[k]], v[order[k]]
        h.set_data(k, c, v)
    h.assemble_joint()
    assert h.factor() == 0
    return h


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("shape", ["unit_square", "conus"])
def test_against_oracle_n1500(shape):
    """both metrics, generic-nu and closed-form models, N = 3 000: the oracle needs ~10 s"""
    from sif_xco2_cokriging_amd import synth
    pb = synth.unit_square_problem(1500, grid_side=30) if shape == "unit_square" else synth.conus_problem(1500)
    pc = pb["pcoords"][::7][:900]
    h = _handle(pb)
    p = orc.Params.from_flat(pb["params"])
    for i in (0, 1):
        pred, err = h.predict(i, pc)
        rp, re = orc.joint_predict(p, pb["coords"], pb["values"], pc, i, pb["metric"])
        assert rel(pred, rp) < 1e-8, shape
        assert np.max(np.abs(err ** 2 - re ** 2)) < 1e-9


@pytest.mark.parametrize("group", [1, 2, 3, 4, 16])
def test_panel_groups(group):
    """The grouped factorisation / solve (trailing updates with K = 512 G from G panel buffers, option
    panel_group) for group sizes that do and do not divide the 7 panels -- predictions and LOOCV against the oracle."""
    from sif_xco2_cokriging_amd import synth
    pb = synth.conus_problem(1700, seed=11)
    pc = pb["pcoords"][::11][:700]
    h = _handle(pb, options={"panel_group": group})
    try:
        p = orc.Params.from_flat(pb["params"])
        pred, err = h.predict(1, pc)
        rp, re = orc.joint_predict(p, pb["coords"], pb["values"], pc, 1, pb["metric"])
        assert rel(pred, rp) < 1e-8
        assert np.max(np.abs(err ** 2 - re ** 2)) < 1e-9
        cp, ce = h.loocv(0, 1700)
        for ix in (0, 611, 1699):   # the reference's leave-one-out loop, three of its 1 700 solves
            op, oe = orc.joint_predict(p, pb["coords"], pb["values"], pb["coords"][0][ix], 0, pb["metric"], cv_ix=ix)
            assert abs(cp[ix] - op[0]) < 1e-8 * max(1.0, abs(op[0])) and abs(ce[ix] ** 2 - oe[0] ** 2) < 1e-9
    finally:
        h.close()


@pytest.mark.parametrize("fused,lookahead", [(0, 0), (1, 0), (2, 0), (3, 0), (2 | 4, 0), (2 | 16, 0), (16, 0), (2 | 16, 1), (2, 1),
                                             (2 | 16, -1)])
def test_panel_fused_variants(fused, lookahead):
    """option panel_fused: the 64-column sub-blocks inside a panel right-looking with three launches each (0) or
    left-looking with the update fused into the factor / row-solve launch, for the factorisation (bit 0) and the
    right-hand-side rows (bit 1); the diagonal block first (bit 2); the whole panel step in one launch of cooperating
    workgroups (bit 4, the default); with and without the panel step on a second stream under the trailing update
    (option lookahead) -- predictions and LOOCV against the oracle."""
    from sif_xco2_cokriging_amd import synth
    pb = synth.conus_problem(1500, seed=12)
    pc = pb["pcoords"][::13][:500]
    h = _handle(pb, options={"panel_fused": fused, "lookahead": lookahead})
    try:
        p = orc.Params.from_flat(pb["params"])
        pred, err = h.predict(0, pc)
        rp, re = orc.joint_predict(p, pb["coords"], pb["values"], pc, 0, pb["metric"])
        assert rel(pred, rp) < 1e-8
        assert np.max(np.abs(err ** 2 - re ** 2)) < 1e-9
        cp, ce = h.loocv(1, 1500)
        for ix in (3, 1499):
            op, oe = orc.joint_predict(p, pb["coords"], pb["values"], pb["coords"][1][ix], 1, pb["metric"], cv_ix=ix)
            assert abs(cp[ix] - op[0]) < 1e-8 * max(1.0, abs(op[0])) and abs(ce[ix] ** 2 - oe[0] ** 2) < 1e-9
    finally:
        h.close()


def test_config2_properties_n5000():
    """BASELINE config 2: n_obs = 5k per process on the unit square, 100 x 100 grid, Euclidean."""
    from sif_xco2_cokriging_amd import synth
    pb = synth.unit_square_problem(5000, grid_side=100)
    rng = np.random.default_rng(3)
    h1 = _handle(pb)
    p1, e1 = h1.predict(0, pb["pcoords"])
    assert p1.shape == (10000,) and np.all(np.isfinite(p1)) and np.all(e1 >= 0)
    c0 = pb["params"][0] ** 2 + pb["params"][8]
    assert np.all(e1 ** 2 <= c0 * (1 + 1e-12))                    # conditioning never adds variance
    # permutation invariance: shuffling the observations changes nothing but rounding
    order = [rng.permutation(5000), rng.permutation(5000)]
    h2 = _handle(pb, order=order)
    p2, e2 = h2.predict(0, pb["pcoords"])
    assert rel(p2, p1) < 1e-8 and np.max(np.abs(e2 ** 2 - e1 ** 2)) < 1e-9
    # linearity in the data: pred(a z + b w) = a pred(z) + b pred(w), same standard error
    w = [rng.standard_normal(5000), rng.standard_normal(5000)]
    hw = _handle(pb, values=w)
    pw, ew = hw.predict(0, pb["pcoords"])
    mix = [2.0 * pb["values"][k] - 0.5 * w[k] for k in range(2)]
    hm = _handle(pb, values=mix)
    pm, em = hm.predict(0, pb["pcoords"])
    assert rel(pm, 2.0 * p1 - 0.5 * pw) < 1e-8
    assert np.array_equal(em, e1) and np.array_equal(ew, e1)       # the error does not depend on the values
    # prediction at the observation sites of a process with nugget tau^2: var < tau^2 + small
    pa, ea = h1.predict(0, pb["coords"][0][:500])
    assert np.all(ea ** 2 <= pb["params"][8] + 1e-9)


def test_config3_properties_n20000():
    """BASELINE config 3 (headline size): N = 40 000, 8 833-point grid, haversine, generic nu."""
    from sif_xco2_cokriging_amd import synth
    pb = synth.conus_problem(20000)
    h = _handle(pb)
    for b in range(3):
        assert h.table_info(b)["enabled"]
    p0, e0 = h.predict(0, pb["pcoords"])
    p1, e1 = h.predict(1, pb["pcoords"])            # second call reuses the resident factor
    assert np.all(np.isfinite(p0)) and np.all(np.isfinite(p1))
    for i, e in ((0, e0), (1, e1)):
        c0 = pb["params"][i] ** 2 + pb["params"][8 + i]
        assert np.all(e >= 0) and np.all(e ** 2 <= c0 * (1 + 1e-12))
    # leave-one-out from the same factor agrees with predicting at a data site after withholding it
    lp, le = h.loocv(0, 20000)
    assert np.all(np.isfinite(lp)) and np.all(le > 0)
    # spot check three withheld sites against a direct solve on a neighbourhood-free sub-problem is
    # not possible at this size; check the identity residual / variance against the predictor instead:
    # at data sites the conditional variance given ALL data is below the leave-one-out variance
    pa, ea = h.predict(0, pb["coords"][0][:200])
    assert np.all(ea ** 2 <= le[:200] ** 2 + 1e-9)


def test_defaults_across_the_schedule_boundaries_agree_with_the_plain_schedule():
    """The factorisation picks its schedule by size -- per-panel updates below 12 panels, the panel step on a second
    stream under the update from 12 to 63, groups of four panels from 64 on (40 without the cooperative panel step) --:
    at 14 panels (N = 7 000: the automatic look-ahead) the default agrees with the strictly sequential, one-launch-per-
    dependency schedule to rounding, for both processes, and reports a not-positive-definite Sigma at the same index."""
    from sif_xco2_cokriging_amd import native, synth
    pb = synth.conus_problem(3500, seed=13)
    pc = pb["pcoords"][::9][:800]
    h1 = _handle(pb)                                                    # defaults
    h2 = _handle(pb, options={"panel_fused": 0, "lookahead": 0, "panel_group": 1})
    try:
        assert h1.num_panels()[0] == 14
        for i in (0, 1):
            a, b = h1.predict(i, pc), h2.predict(i, pc)
            assert rel(a[0], b[0]) < 1e-11 and rel(a[1], b[1]) < 1e-11
        assert h1.timings()["panel_coop_redone"] == 0
    finally:
        h1.close()
        h2.close()
    # not positive definite (rho_12 beyond the admissible range of the bivariate Matern): same minor from both schedules
    pv = list(pb["params"])
    pv[10] = -0.999
    pv[3] = 3.4          # nu_12 far above (nu_11 + nu_22) / 2
    infos = []
    for opts in ({}, {"panel_fused": 0, "lookahead": 0, "panel_group": 1}):
        h = native.Handle(0)
        for k, v in opts.items():
            h.set_option(k, v)
        h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        h.set_metric(pb["metric"])
        for k in range(2):
            h.set_data(k, pb["coords"][k], pb["values"][k])
        h.assemble_joint()
        infos.append(h.factor())
        h.close()
    assert infos[0] == infos[1] and infos[0] > 0, infos
