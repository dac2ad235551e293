"""GPU parity tests: the HIP path through the C ABI (libcokrige_hip.so) against the CPU
oracle and against the golden fixtures generated from the reference."""
import numpy as np
import pytest
from numpy.linalg import LinAlgError

from oracle import cokrige_oracle as orc
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

HAV, EUC = 0, 1


@pytest.fixture(scope="module")
def native():
    from sif_xco2_cokriging_amd import native as nat
    assert nat.device_count() >= 1
    return nat


def handle_for(native, params, metric):
    p = orc.Params.from_flat(params)
    h = native.Handle(0)
    if p.n_procs == 2:
        h.set_model(2, p.sigma, [p.nu[0, 0], p.nu[0, 1], p.nu[1, 1]],
                    [p.len_scale[0, 0], p.len_scale[0, 1], p.len_scale[1, 1]], p.nugget, p.rho)
    else:
        h.set_model(1, p.sigma, [p.nu[0, 0]] * 3, [p.len_scale[0, 0]] * 3, p.nugget, 0.0)
    h.set_metric(metric)
    return h, p


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def test_mfma_f64_layout(native):
    """v_mfma_f64_16x16x4_f64: lane l, register r holds D[(l >> 4) + 4 r][l & 15]."""
    h = native.Handle(0)
    pr = h.mfma_probe()
    lane = np.arange(64)[:, None]
    reg = np.arange(4)[None, :]
    print("probe rows lane0..3:", pr[:4, :, 0].tolist(), "lane16:", pr[16, :, 0].tolist())
    assert np.array_equal(pr[:, :, 2], np.full((64, 4), 7))
    assert np.array_equal(pr[:, :, 1], np.broadcast_to(lane & 15, (64, 4)))
    assert np.array_equal(pr[:, :, 0], (lane >> 4) + 4 * reg)


def test_distance_dense(native):
    g = load_golden("cov_blocks")
    h = native.Handle(0)
    h.set_metric(HAV)
    d = h.distance_dense(g["A"], g["B"])
    assert rel(d, g["hav_AB"]) < 1e-13
    d = h.distance_dense(g["A"], g["A"])
    assert rel(d, g["hav_AA"]) < 1e-13
    assert np.array_equal(d == 0, g["hav_AA"] == 0)   # exact zeros where coordinates coincide
    h.set_metric(EUC)
    assert rel(h.distance_dense(g["A"], g["B"]), g["euc_AB"]) < 1e-15


def test_matern_correlation_grid(native):
    """rho(h) for every nu of the fixture grid: general K_nu (Temme / CF2) and closed forms."""
    g = load_golden("kv_grid")
    for k, nu in enumerate(g["nus"]):
        h = native.Handle(0)
        h.set_model(1, [1.0], [nu] * 3, [1.0] * 3, [0.0], 0.0)
        r = h.cov_lags(0, 0, g["h"], use_nugget=False)
        ref = g["rho"][k]
        # scipy's kv itself is only good to ~1e-13 relative around x = 2 (AMOS regime switch);
        # where rho < 1e-290 the reference has already lost digits to subnormal K_nu
        big = ref > 1e-290
        assert np.max(np.abs(r[big] / ref[big] - 1)) < 5e-13, nu
        assert np.max(np.abs(r[~big] - ref[~big])) < 1e-290
        h2 = native.Handle(0)
        h2.set_model(1, [1.0], [nu] * 3, [460.0] * 3, [0.0], 0.0)
        r = h2.cov_lags(0, 0, g["h"] * 460.0, use_nugget=False)
        ref = g["rho_len460"][k]
        big = ref > 1e-290
        assert np.max(np.abs(r[big] / ref[big] - 1)) < 5e-13, nu


@pytest.mark.parametrize("tag", ["A", "B", "R", "S"])
def test_cov_dense_blocks(native, tag):
    g = load_golden("cov_blocks")
    h, p = handle_for(native, g[f"params_{tag}"], HAV)
    S = g[f"Sigma_{tag}"]
    nA = len(g["A"])
    b00 = h.cov_dense(0, 0, g["A"], g["A"])
    b01 = h.cov_dense(0, 1, g["A"], g["B"])
    b11 = h.cov_dense(1, 1, g["B"], g["B"])
    for got, ref in ((b00, S[:nA, :nA]), (b01, S[:nA, nA:]), (b11, S[nA:, nA:])):
        np.testing.assert_allclose(got, ref, rtol=5e-13, atol=1e-300)
    for i in (0, 1):
        c0 = np.vstack([h.cov_dense(i, 0, g["A"], g["G"]) if i == 0 else h.cov_dense(1, 0, g["A"], g["G"]),
                        h.cov_dense(i, 1, g["B"], g["G"])])
        np.testing.assert_allclose(c0, g[f"c0_{tag}_{i}"], rtol=5e-13, atol=1e-300)


def test_gemm_nt_mfma(native):
    """the plain MFMA GEMM entry (128 x 128 register-staged tile; 256 x 64 tile for narrow N) against torch fp64"""
    import torch
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    h = native.Handle(0)
    for (M, N, K, lower) in ((256, 128, 16, False), (512, 256, 64, False), (768, 192, 64, False),
                             (1024, 512, 512, True), (256, 64, 64, False)):
        A = torch.randn(M, K, dtype=torch.float64, device=dev)
        B = torch.randn(N, K, dtype=torch.float64, device=dev)
        C = torch.randn(M, N, dtype=torch.float64, device=dev)
        ref = (C - A @ B.T).cpu().numpy()
        C0 = C.clone()
        torch.cuda.synchronize()
        h.dev_gemm_nt(C.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K, lower=lower)
        h.synchronize()
        got = C.cpu().numpy()
        if lower:
            # tiles strictly above the diagonal are skipped (left untouched)
            BN = 128 if N % 128 == 0 else 64
            BM = 128 if N % 128 == 0 else 256
            c0 = C0.cpu().numpy()
            for tm in range(M // BM):
                for tn in range(N // BN):
                    blk = (slice(tm * BM, tm * BM + BM), slice(tn * BN, tn * BN + BN))
                    if tm * BM + BM - 1 < tn * BN:
                        assert np.array_equal(got[blk], c0[blk])
                    else:
                        np.testing.assert_allclose(got[blk], ref[blk], rtol=1e-12, atol=1e-11)
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-11)


def _assembled(native, params, coords, values, metric, exact=False, site_order=None):
    h, p = handle_for(native, params, metric)
    for k in range(p.n_procs):
        h.set_data(k, coords[k], values[k])
    if exact:
        h.set_option("exact_cov", 1)
    if site_order is not None:
        h.set_option("site_order", site_order)
    h.assemble_joint()
    return h, p


def _internal_index(h, coords):
    """Stacked caller indices in the handle's internal order (ck_debug_site_order)."""
    n0 = len(coords[0])
    return np.concatenate([h.debug_site_order(0, n0), n0 + h.debug_site_order(1, len(coords[1]))])


@pytest.mark.parametrize("site_order", [0, 1])
@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("tag", ["A", "R"])
def test_assemble_and_factor_vs_oracle(native, exact, tag, site_order):
    """Sigma entry by entry (tabulated and per-entry Bessel paths), then L against numpy -- in the
    caller's order (site_order 0) and in the library's Hilbert order (1: Sigma permuted symmetrically)."""
    g = load_golden("joint_solve")
    coords = [g[f"coords0_{tag}"], g[f"coords1_{tag}"]]
    values = [g[f"values0_{tag}"], g[f"values1_{tag}"]]
    h, p = _assembled(native, g[f"params_{tag}"], coords, values, HAV, exact=exact, site_order=site_order)
    if not exact:
        for b in range(3):
            ti = h.table_info(b)
            assert ti["enabled"] and ti["max_rel_err"] < 2e-13, ti
    N = 400
    S = orc.joint_cov(p, coords, HAV)
    idx = _internal_index(h, coords)
    if site_order == 0:
        assert np.array_equal(idx, np.arange(N))
    else:
        assert sorted(idx.tolist()) == list(range(N)) and not np.array_equal(idx, np.arange(N))
    S = S[np.ix_(idx, idx)]
    low = h.debug_get_lower(N)
    if exact:
        # 5e-13 relative; entries below 1e-30 (e^-70 of the variance) only to 1e-30 absolute
        np.testing.assert_allclose(low, np.tril(S), rtol=5e-13, atol=1e-30)
    else:
        # the table is gated on |err| <= 2e-13 |amp| max(rho, 1e-6) (k_table_check): relative down to
        # rho = 1e-6, absolute in units of 1e-6 of the variance below
        np.testing.assert_allclose(low, np.tril(S), rtol=5e-13, atol=5e-19 * np.max(np.diag(S)))
    assert h.factor() == 0
    L = h.debug_get_lower(N)
    Lref = np.linalg.cholesky(S)
    assert rel(L, Lref) < 1e-11
    assert rel(L @ L.T, S) < 1e-13


@pytest.mark.parametrize("tag", ["A", "R", "B"])
def test_joint_predict_fixture(native, tag):
    g = load_golden("joint_solve")
    coords = [g[f"coords0_{tag}"], g[f"coords1_{tag}"]]
    values = [g[f"values0_{tag}"], g[f"values1_{tag}"]]
    h, p = _assembled(native, g[f"params_{tag}"], coords, values, HAV)
    assert h.factor() == 0
    for i in (0, 1):
        pred, err = h.predict(i, g[f"pcoords_{tag}"])
        # north-star tolerance: 1e-6 relative to the scipy path; we hold 1e-9
        assert rel(pred, g[f"pred_{tag}_{i}"]) < 1e-9
        assert np.max(np.abs(err ** 2 - g[f"pred_err_{tag}_{i}"] ** 2)) < 1e-10


def test_kat_simulation_experiment(native):
    """The reference's one recorded known-answer test (simulation_experiment.ipynb:762-763)."""
    g = load_golden("kat_simulation_experiment")
    h, p = _assembled(native, g["params"], [g["coords0"], g["coords1"]], [g["values0"], g["values1"]], EUC)
    assert h.factor() == 0
    pred, err = h.predict(1, g["pcoords"])
    assert np.allclose(pred[:4], [1.025, 1.129, 1.177, 1.106], atol=6e-4)
    assert np.allclose(pred[-3:], [-0.3236, -0.2804, -0.2439], atol=6e-5)
    assert np.allclose(err[:4], [0.2072, 0.1824, 0.1494, 0.0871], atol=6e-5)
    assert np.allclose(err[-3:], [0.6993, 0.7249, 0.754], atol=6e-4)
    # nugget-free, cond(Sigma) ~ 1e7: the oracle's own solver-order noise is ~1e-9
    assert rel(pred, g["pred"]) < 1e-6
    assert np.max(np.abs(err - g["pred_err"])) < 1e-6
    # univariate kriging of process 1 alone (notebook cell 14)
    hu, pu = _assembled(native, g["params_uni"], [g["coords1"]], [g["values1"]], EUC)
    assert hu.factor() == 0
    pred, err = hu.predict(0, g["pcoords"])
    assert rel(pred, g["pred_uni"]) < 1e-6
    assert np.max(np.abs(err - g["pred_err_uni"])) < 1e-6


@pytest.mark.parametrize("site_order", [0, 1])
def test_not_positive_definite_reports_minor(native, site_order):
    """numpy's index of the failing leading minor -- also when the library factorised in its own
    (Hilbert) order first: ck_factor repeats the factorisation in the caller's order to report it."""
    g = load_golden("joint_not_pd")
    h, p = _assembled(native, g["params"], [g["coords0"], g["coords1"]], [np.zeros(260), np.zeros(260)], HAV,
                      site_order=site_order)
    info = h.factor()
    assert info == int(g["minor"])


def test_site_order_invariance(native):
    """Predictions and LOOCV results do not depend on the internal site order (Sigma is permuted
    symmetrically); prediction points come back in the caller's order; ck_sample refuses the Hilbert order."""
    from sif_xco2_cokriging_amd import synth
    pb = synth.conus_problem(700, params=synth.SET_A, seed=5)
    pb["pcoords"] = pb["pcoords"][::9]   # 982 points: enough for the library to sort them (>= 256)
    out = []
    for so in (0, 1):
        h, p = _assembled(native, pb["params"], pb["coords"], pb["values"], HAV, site_order=so)
        assert h.factor() == 0
        pr = [h.predict(i, pb["pcoords"]) for i in (0, 1)]
        cv = [h.loocv(i, 700) for i in (0, 1)]
        out.append((pr, cv))
        if so == 1:
            with pytest.raises(RuntimeError, match="site_order"):
                h.sample(np.zeros(1400))
            # switching the order after the layout lays the sites out again on the next assemble
            h.set_option("site_order", 0)
            h.assemble_joint()
            assert h.factor() == 0
            again = h.predict(0, pb["pcoords"])
            assert rel(again[0], pr[0][0]) < 1e-9 and rel(again[1], pr[0][1]) < 1e-9
            assert np.array_equal(h.debug_site_order(0, 700), np.arange(700))
        h.close()
    for i in (0, 1):
        for a, b in zip(out[0][0][i] + out[0][1][i], out[1][0][i] + out[1][1][i]):
            assert rel(a, b) < 1e-9


def test_predictor_class_api(native):
    """The reference-signature classes end to end (Predictor over MultivariateMatern + MultiField)."""
    import pandas as pd
    from sif_xco2_cokriging_amd import fields, joint_prediction, model
    g = load_golden("joint_solve")
    mp = model.MaternParams().set_values(g["params_A"])
    mod = model.MultivariateMatern(params=mp)
    mf = fields.MultiField([fields.Field(g["coords0_A"], g["values0_A"]), fields.Field(g["coords1_A"], g["values1_A"])])
    P = joint_prediction.Predictor(mod, mf)
    pc = pd.DataFrame(g["pcoords_A"], columns=["lat", "lon"])
    with pytest.warns(UserWarning):   # prediction sites on data sites (fixture rows 0..5)
        out = P(1, pc, postprocess=False)
    df = out.to_dataframe().reset_index() if hasattr(out, "to_dataframe") else out.reset_index()
    df = pc.merge(df, on=["lat", "lon"], how="left")
    assert rel(df["pred"].values, g["pred_A_1"]) < 1e-9
    # a grid larger than the right-hand-side budget goes through the resident factor in batches
    big = np.tile(g["pcoords_A"], (30, 1))[:2600]
    whole = P.predict_arrays(1, big)
    P.rhs_budget_bytes = 1      # -> batches of 1 024 points
    batched = P.predict_arrays(1, big)
    assert rel(batched[0], whole[0]) < 1e-12 and rel(batched[1], whole[1]) < 1e-12
    P.rhs_budget_bytes = 48 << 30
    # model mirror
    hlag = np.array([0.0, 1.0, 50.0, 700.0])
    po = orc.Params.from_flat(g["params_A"])
    np.testing.assert_allclose(mod.covariance(0, hlag), orc.covariance(po, 0, hlag), rtol=5e-13)
    np.testing.assert_allclose(mod.cross_covariance(1, 0, hlag), orc.cross_covariance(po, 0, 1, hlag), rtol=5e-13)
    # not-PD model raises like scipy's cho_factor
    g2 = load_golden("joint_not_pd")
    mod2 = model.MultivariateMatern(params=model.MaternParams().set_values(g2["params"]))
    mf2 = fields.MultiField([fields.Field(g2["coords0"], np.zeros(260)), fields.Field(g2["coords1"], np.zeros(260))])
    with pytest.raises(LinAlgError) as e:
        joint_prediction.Predictor(mod2, mf2)(0, pc, postprocess=False)
    assert str(e.value) == str(g2["message"])


@pytest.mark.parametrize("refactor_each", [False, True])
def test_joint_loocv(native, refactor_each):
    """leave-one-out: from one factorisation (ck_loocv) and by the reference's n-solve loop"""
    from sif_xco2_cokriging_amd import fields, joint_prediction, model
    g = load_golden("joint_loocv")
    mod = model.MultivariateMatern(params=model.MaternParams().set_values(g["params"]))
    mf = fields.MultiField([fields.Field(g["coords0"], g["values0"]), fields.Field(g["coords1"], g["values1"])])
    P = joint_prediction.Predictor(mod, mf)
    for i in (0, 1):
        df = P.cross_validation(i, postprocess=False, refactor_each=refactor_each)
        # rows come back sorted by the coordinates, like the reference's xr.merge + outer merge (:248-254)
        c = g[f"coords{i}"]
        o = np.lexsort((c[:, 1], c[:, 0]))
        np.testing.assert_array_equal(df[["d1", "d2"]].values, c[o])
        assert rel(df["pred"].values, g[f"pred_{i}"][o]) < 1e-9
        assert rel(df["pred_err"].values, g[f"pred_err_{i}"][o]) < 1e-9
        np.testing.assert_allclose(df["residual"].values, (g[f"values{i}"] - g[f"pred_{i}"])[o], rtol=1e-7, atol=1e-9)


def test_sim_field_draw(native):
    """sim.BivariateRandomField: cmat assembly + Cholesky + L @ noise against the reference's draw
    (fixture from src/sim.py on a 13 x 11 grid, seed 7)."""
    from sif_xco2_cokriging_amd import model, sim
    g = load_golden("sim_field")
    grid = sim.CartesianGrid(xcount=13, ycount=11)
    np.testing.assert_allclose(grid.coords.values, g["coords"], rtol=0, atol=0)
    mod = model.MultivariateMatern(params=model.MaternParams().set_values(g["params"]))
    rf = sim.BivariateRandomField(mod, grid, seed=7)
    np.testing.assert_allclose(rf.fields[0]["value"].values, g["field0"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(rf.fields[1]["value"].values, g["field1"], rtol=1e-7, atol=1e-9)
    samples = rf.sample(size=40, epsilon=0.1)
    mf = rf.to_fields(samples)
    assert mf.n_procs == 2 and mf.fields[0].size == 40 and mf.fields[1].size == 40
    # co-located half: the first ceil(40/2) sampled sites are shared by the two processes
    a = {tuple(r) for r in mf.fields[0].coords}
    b = {tuple(r) for r in mf.fields[1].coords}
    assert len(a & b) == 20


def test_predict_zero_points(native):
    g = load_golden("joint_loocv")
    h, p = _assembled(native, g["params"], [g["coords0"], g["coords1"]], [g["values0"], g["values1"]], HAV)
    assert h.factor() == 0
    pred, err = h.predict(0, np.zeros((0, 2)))
    assert pred.shape == (0,) and err.shape == (0,)


def test_notebook_flow_end_to_end(native):
    """The reference's simulation_experiment notebook, every step on the GPU (simulate -> sample ->
    cokrige), against the digits the notebook records (research/simulation_experiment.ipynb:762-763)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ck_example_simexp", os.path.join(root, "examples", "simulation_experiment.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pred, err = mod.run()
    assert pred.shape == (2601,)
    assert np.allclose(pred[:4], [1.025, 1.129, 1.177, 1.106], atol=6e-4)
    assert np.allclose(pred[-3:], [-0.3236, -0.2804, -0.2439], atol=6e-5)
    assert np.allclose(err[:4], [0.2072, 0.1824, 0.1494, 0.0871], atol=6e-5)
    assert np.allclose(err[-3:], [0.6993, 0.7249, 0.754], atol=6e-4)
    g = load_golden("kat_simulation_experiment")
    assert rel(pred, g["pred"]) < 1e-5


@pytest.mark.parametrize("metric", [HAV, EUC])
def test_table_path_across_the_parameter_box(native, metric):
    """The tabulated assembly against the per-entry evaluator over the reference's whole parameter box
    (src/model.py:122-129: nu in [0.2, 3.5], len_scale in [100, 2000] km, nugget in [0, 0.2]) including its
    corners, on both metrics: wherever a block's table passed its gate the entries agree to the gate's
    measure, and a table that fails the gate must have handed the block to the exact kernel."""
    rng = np.random.default_rng(77)
    n0, n1 = 190, 170
    if metric == HAV:
        c0 = np.column_stack([rng.uniform(25, 50, n0), rng.uniform(-120, -70, n0)])
        c1 = np.column_stack([rng.uniform(25, 50, n1), rng.uniform(-120, -70, n1)])
        ls_lo, ls_hi = 100.0, 2000.0
    else:
        c0, c1 = rng.random((n0, 2)), rng.random((n1, 2))
        ls_lo, ls_hi = 0.02, 1.5
    c1[:40] = c0[:40]                                   # co-located sites
    z = [np.zeros(n0), np.zeros(n1)]
    corners = [(0.2, 0.2, 0.2), (3.5, 3.5, 3.5), (0.2, 1.85, 3.5), (0.5, 1.5, 2.5)]
    cases = []
    for nu in corners:
        for ls in ((ls_lo,) * 3, (ls_hi,) * 3, (ls_lo, 0.5 * (ls_lo + ls_hi), ls_hi)):
            cases.append([1.3, 0.6, *nu, *ls, 0.0, 0.2, 0.35])
    for _ in range(6):
        cases.append([rng.uniform(0.4, 3.5), rng.uniform(0.4, 3.5), *rng.uniform(0.2, 3.5, 3),
                      *rng.uniform(ls_lo, ls_hi, 3), *rng.uniform(0.0, 0.2, 2), rng.uniform(-0.9, 0.9)])
    for pv in cases:
        pv = np.array(pv, dtype=float)
        low = {}
        for exact in (0, 1):
            h, _ = _assembled(native, pv, [c0, c1], z, metric, exact=bool(exact), site_order=0)
            low[exact] = h.debug_get_lower(n0 + n1)
            if not exact:
                info = [h.table_info(b) for b in range(3)]
            h.close()
        scale = np.max(np.abs(np.diag(low[1])))
        if all(t["enabled"] for t in info):
            assert max(t["max_rel_err"] for t in info) < 2e-13
            np.testing.assert_allclose(low[0], low[1], rtol=5e-13, atol=5e-19 * scale, err_msg=str(pv))
        else:   # a gated-out table: the whole assembly went through the exact kernel
            assert np.array_equal(low[0], low[1]), pv


@pytest.mark.parametrize("tag", ["A", "R"])
def test_parity_in_units_of_the_oracles_own_perturbation_floor(native, tag):
    """the tolerance made interpretable.  Nudging every input coordinate by one ulp moves
    the ORACLE's predictions by `floor` (cond(Sigma) times rounding); the HIP path has to stay within a
    small multiple of that floor (and of the 1e-6 the north star asks for by a wide margin)."""
    g = load_golden("joint_solve")
    coords = [g[f"coords0_{tag}"], g[f"coords1_{tag}"]]
    values = [g[f"values0_{tag}"], g[f"values1_{tag}"]]
    p = orc.Params.from_flat(g[f"params_{tag}"])
    pc = g[f"pcoords_{tag}"]
    ref, ref_err = orc.joint_predict(p, coords, values, pc, 0, HAV)
    nudged = [np.nextafter(c, np.inf) for c in coords]
    alt, alt_err = orc.joint_predict(p, nudged, values, np.nextafter(pc, np.inf), 0, HAV)
    floor = max(rel(alt, ref), 1e-16)
    cond = np.linalg.cond(orc.joint_cov(p, coords, HAV))
    h, _ = _assembled(native, g[f"params_{tag}"], coords, values, HAV)
    assert h.factor() == 0
    pred, err = h.predict(0, pc)
    e = rel(pred, ref)
    print(f"set {tag}: cond(Sigma) = {cond:.3g}, oracle 1-ulp floor = {floor:.2e}, HIP vs oracle = {e:.2e}")
    assert e < 1e-9 and e < 200.0 * floor + 1e-13


def _hilbert_keys(xy, lo, hi):
    """numpy restatement of the library's order-16 Hilbert key (csrc/ck_api.hip: hilbert_key / hilbert_order)."""
    s = np.where(hi > lo, 65536.0 / np.where(hi > lo, hi - lo, 1.0), 0.0)
    f = np.clip((xy - lo) * s, 0.0, 65535.0)
    x, y = f[:, 0].astype(np.uint64), f[:, 1].astype(np.uint64)
    d = np.zeros(len(xy), dtype=np.uint64)
    lvl = 32768
    while lvl > 0:
        rx = ((x & np.uint64(lvl)) != 0).astype(np.uint64)
        ry = ((y & np.uint64(lvl)) != 0).astype(np.uint64)
        d += np.uint64(lvl) * np.uint64(lvl) * ((np.uint64(3) * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, np.uint64(lvl - 1) - x, x)
        y = np.where(flip, np.uint64(lvl - 1) - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        lvl >>= 1
    return d


@pytest.mark.parametrize("n0", [3000, 9000])   # comparison sort below 4 096 points, threaded keys + radix sort above
def test_hilbert_site_order_matches_its_definition(native, n0):
    """The internal site order is the STABLE sort by Hilbert key (coincident sites and equal keys keep the caller's
    order): both host implementations against a numpy restatement, with duplicated sites in the data."""
    rng = np.random.default_rng(8)
    c0 = np.column_stack([rng.uniform(25, 50, n0), rng.uniform(-120, -70, n0)])
    c0[n0 // 2: n0 // 2 + 200] = c0[:200]                 # coincident sites
    c0[-50:] = np.round(c0[-50:], 1)                      # a coarse lattice: more equal keys
    c1 = np.column_stack([rng.uniform(20, 55, 300), rng.uniform(-125, -65, 300)])   # widens the bounding box
    pv = [0.99, 0.81, 0.39, 0.695, 1.0, 460.0, 460.0, 460.0, 0.02, 0.025, -0.19]
    h = native.Handle(0)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(HAV)
    h.set_data(0, c0, np.zeros(n0))
    h.set_data(1, c1, np.zeros(300))
    allc = np.vstack([c0, c1])
    lo, hi = allc.min(axis=0), allc.max(axis=0)
    for k, c in ((0, c0), (1, c1)):
        ref = np.argsort(_hilbert_keys(c, lo, hi), kind="stable")
        assert np.array_equal(h.debug_site_order(k, len(c)), ref)
    h.close()


def _verify_predictor(g, dset):
    from sif_xco2_cokriging_amd import fields, joint_prediction, model
    mod = model.MultivariateMatern(params=model.MaternParams().set_values(g[f"{dset}_params"]))
    mf = fields.MultiField([fields.Field(g[f"{dset}_c0"], g[f"{dset}_v0"]), fields.Field(g[f"{dset}_c1"], g[f"{dset}_v1"])])
    return joint_prediction.Predictor(mod, mf)


def test_verify_model_matches_reference(native):
    """_verify_model (src/joint_prediction.py:60-66, 260-274) as the Cholesky of the m x m Schur complement on the
    device: warns exactly where the reference's (m+N)^2 factorisation did -- including the indefinite stacked matrix
    whose prediction variances are all positive (the variance test of round 1 stayed silent there)."""
    import warnings
    g = load_golden("verify_model")
    for tag, dset in (("indef", "indef"), ("plain", "A"), ("dup", "A"), ("ondata", "A"), ("onother", "A")):
        P = _verify_predictor(g, dset)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            pred, err = P.predict_arrays(0, g[f"{tag}_pc"])
            P._warn_if_invalid(err)
        warned = any("not positive definte" in str(x.message) for x in w)
        assert warned == bool(g[f"{tag}_warned"]), tag
        assert rel(pred, g[f"{tag}_pred"]) < 1e-9, tag
        # a site on a datum has variance 0 up to rounding: sqrt(1e-16 sigma^2) = 1e-8 either way
        np.testing.assert_allclose(err, g[f"{tag}_pred_err"], rtol=1e-8, atol=1e-7, err_msg=tag)
    # the device check itself (no coordinate shortcut): indefinite -> info > 0, plain -> 0; repeatable
    P = _verify_predictor(g, "indef")
    pred, err = P.predict_arrays(0, g["indef_pc"])
    assert np.all(err > 0)                       # positive diagonal ...
    h = P._factored_handle()
    assert h.verify_model() > 0 and h.verify_model() > 0   # ... but not positive definite
    P = _verify_predictor(g, "A")
    P.predict_arrays(0, g["plain_pc"])
    assert P._factored_handle().verify_model() == 0
    # more prediction sites than one 512-column panel, both site orders, against the oracle's Schur complement
    rng = np.random.default_rng(77)
    po = orc.Params.from_flat(g["A_params"])
    c = [g["A_c0"], g["A_c1"]]
    pc = np.column_stack([rng.uniform(26, 49, 700), rng.uniform(-120, -72, 700)])
    S = orc.joint_cov(po, c, HAV)
    c0 = orc.pred_cross_cov(po, c, pc, 1, HAV)
    sch = orc.pred_cov(po, pc, 1, HAV) - c0.T @ np.linalg.solve(S, c0)
    assert np.linalg.eigvalsh(sch).min() > 1e-6
    P.predict_arrays(1, pc)
    assert P._factored_handle().verify_model() == 0
    P.verify_model = False
    P.predict_arrays(1, pc)
    assert P._verdict is None


class _StubTrend:
    """stands in for the sklearn LinearRegression the reference stores in ds.attrs["spatial_model"] (src/fields.py:345-375)"""

    def __init__(self, coef, intercept):
        self.coef, self.intercept = np.asarray(coef, dtype=float), float(intercept)

    def predict(self, X):
        return np.asarray(X, dtype=float) @ self.coef + self.intercept


class _Attrs:
    def __init__(self, attrs):
        self.attrs = attrs


def test_postprocess_and_stale_state(native):
    """postprocess=True (src/joint_prediction.py:155-205): rescale by scale_fact, add spatial_mean, the OLS spatial trend
    of the standardised [lon, lat] covariates and the temporal trend -- against the same arithmetic done by hand on
    the postprocess=False output.  Then: parameters changed after the first call are NOT ignored (the reference
    re-reads mod.params on every call)."""
    import pandas as pd
    from sif_xco2_cokriging_amd import fields, joint_prediction, model
    g = load_golden("joint_solve")
    mod = model.MultivariateMatern(params=model.MaternParams().set_values(g["params_A"]))
    f0, f1 = fields.Field(g["coords0_A"], g["values0_A"]), fields.Field(g["coords1_A"], g["values1_A"])
    at = dict(scale_fact=1.7, spatial_mean=0.25, temporal_trend=-0.4, covariate_means=[-95.0, 37.0],
              covariate_scales=[12.0, 6.0], spatial_model=_StubTrend([0.3, -0.2], 0.05))
    for f in (f0, f1):
        f.ds = _Attrs(at)
        f.timestamp = "2020-07-01"
    mf = fields.MultiField([f0, f1])
    P = joint_prediction.Predictor(mod, mf)
    pc = pd.DataFrame(g["pcoords_A"][6:], columns=["lat", "lon"])     # rows 0..5 sit on data sites
    raw = P(1, pc, postprocess=False)
    raw = raw.to_dataframe().reset_index() if hasattr(raw, "to_dataframe") else raw.reset_index()
    raw = pc.merge(raw, on=["lat", "lon"], how="left")
    out = P(1, pc, postprocess=True)
    out = out.to_dataframe().reset_index() if hasattr(out, "to_dataframe") else out.reset_index()
    out = pc.merge(out, on=["lat", "lon"], how="left")
    trend = at["spatial_model"].predict(np.column_stack([(pc["lon"] - at["covariate_means"][0]) / at["covariate_scales"][0],
                                                         (pc["lat"] - at["covariate_means"][1]) / at["covariate_scales"][1]]))
    np.testing.assert_allclose(out["pred"].values, raw["pred"].values * 1.7 + 0.25 + trend - 0.4, rtol=1e-13)
    np.testing.assert_allclose(out["pred_err"].values, raw["pred_err"].values * 1.7, rtol=1e-13)
    assert rel(raw["pred"].values, g["pred_A_1"][6:]) < 1e-9
    # stale state: new parameters -> new factor, new numbers (== a fresh Predictor's)
    mod.params.set_values(g["params_R"])
    again = P.predict_arrays(1, pc.values)
    fresh = joint_prediction.Predictor(mod, mf).predict_arrays(1, pc.values)
    assert np.array_equal(again[0], fresh[0]) and np.array_equal(again[1], fresh[1])
    assert rel(again[0], raw["pred"].values) > 1e-6
    # correlation does not depend on rho: rho_12 = 0 must not divide by zero (src/model.py:188-191)
    vals = np.array(g["params_A"], dtype=float)
    vals[-1] = 0.0
    m0 = model.MultivariateMatern(params=model.MaternParams().set_values(vals))
    hlag = np.array([0.0, 10.0, 300.0])
    po = orc.Params.from_flat(g["params_A"])
    np.testing.assert_allclose(m0.correlation(0, 1, hlag), orc.matern_correlation(po.nu[0, 1], po.len_scale[0, 1], hlag), rtol=5e-13)
    np.testing.assert_allclose(m0.cross_covariance(0, 1, hlag), 0.0, atol=0)


@pytest.mark.parametrize("tag", ["A", "R", "B"])
def test_factor_predict_overlapped_matches_the_fixture_and_the_sequence(native, tag):
    """ck_factor_predict (factorisation and substitution as two overlapped sweeps): the reference's fixture at 1e-9, the
    same bits as ck_factor + ck_predict, and the factor stays resident for further predictions."""
    g = load_golden("joint_solve")
    coords = [g[f"coords0_{tag}"], g[f"coords1_{tag}"]]
    values = [g[f"values0_{tag}"], g[f"values1_{tag}"]]
    h, p = _assembled(native, g[f"params_{tag}"], coords, values, HAV)
    info, pred, err = h.factor_predict(0, g[f"pcoords_{tag}"])
    assert info == 0
    assert rel(pred, g[f"pred_{tag}_0"]) < 1e-9
    assert np.max(np.abs(err ** 2 - g[f"pred_err_{tag}_0"] ** 2)) < 1e-10
    pred1, err1 = h.predict(1, g[f"pcoords_{tag}"])          # on the resident factor
    assert rel(pred1, g[f"pred_{tag}_1"]) < 1e-9
    h2, _ = _assembled(native, g[f"params_{tag}"], coords, values, HAV)
    assert h2.factor() == 0
    pred2, err2 = h2.predict(0, g[f"pcoords_{tag}"])
    assert np.array_equal(pred, pred2) and np.array_equal(err, err2)
    with pytest.raises(native.NativeError):
        h.factor_predict(0, g[f"pcoords_{tag}"])              # already factored


@pytest.mark.parametrize("group", [0, 1, 2, 3, 4, 5, 11, 16])
@pytest.mark.parametrize("m", [700, 0, 1, 255, 256])
def test_tall_sweep_many_panels_every_group_size(native, group, m):
    """ck_factor_predict's default schedule (round 4): ONE sweep over the tall matrix [Sigma; c0^T; z^T] -- the right-hand-side
    rows as further workgroups of the cooperative panel step and further tiles of every update launch -- at 11 panels
    (N = 5 200), for group sizes that do and do not divide them (the look-ahead's A / B1 / B2 launches with one, two, three
    groups left; one group for everything), and for right-hand-side blocks of one and several tile rows: the SAME BITS as
    ck_factor + ck_predict with that grouping, the factor resident afterwards."""
    if m != 700 and group not in (0, 3):
        pytest.skip("the small right-hand-side blocks run with the default and one forced grouping")
    rng = np.random.default_rng(11)
    n = 2600
    coords = [np.column_stack([rng.uniform(25, 49, n), rng.uniform(-124, -67, n)]) for _ in range(2)]
    values = [rng.standard_normal(n), rng.standard_normal(n)]
    params = load_golden("joint_solve")["params_A"]
    pc = np.column_stack([rng.uniform(25, 49, m), rng.uniform(-124, -67, m)])
    h, p = _assembled(native, params, coords, values, HAV)
    if group:
        h.set_option("panel_group", group)      # (any grouping gives the same bits; the sequence takes the same one anyway)
    assert h.factor() == 0
    ref = h.predict(1, pc)
    h2, _ = _assembled(native, params, coords, values, HAV)
    h2.set_option("fused_group", group)
    h2.set_option("time_gemm", 1)
    info, pred, err = h2.factor_predict(1, pc)
    assert info == 0
    assert np.array_equal(pred, ref[0]) and np.array_equal(err, ref[1])
    t = h2.timings()
    assert t["fused_sweeps_ms"] > 0 and t["panel_coop_redone"] == 0
    G = group if group else 1                   # automatic: 1 below 14 panels, 2 below 40, 4 from there
    ng = -(-11 // G)
    launches = sum(min(G, 11 - g * G) - 1 for g in range(ng)) + sum(1 for g in range(ng) for d in (1, 2, 3) if g + d < ng)
    assert t["syrk_launches"] == launches, (t["syrk_launches"], launches)
    again = h2.predict(0, pc)                   # on the factor the tall sweep left resident
    assert np.array_equal(again[0], h.predict(0, pc)[0])
    if group != 1:
        # ck_predict's sweep with the chain of the next group under the bulk of the current one (option solve_la; automatic
        # from 40 panels) against the launches one after the other: same bits, for both processes
        h2.set_option("panel_group", group if group else 3)     # (11 panels: the automatic group size is one -- no look-ahead)
        for la in (1, 0):
            h2.set_option("solve_la", la)
            for i in (1, 0):
                a, b = h2.predict(i, pc), h.predict(i, pc)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (la, i)
        h2.set_option("solve_la", -1)
        h2.set_option("panel_group", 0)
    h3, _ = _assembled(native, params, coords, values, HAV)
    h3.set_option("tall_sweep", 0)              # round 3's two overlapped sweeps: still the same bits
    h3.set_option("fused_group", group)
    info, p3, e3 = h3.factor_predict(1, pc)
    assert info == 0 and np.array_equal(p3, ref[0]) and np.array_equal(e3, ref[1])
    # the split panel step (option tall_split; automatic from 12 288 rows behind the first group): the cooperative launch on the
    # 512 x 512 head only, every other row -- Sigma's and the right-hand sides' -- through k_panel_rows_all in one launch
    # (and B2(g) behind B1(g) on one stream instead of beside it on the handle's own: option tall_b2_stream)
    for split, rows in ((1, 0), (2, 1024), (0, 0)):
        h4, _ = _assembled(native, params, coords, values, HAV)
        h4.set_option("fused_group", group)
        h4.set_option("tall_split", split)
        h4.set_option("tall_split_rows", rows)
        h4.set_option("tall_b2_stream", 0 if split == 1 else 1)
        info, p4, e4 = h4.factor_predict(1, pc)
        assert info == 0 and np.array_equal(p4, ref[0]) and np.array_equal(e4, ref[1]), (split, rows)
        assert h4.timings()["panel_coop_redone"] == 0


@pytest.mark.parametrize("m", [100, 982])
def test_right_hand_side_assembly_transforms_its_own_sites_and_reads_nothing_stale(native, m):
    """K2 is ONE launch since round 4: every strip transforms its 64 prediction sites itself and the strips of block column 0
    store them for the later users.  No strip may read those stored vectors inside the launch (the edge sub-tiles at the
    process boundary and at the end of the matrix once did: a race that showed as results depending on the PREVIOUS call's
    sites): a handle that has just predicted at other sites gives the bits of a fresh handle, repeatedly."""
    from sif_xco2_cokriging_amd import synth
    pb = synth.conus_problem(700, params=synth.SET_A, seed=5)       # n0 = 700: not a multiple of 64 -> edge sub-tiles
    rng = np.random.default_rng(m)
    pcs = [np.column_stack([rng.uniform(25, 49, m), rng.uniform(-124, -67, m)]) for _ in range(3)]
    fresh = []
    for pc in pcs:
        h, _ = _assembled(native, pb["params"], pb["coords"], pb["values"], HAV)
        assert h.factor() == 0
        fresh.append(h.predict(0, pc))
        h.close()
    h, _ = _assembled(native, pb["params"], pb["coords"], pb["values"], HAV)
    assert h.factor() == 0
    fb = []
    for rep in range(2):
        for k in (0, 1, 2, 1):
            h.table_fallbacks()
            p, e = h.predict(0, pcs[k])
            fb.append((k, h.table_fallbacks()))
            assert np.array_equal(p, fresh[k][0]) and np.array_equal(e, fresh[k][1]), (rep, k)
    assert len({c for k, c in fb if k == 1}) == 1           # the exact pass sees the same pairs every time
    h.assemble_joint()                                       # ... and through the product call
    info, p, e = h.factor_predict(0, pcs[2])
    assert info == 0 and np.array_equal(p, fresh[2][0]) and np.array_equal(e, fresh[2][1])


@pytest.mark.parametrize("entry", ["factor", "factor_predict", "factor_predict_split", "factor_predict_two_sweeps"])
def test_cooperative_panel_step_timeout_is_detected_and_the_factorisation_redone(native, entry):
    """k_panel_coop's safety net: option coop_inject_panel makes one workgroup of the diagonal block
    skip its flag store, so the bounded waits of the chunks below it trip, the error word is set, and the host must notice,
    switch the cooperative step off and repeat the factorisation with one launch per dependency -- for ck_factor and for
    both forms of ck_factor_predict -- with results equal to the plain schedule's and the event visible in ck_timings."""
    rng = np.random.default_rng(21)
    n = 1500
    coords = [np.column_stack([rng.uniform(25, 49, n), rng.uniform(-124, -67, n)]) for _ in range(2)]
    values = [rng.standard_normal(n), rng.standard_normal(n)]
    params = load_golden("joint_solve")["params_A"]
    pc = np.column_stack([rng.uniform(25, 49, 300), rng.uniform(-124, -67, 300)])
    href, _ = _assembled(native, params, coords, values, HAV)
    href.set_option("panel_fused", 2)           # never cooperative
    assert href.factor() == 0
    ref = href.predict(0, pc)
    h, _ = _assembled(native, params, coords, values, HAV)
    h.set_option("coop_spins", 20000)           # ~20 ms instead of ~2 s per timed-out wait
    h.set_option("coop_inject_panel", 2)        # panel 2 of 6
    if entry == "factor":
        assert h.factor() == 0
        assert h.timings()["panel_coop_redone"] == 1
        got = h.predict(0, pc)
    else:
        if entry.endswith("two_sweeps"):
            h.set_option("tall_sweep", 0)
        if entry.endswith("split"):                 # the cooperative launch on the head only (large panels take this form)
            h.set_option("tall_split", 1)
        info, *got = h.factor_predict(0, pc)
        assert info == 0
        assert h.timings()["panel_coop_redone"] == 1
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    # the handle keeps working, without the cooperative step
    h.assemble_joint()
    info, p2, e2 = h.factor_predict(0, pc)
    assert info == 0 and np.array_equal(p2, ref[0]) and h.timings()["panel_coop_redone"] == 0


def test_step_wise_driver_sweeps_again_after_a_cooperative_timeout(native):
    """The same event in the step-wise form every rank of a multi-GPU run executes (distributed.DistributedJoint, world = 1):
    ck_factor_info reports it, the driver encodes it in the gathered status and repeats the pass."""
    import torch
    from sif_xco2_cokriging_amd.distributed import DistributedJoint
    rng = np.random.default_rng(22)
    n = 1300
    coords = [np.column_stack([rng.uniform(25, 49, n), rng.uniform(-124, -67, n)]) for _ in range(2)]
    values = [rng.standard_normal(n), rng.standard_normal(n)]
    params = load_golden("joint_solve")["params_A"]
    pc = np.column_stack([rng.uniform(25, 49, 200), rng.uniform(-124, -67, 200)])
    href, _ = _assembled(native, params, coords, values, HAV)
    assert href.factor() == 0
    ref = href.predict(1, pc)
    pv = params
    h = native.Handle(0)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(HAV)
    for k in range(2):
        h.set_data(k, coords[k], values[k])
    h.set_option("coop_spins", 20000)
    h.set_option("coop_inject_panel", 1)
    r = DistributedJoint(h, 0, 1, device=torch.device("cuda", 0)).prepare(len(pc))
    pred, err = r.predict(1, pc)
    assert r._coop_resweeps == 1
    assert rel(pred, ref[0]) < 1e-11 and rel(err, ref[1]) < 1e-11


@pytest.mark.parametrize("la", [0, 1])
@pytest.mark.parametrize("prio,group", [(0, 0), (1, 0), (2, 0), (0, 1), (0, 2), (0, 4)])
def test_factor_predict_many_panels_every_schedule(native, prio, group, la):
    """Round 3's form (option tall_sweep = 0): factorisation and substitution as two overlapped sweeps.
    Enough panels for groups, look-ahead boundaries and the last, shorter group (N = 5 200: 11 panels); every stream
    assignment and group size gives the sequence's result."""
    rng = np.random.default_rng(11)
    n = 2600
    coords = [np.column_stack([rng.uniform(25, 49, n), rng.uniform(-124, -67, n)]) for _ in range(2)]
    values = [rng.standard_normal(n), rng.standard_normal(n)]
    params = load_golden("joint_solve")["params_A"]
    pc = np.column_stack([rng.uniform(25, 49, 700), rng.uniform(-124, -67, 700)])
    h, p = _assembled(native, params, coords, values, HAV)
    assert h.factor() == 0
    ref = h.predict(1, pc)
    h2, _ = _assembled(native, params, coords, values, HAV)
    h2.set_option("tall_sweep", 0)
    h2.set_option("fused_prio", prio)
    h2.set_option("fused_group", group)
    h2.set_option("fused_la", la)     # three streams: chain + next group's update / bulk / substitution
    info, pred, err = h2.factor_predict(1, pc)
    assert info == 0
    # ANY grouping gives the same bits: an element's updates are accumulated k ascending inside a launch and the tile is stored and
    # re-read exactly between launches
    assert np.array_equal(pred, ref[0]) and np.array_equal(err, ref[1])
    assert h2.timings()["fused_sweeps_ms"] > 0


@pytest.mark.parametrize("site_order", [0, 1])
def test_factor_predict_not_positive_definite_reports_minor(native, site_order):
    g = load_golden("joint_not_pd")
    h, p = _assembled(native, g["params"], [g["coords0"], g["coords1"]], [np.zeros(260), np.zeros(260)], HAV,
                      site_order=site_order)
    info, pred, err = h.factor_predict(0, np.array([[35.0, -100.0], [36.0, -101.0]]))
    assert info == int(g["minor"])


@pytest.mark.parametrize("m", [100, 511, 512, 700, 1300])
def test_verify_model_across_panel_boundaries(native, m):
    """ck_verify_model's Schur complement (exact launch grid: one workgroup per lower tile in front of the padding) for
    numbers of prediction sites around the 512-column panel and 256-row alignment boundaries: positive definite for
    distinct sites of a valid model, as numpy finds the stacked matrix of the oracle."""
    rng = np.random.default_rng(m)
    n = 300
    coords = [np.column_stack([rng.uniform(30, 45, n), rng.uniform(-115, -80, n)]) for _ in range(2)]
    values = [rng.standard_normal(n), rng.standard_normal(n)]
    params = load_golden("joint_solve")["params_A"]
    pc = np.column_stack([rng.uniform(30, 45, m), rng.uniform(-115, -80, m)])
    h, p = _assembled(native, params, coords, values, HAV)
    assert h.factor() == 0
    pred, err = h.predict(0, pc)
    assert h.verify_model() == 0
    if m <= 700:   # the reference's own check (src/joint_prediction.py:260-274) on the oracle's matrices
        S = orc.joint_cov(p, coords, HAV)
        c0 = orc.pred_cross_cov(p, coords, pc, 0, HAV)
        cpp = orc.pred_cov(p, pc, 0, HAV)
        np.linalg.cholesky(np.block([[cpp, c0.T], [c0, S]]))


def test_factor_predict_zero_and_one_point_and_univariate(native):
    """Edge shapes of the overlapped form: no prediction site, one site, a univariate model, a single tiny panel."""
    g = load_golden("joint_loocv")
    coords, values = [g["coords0"], g["coords1"]], [g["values0"], g["values1"]]
    h, p = _assembled(native, g["params"], coords, values, HAV)
    info, pred, err = h.factor_predict(0, np.zeros((0, 2)))
    assert info == 0 and pred.shape == (0,) and err.shape == (0,)
    one = np.array([[38.0, -95.0]])
    p1, e1 = h.predict(1, one)                               # on the factor the empty call left resident
    h2, _ = _assembled(native, g["params"], coords, values, HAV)
    info, p2, e2 = h2.factor_predict(1, one)
    assert info == 0 and np.array_equal(p1, p2) and np.array_equal(e1, e2)
    h3, _ = _assembled(native, g["params"], coords, values, HAV)
    h3.set_option("fused_sweeps", 0)                         # the call's sequential form (automatic beyond 128 panels)
    info, p3, e3 = h3.factor_predict(1, one)
    assert info == 0 and np.array_equal(p1, p3) and np.array_equal(e1, e3) and h3.timings()["fused_sweeps_ms"] == 0
    rp, re = orc.joint_predict(p, coords, values, one, 1, HAV)
    assert rel(p2, rp) < 1e-9 and abs(e2[0] - re[0]) < 1e-9
    k = load_golden("kat_simulation_experiment")
    hu, pu = _assembled(native, k["params_uni"], [k["coords1"]], [k["values1"]], EUC)
    info, pred, err = hu.factor_predict(0, k["pcoords"])
    assert info == 0 and rel(pred, k["pred_uni"]) < 1e-6 and np.max(np.abs(err - k["pred_err_uni"])) < 1e-6
    ht, pt = _assembled(native, g["params"], [coords[0][:3], coords[1][:2]], [values[0][:3], values[1][:2]], HAV)
    info, pred, err = ht.factor_predict(0, one)
    rp, re = orc.joint_predict(pt, [coords[0][:3], coords[1][:2]], [values[0][:3], values[1][:2]], one, 0, HAV)
    assert info == 0 and rel(pred, rp) < 1e-9 and abs(err[0] - re[0]) < 1e-9
