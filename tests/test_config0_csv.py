"""BASELINE.json configs[0]: the reference's own CPU-runnable case -- a notebook flow on the gridded SIF-residual CSV
(l2_north_america/gridded_sif_residuals_north_america.csv; the file itself is missing from the reference,
.MISSING_LARGE_BLOBS, and research/cokriging_demo.ipynb calls modules that no longer exist).
What can be rehearsed is the plumbing of the CURRENT API on a table of that schema and size:
CSV -> standardised residual columns -> Field / MultiField -> MaternParams.set_values -> MultivariateMatern ->
Predictor -> __call__ (research/simulation_experiment.ipynb cells 3-11), sub-sampled to a CPU-feasible n.
CPU: the oracle runs the flow (numpy / scipy, no GPU).  GPU: the product classes on the same CSV, against it."""
import numpy as np
import pandas as pd
import pytest

from oracle import cokrige_oracle as orc

N_SUB = 350


def _tables(tmp_path):
    from sif_xco2_cokriging_amd import synth
    sif, xco2 = synth.residual_tables()
    assert len(sif) == 47562 and len(xco2) == 23080
    assert list(sif.columns) == ["lon", "lat", "evi", "sif", "lon_std", "lat_std", "evi_std", "ols_mean", "sif_residuals",
                                 "sif_residuals_std"]
    a, b = tmp_path / "gridded_sif_residuals_north_america.csv", tmp_path / "gridded_xco2_residuals_north_america.csv"
    sif.to_csv(a, index=False)
    xco2.to_csv(b, index=False)
    return pd.read_csv(a), pd.read_csv(b)


def _subsample(df, name, seed):
    d = df.sample(N_SUB, random_state=seed)
    return d[["lat", "lon"]].values, d[f"{name}_residuals_std"].values


def _grid():
    lat, lon = np.arange(30.0, 46.0, 1.0), np.arange(-110.0, -80.0, 1.5)
    la, lo = np.meshgrid(lat, lon, indexing="ij")
    return pd.DataFrame({"lat": la.ravel(), "lon": lo.ravel()})


def _oracle_flow(sif, xco2):
    from sif_xco2_cokriging_amd import synth
    c0, v0 = _subsample(sif, "sif", 1)
    c1, v1 = _subsample(xco2, "xco2", 2)
    p = orc.Params.from_flat(synth.SET_A)
    pc = _grid()
    pred, err = orc.joint_predict(p, [c0, c1], [v0, v1], pc.values, 0, orc.METRIC_HAVERSINE)
    return (c0, v0, c1, v1, pc), pred, err


def test_config0_csv_plumbing_cpu(tmp_path):
    sif, xco2 = _tables(tmp_path)
    (c0, v0, c1, v1, pc), pred, err = _oracle_flow(sif, xco2)
    assert pred.shape == (len(pc),) and np.all(np.isfinite(pred)) and np.all(err > 0)
    p = orc.Params.from_flat([0.99, 0.81, 0.39, 0.695, 1.0, 460.0, 460.0, 460.0, 0.02, 0.025, -0.19])
    assert np.all(err ** 2 <= p.sigma[0] ** 2 + p.nugget[0] + 1e-12)
    # the empirical variogram of the same table (the notebook's other cell), on the CPU path
    cen, edg, mean, cnt = orc.variogram(c0, v0, c0, v0, True, orc.METRIC_HAVERSINE, 1500.0, 30)
    assert cnt.sum() > 0 and len(cen) == 30


@pytest.mark.gpu
def test_config0_csv_plumbing_gpu(tmp_path):
    from sif_xco2_cokriging_amd import fields, joint_prediction, model, synth
    sif, xco2 = _tables(tmp_path)
    (c0, v0, c1, v1, pc), rp, re = _oracle_flow(sif, xco2)
    mod = model.MultivariateMatern(params=model.MaternParams().set_values(synth.SET_A))
    mf = fields.MultiField([fields.Field(c0, v0), fields.Field(c1, v1)])
    out = joint_prediction.Predictor(mod, mf)(0, pc, postprocess=False)
    df = out.to_dataframe().reset_index() if hasattr(out, "to_dataframe") else out.reset_index()
    df = pc.merge(df, on=["lat", "lon"], how="left")
    assert np.max(np.abs(df["pred"].values - rp)) / np.max(np.abs(rp)) < 1e-9
    assert np.max(np.abs(df["pred_err"].values - re)) / np.max(np.abs(re)) < 1e-9
    ev = mf.empirical_variograms(fields.VarioConfig(1500.0, 30))
    ref = orc.variogram(c0, v0, c0, v0, True, orc.METRIC_HAVERSINE, 1500.0, 30)
    d = ev.df.loc[(0, 0)]
    assert np.array_equal(d["bin_count"].values, ref[3])
    np.testing.assert_allclose(d["bin_mean"].values, ref[2], rtol=1e-11)
