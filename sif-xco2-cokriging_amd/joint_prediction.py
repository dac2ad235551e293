"""Global ("joint") cokriging -- same call signatures as the reference's
``joint_prediction`` module (src/joint_prediction.py), numeric core in HIP.

    from sif_xco2_cokriging_amd import joint_prediction as prediction
    cokrig = prediction.Predictor(mod, mf, fast_dist=False, dist_units=None)
    ds = cokrig(1, pcoords, postprocess=False)

``mod`` / ``mf`` may be this package's ``model.MultivariateMatern`` / ``fields.MultiField``
or the reference's own objects: only ``mod.n_procs``, ``mod.params.<p>.values``,
``mf.n_procs`` and ``mf.fields[k].coords_main / values_main / timestamp / ds.attrs``
are read.

What runs on the GPU (include/cokrige.h): assembly of Sigma and c0, blocked FP64-MFMA
Cholesky, forward substitution fused with the prediction / variance reductions.  The
factor is kept on the device, so further calls with new ``pcoords`` or another ``i`` cost
one substitution sweep each.

Deviations from the reference, all outside the arithmetic:
  * the reference's ``_verify_model`` factorises the (m+N)x(m+N) stacked matrix to decide
    whether to warn (src/joint_prediction.py:60-66,260-274).  Sigma is positive definite at
    that point, so the stacked matrix is positive definite iff the m x m Schur complement
    C_pp - c0^T Sigma^-1 c0 is: ``ck_verify_model`` factorises THAT, from the solved
    right-hand sides the prediction left on the device (see ``_verify``).  Where the stacked
    matrix is exactly singular (duplicate prediction sites; a site on a datum of the predicted
    process) the reference's own outcome hangs on rounding; here those cases always warn.
  * without xarray installed the result is a pandas DataFrame indexed by the coordinate
    columns instead of an ``xarray.Dataset`` (same columns ``pred``, ``pred_err``).
"""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd
from numpy.linalg import LinAlgError

from . import native
from .fields import metric_of
from .model import configure_handle

try:  # the reference returns xarray objects; keep that when xarray exists
    import xarray as xr
except Exception:  # pragma: no cover - absent in the build image
    xr = None


class Predictor:
    """Multivariate prediction framework (src/joint_prediction.py:13-33)."""

    def __init__(self, mod, mf, covariates=None, dist_units: str = "km", fast_dist: bool = True,
                 device: int = 0, devices=None) -> None:
        """``devices=[0, 1, ...]``: the multi-GPU form (BASELINE configs[3]) -- one worker process per entry, Sigma
        block-column-cyclic over them, panels exchanged over RCCL/xGMI at each Cholesky step, prediction points sharded
        (workers.RankPool + distributed.DistributedJoint; the same ordinal twice rehearses it on one GPU over gloo).
        The reference's parallel entry is likewise a keyword on the predictor (src/point_prediction.py:45-52,69-81).
        Same results as ``device=`` to rounding; ``cross_validation`` and the exact ``_verify_model`` check are
        single-device paths and run on ``devices[0]``."""
        if mod.n_procs != mf.n_procs:
            raise ValueError("Number of theoretical processes different from empirical processes.")
        self.n_procs = mod.n_procs
        self.mod = mod
        self.mf = mf
        self.covariates = covariates
        self.dist_units = dist_units
        self.fast_dist = fast_dist
        self.devices = None if devices is None else [int(d) for d in devices]
        self.device = device if self.devices is None else self.devices[0]
        self._pool, self._pool_key = None, None
        self.comm = {}
        self.timings = {}
        self.rhs_budget_bytes = 48 << 30   # device memory for the right-hand sides of one ck_predict call
        # _verify_model: None = exact check (Cholesky of the m x m Schur complement on the device) for up to
        # `verify_max_points` prediction sites and the variance test beyond; True / False force it on / off
        self.verify_model = None
        self.verify_max_points = 16384
        self._h = None
        self._key = None
        self._verdict = None

    # -- device state -------------------------------------------------------------------------
    def _new_handle(self, drop=None):
        """Handle with model, metric and data loaded; ``drop=(i, ix)`` withholds datum ix of
        process i (LOOCV, src/joint_prediction.py:56-58,112-113,140-146)."""
        h = native.Handle(self.device)
        configure_handle(h, self.mod)
        h.set_metric(metric_of(self.dist_units, self.fast_dist))
        for k in range(self.n_procs):
            c = np.asarray(self.mf.fields[k].coords_main, dtype=np.float64)
            v = np.asarray(self.mf.fields[k].values_main, dtype=np.float64)
            if drop is not None and drop[0] == k:
                c = np.delete(c, drop[1], axis=0)
                v = np.delete(v, drop[1], axis=0)
            h.set_data(k, c, v)
        return h

    @staticmethod
    def _factor(h):
        h.assemble_joint()
        info = h.factor()
        if info != 0:
            # scipy.linalg.cho_factor's message (raised uncaught at src/joint_prediction.py:69)
            raise LinAlgError(f"{info}-th leading minor of the array is not positive definite")

    @staticmethod
    def _factor_predict(h, i, pc):
        """cho_factor + the solve of src/joint_prediction.py:67-78 in one library call: the factorisation and the forward
        substitution run as two overlapped sweeps (include/cokrige.h: ck_factor_predict); the factor stays resident."""
        h.assemble_joint()
        info, pred, err = h.factor_predict(i, pc)
        if info != 0:
            raise LinAlgError(f"{info}-th leading minor of the array is not positive definite")
        return pred, err

    def _state_key(self):
        """What the resident factor depends on: the model's parameters, the metric, and the data arrays.  The
        reference re-reads `mod.params` and `mf` on every __call__ (src/joint_prediction.py:50-55); here the factor
        is reused only while none of them has changed (mod.fit(...), params.set_values(...), new fields -> refactor)."""
        from .model import model_arrays
        n, sig, nu, ls, nug, rho = model_arrays(self.mod)
        key = [n, sig.tobytes(), nu.tobytes(), ls.tobytes(), nug.tobytes(), float(rho),
               metric_of(self.dist_units, self.fast_dist)]
        for k in range(self.n_procs):
            f = self.mf.fields[k]
            c = np.ascontiguousarray(f.coords_main, dtype=np.float64)
            v = np.ascontiguousarray(f.values_main, dtype=np.float64)
            key += [c.shape, hash(c.tobytes()), hash(v.tobytes())]
        return tuple(key)

    def invalidate(self):
        """Drop the resident factor (it is rebuilt on the next call)."""
        if self._h is not None:
            self._h.close()
        self._h, self._key = None, None
        self._pool_key = None

    def close(self):
        """Release the device state; with ``devices=`` also end the worker processes."""
        self.invalidate()
        if self._pool is not None:
            self._pool.close()
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _predict_on_ranks(self, i, pc):
        """The multi-GPU form: ship model and data to the ranks when they have changed, predict on the resident factor
        otherwise."""
        from . import workers
        from .model import model_arrays
        key = self._state_key()
        if self._pool is None:
            self._pool = workers.RankPool(self.devices)
        reuse = key == self._pool_key
        if not reuse:
            self._pool.load(model_arrays(self.mod), metric_of(self.dist_units, self.fast_dist),
                            [np.asarray(self.mf.fields[k].coords_main, dtype=np.float64) for k in range(self.n_procs)],
                            [np.asarray(self.mf.fields[k].values_main, dtype=np.float64) for k in range(self.n_procs)])
            self._pool_key = None
        pred, err = self._pool.predict_joint(i, pc, reuse_factor=reuse)
        self._pool_key = key
        self.timings, self.comm = self._pool.last_timings, self._pool.last_comm
        return pred, err

    def _factored_handle(self):
        key = self._state_key()
        if self._h is not None and key != self._key:
            self.invalidate()
        if self._h is None:
            h = self._new_handle()
            self._factor(h)
            self._h, self._key = h, key
        return self._h

    def predict_arrays(self, i: int, pcoords, cv_ix: int = None):
        """(pred, pred_err) as arrays -- the numeric body of ``__call__``
        (src/joint_prediction.py:49-78)."""
        pc = np.ascontiguousarray(np.atleast_2d(np.asarray(pcoords, dtype=np.float64)))
        if cv_ix is None and self.devices is not None and len(self.devices) > 1:
            self._verdict = None     # the exact _verify_model check is a single-device path: the variance test stands in
            return self._predict_on_ranks(i, pc)
        if cv_ix is None:
            key = self._state_key()
            if self._h is not None and key != self._key:
                self.invalidate()
            fresh = self._h is None
            h = self._new_handle() if fresh else self._h
            # The right-hand sides take (m + 1) x N doubles on the device: very large grids go through the
            # resident factor in batches (one forward sweep each), sized by `rhs_budget_bytes`.
            n_pad = h.num_panels()[2]
            chunk = max(1024, int(self.rhs_budget_bytes // (8 * max(n_pad, 1))))
            self._verdict = None
            try:
                if fresh and len(pc) > chunk:
                    self._factor(h)
                if len(pc) <= chunk:
                    # first call on this model and data: factorisation and substitution overlapped
                    pred, err = self._factor_predict(h, i, pc) if fresh else h.predict(i, pc)
            except Exception:
                if fresh:
                    h.close()
                raise
            self._h, self._key = h, key
            if len(pc) <= chunk:
                self._verdict = self._verify(h, i, pc, err)
            else:
                parts = [h.predict(i, pc[a:a + chunk]) for a in range(0, len(pc), chunk)]
                pred = np.concatenate([p for p, _ in parts])
                err = np.concatenate([e for _, e in parts])
            self.timings = h.timings()
        else:
            h = self._new_handle(drop=(i, cv_ix))
            try:
                pred, err = self._factor_predict(h, i, pc)
            finally:
                h.close()
        return pred, err

    def _verify(self, h, i, pc, pred_err):
        """True: the joint covariance of the data and these prediction sites is NOT positive definite
        (the reference's _verify_model raises LinAlgError, src/joint_prediction.py:260-274); False: it is;
        None: not checked exactly (switched off, or more than `verify_max_points` sites)."""
        want = self.verify_model
        if want is False or (want is None and len(pc) > self.verify_max_points):
            return None
        # exactly singular stacked matrices: two identical rows of pcoords, or a prediction site on a datum of
        # process i (h == 0 puts the nugget into c0 as well, src/model.py:195-196) -- decided on the coordinates
        rows = np.ascontiguousarray(pc).view([("a", np.float64), ("b", np.float64)]).ravel()
        if len(np.unique(rows)) < len(rows):
            return True
        data = np.ascontiguousarray(np.asarray(self.mf.fields[i].coords_main, dtype=np.float64)[:, :2])
        if np.isin(rows, data.view([("a", np.float64), ("b", np.float64)]).ravel()).any():
            return True
        return h.verify_model() != 0

    def _warn_if_invalid(self, pred_err):
        bad = self._verdict
        if bad is None:
            # not checked exactly: the necessary condition the variances give (a non-positive Schur diagonal)
            bad = bool(np.any(pred_err <= 0.0))
        if bad:
            warnings.warn("Prediction joint covariance matrix is not positive definte; model"
                          " technically invalid.")

    # -- reference call signature ----------------------------------------------------------------
    def __call__(self, i: int, pcoords: pd.DataFrame, postprocess: bool = True, cv_ix: int = None):
        """Prediction and standard error of process ``i`` at ``pcoords`` (format [[lat, lon]])
        (src/joint_prediction.py:35-92)."""
        self.i = i
        if cv_ix is not None:
            p = np.asarray(pcoords, dtype=np.float64).ravel()
            pcoords = pd.DataFrame({"d1": p[0], "d2": p[1]}, index=[0])
        elif not isinstance(pcoords, pd.DataFrame):
            a = np.atleast_2d(np.asarray(pcoords, dtype=np.float64))
            pcoords = pd.DataFrame({"d1": a[:, 0], "d2": a[:, 1]})
        pred, err = self.predict_arrays(i, pcoords.values[:, :2], cv_ix=cv_ix)
        if cv_ix is None:
            self._warn_if_invalid(err)
        df_pred = pcoords.copy()
        df_pred["pred"] = pred
        df_pred["pred_err"] = err
        if postprocess:
            df_pred = df_pred.rename(columns={"d1": "lat", "d2": "lon"})
            return self._postprocess_predictions(df_pred)
        out = df_pred.set_index(pcoords.columns.values.tolist())
        if xr is None:
            return out
        ds = out.to_xarray()
        ts = self.mf.fields[self.i].timestamp
        try:
            np.isnan(ts)
            return ds
        except TypeError:
            return ds.assign_coords(coords={"time": np.datetime64(ts)})

    def _postprocess_predictions(self, df: pd.DataFrame):
        """Back to the scale of the original data: undo the standardisation, add the OLS
        spatial trend and the temporal trend (src/joint_prediction.py:155-205).  O(m) host
        work on the attributes src/fields.py:345-375 stored."""
        at = self.mf.fields[self.i].ds.attrs
        out = df[["lon", "lat"]].copy()
        out["pred"] = df["pred"].values * at["scale_fact"] + at["spatial_mean"]
        out["pred_err"] = df["pred_err"].values * at["scale_fact"]
        if self.covariates is None:
            cov = df[["lon", "lat"]].copy()
            keep = np.ones(len(df), dtype=bool)
        else:
            if xr is None:
                raise RuntimeError("covariates are xarray objects in the reference; xarray is not installed")
            sel = self.covariates.sel(time=self.mf.fields[self.i].timestamp)
            vals = sel.to_dataframe(name="covariates").reset_index()
            merged = df[["lon", "lat"]].merge(vals[["lon", "lat", "covariates"]], on=["lon", "lat"], how="left")
            keep = merged["covariates"].notna().values
            cov = merged.loc[keep, ["covariates"]].copy()
        for k, name in enumerate(cov.columns):
            cov[name] = (cov[name] - at["covariate_means"][k]) / at["covariate_scales"][k]
        trend = np.full(len(df), np.nan)
        trend[keep] = at["spatial_model"].predict(cov)
        out["pred"] = out["pred"] + trend + at["temporal_trend"]
        out = out.set_index(["lon", "lat"])
        if xr is None:
            return out
        ds = out.to_xarray()
        return ds.assign_coords(coords={"time": np.datetime64(self.mf.fields[self.i].timestamp)})

    def cross_validation(self, i: int, postprocess: bool = True, refactor_each: bool = False) -> pd.DataFrame:
        """Leave-one-out cross-validation at each data location of process ``i``
        (src/joint_prediction.py:207-257).  The reference withholds one datum and re-assembles
        and re-factorises everything, n times; here all n leave-one-out predictions come from
        ONE factorisation (``ck_loocv``: the Gaussian conditional of z_q given the rest,
        pred_q = z_q - (Sigma^-1 z)_q / (Sigma^-1)_qq, var_q = 1 / (Sigma^-1)_qq -- the same
        numbers).  ``refactor_each=True`` runs the reference's n-solve loop instead."""
        names = ["lat", "lon"] if postprocess else ["d1", "d2"]
        f = self.mf.fields[i]
        data = pd.DataFrame(np.hstack((f.coords_main, np.atleast_2d(f.values_main).T)), columns=names + ["data"])
        if refactor_each:
            pred = np.empty(len(data))
            err = np.empty(len(data))
            for ix in range(len(data)):
                p, e = self.predict_arrays(i, f.coords_main[ix], cv_ix=ix)
                pred[ix], err[ix] = p[0], e[0]
        else:
            pred, err = self._factored_handle().loocv(i, len(data))
        if postprocess:
            at = f.ds.attrs
            tmp = pd.DataFrame({"lat": data["lat"], "lon": data["lon"], "pred": pred, "pred_err": err})
            self.i = i
            pp = self._postprocess_predictions(tmp)
            pp = pp.to_dataframe().reset_index() if xr is not None and not isinstance(pp, pd.DataFrame) else pp.reset_index()
            data = data.merge(pp.dropna(subset=["pred"]), on=names, how="outer")
        else:
            data["pred"], data["pred_err"] = pred, err
        data["residual"] = data["data"] - data["pred"]
        # the reference's xr.merge(...).to_dataframe() + outer merge hands the rows back sorted by the coordinates
        # (src/joint_prediction.py:248-254)
        data = data.sort_values(names, kind="stable").reset_index(drop=True)
        return data[names + ["data", "pred", "residual", "pred_err"]]


def prediction_coords(extents: tuple = (-125, -65, 22, 58), lon_res: float = 0.5, lat_res: float = 0.5,
                      land_only: bool = True) -> pd.DataFrame:
    """Prediction grid [lat, lon] (src/joint_prediction.py:277-283).  The reference keeps land
    cells only, through regionmask's Natural Earth polygons; where regionmask is not
    installed ask for the full rectangle with ``land_only=False``."""
    lon = np.arange(extents[0], extents[1] + 0.5 * lon_res, lon_res)
    lat = np.arange(extents[2], extents[3] + 0.5 * lat_res, lat_res)
    if land_only:
        try:
            import regionmask  # noqa: F401
        except Exception as e:
            raise RuntimeError("land masking needs regionmask (as in the reference); "
                               "use land_only=False for the full rectangle") from e
        land = regionmask.defined_regions.natural_earth_v5_0_0.land_110
        mask = land.mask(lon, lat)
        la, lo = np.meshgrid(lat, lon, indexing="ij")
        ok = ~np.isnan(np.asarray(mask))
        return pd.DataFrame({"lat": la[ok], "lon": lo[ok]})
    la, lo = np.meshgrid(lat, lon, indexing="ij")
    return pd.DataFrame({"lat": la.ravel(), "lon": lo.ravel()})
