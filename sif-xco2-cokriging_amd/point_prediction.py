"""Local-neighbourhood cokriging -- same call signatures as the reference's ``point_prediction``
module (src/point_prediction.py), one GPU workgroup per prediction point.

    from sif_xco2_cokriging_amd import point_prediction as prediction
    P = prediction.Predictor(mod, mf)
    ds = P(0, pcoords, max_dist=1e3, postprocess=False)

Differences from the reference, none in the arithmetic:
  * no global ``Sigma`` blocks are precomputed or gathered (src/point_prediction.py:98-113,
    153-181): each workgroup assembles the covariance of its own neighbours;
  * ``partitions`` (a ``multiprocessing.Pool`` in the reference, :69-81) is accepted and ignored:
    the prediction points are already processed in parallel;
  * the reference warns once per affected point (:219-221, 230-232); here one warning per
    call and kind, carrying the number of points.
"""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd

from . import native
from .fields import metric_of
from .joint_prediction import Predictor as _JointPredictor
from .joint_prediction import prediction_coords, xr  # noqa: F401  (same helper, same signature)
from .model import configure_handle


class Predictor:
    """Multivariate prediction framework (src/point_prediction.py:21-43)."""

    def __init__(self, mod, mf, covariates=None, dist_units: str = "km", fast_dist: bool = True, device: int = 0,
                 devices=None, reserve_scratch=None):
        """``devices=[0, 1, ...]``: the prediction points are sharded over one worker process per GPU (observations
        replicated, no exchange inside the computation) -- what ``partitions`` is to the reference's CPU pool
        (src/point_prediction.py:45-52, 69-81)."""
        if mod.n_procs != mf.n_procs:
            raise ValueError("Number of theoretical processes different from empirical processes.")
        self.n_procs = mod.n_procs
        self.mod, self.mf, self.covariates = mod, mf, covariates
        self.dist_units, self.fast_dist = dist_units, fast_dist
        self.devices = None if devices is None else [int(d) for d in devices]
        self.device = device if self.devices is None else self.devices[0]
        self._pool, self._pool_key = None, None
        self.cv = False  # placeholder for cross-validation (src/point_prediction.py:43)
        self.info = {}
        self._h = None
        self._key = None
        # The reference builds its state -- the full Sigma blocks -- here, once (src/point_prediction.py:24-43).  Ours is the
        # scratch slab of the large-neighbourhood paths: reserve_scratch = bytes, or "auto" for the library's budget (a quarter
        # of the free device memory, at most 32 GiB), allocates it now so that no later call pays a hipMalloc of tens of GiB
        # (up to seconds: include/cokrige.h, ck_local_reserve); None (default): grown by the first call that needs it.
        self.reserve_scratch = reserve_scratch
        if reserve_scratch is not None and (self.devices is None or len(self.devices) <= 1):
            self._handle()

    def _handle(self):
        key = _JointPredictor._state_key(self)   # model parameters, metric, data: a change rebuilds the device state
        if self._h is not None and key != self._key:
            self._h.close()
            self._h = None
        self._key = key
        if self._h is None:
            h = native.Handle(self.device)
            configure_handle(h, self.mod)
            h.set_metric(metric_of(self.dist_units, self.fast_dist))
            for k in range(self.n_procs):
                h.set_data(k, self.mf.fields[k].coords_main, self.mf.fields[k].values_main)
            if self.reserve_scratch is not None:
                h.local_reserve(0 if self.reserve_scratch == "auto" else int(self.reserve_scratch))
            self._h = h
        return self._h

    def close(self):
        if self._h is not None:
            self._h.close()
            self._h = None
        if self._pool is not None:
            self._pool.close()
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _predict_on_ranks(self, i, pcoords, max_dist):
        from . import workers
        from .model import model_arrays
        key = _JointPredictor._state_key(self)
        if self._pool is None:
            self._pool = workers.RankPool(self.devices)
        if key != self._pool_key:
            self._pool.load(model_arrays(self.mod), metric_of(self.dist_units, self.fast_dist),
                            [np.asarray(self.mf.fields[k].coords_main, dtype=np.float64) for k in range(self.n_procs)],
                            [np.asarray(self.mf.fields[k].values_main, dtype=np.float64) for k in range(self.n_procs)])
            self._pool_key = key
        return self._pool.predict_local(i, pcoords, max_dist=max_dist, cv=self.cv)

    def predict_arrays(self, i: int, pcoords, max_dist: float = 1e3):
        if self.devices is not None and len(self.devices) > 1:
            pred, err, info = self._predict_on_ranks(i, pcoords, max_dist)
        else:
            pred, err, info = self._handle().predict_local(i, pcoords, max_dist=max_dist, cv=self.cv)
        self.info = info
        if info["n_empty"]:
            warnings.warn(f"No data within maximum distance {max_dist} at {info['n_empty']} location(s).")
        if info["n_not_pd"]:
            warnings.warn(f"Local covariance matrix not positive definte at {info['n_not_pd']} location(s);"
                          " returning NaN.")
        return pred, err

    def __call__(self, i: int, pcoords: pd.DataFrame, max_dist: float = 1e3, partitions: int = None,
                 postprocess: bool = True):
        """src/point_prediction.py:45-96."""
        self.i = i
        if not isinstance(pcoords, pd.DataFrame):
            a = np.atleast_2d(np.asarray(pcoords, dtype=np.float64))
            pcoords = pd.DataFrame({"d1": a[:, 0], "d2": a[:, 1]})
        pred, err = self.predict_arrays(i, pcoords.values[:, :2], max_dist=max_dist)
        df = pcoords.copy()
        df["pred"], df["pred_err"] = pred, err
        if postprocess:
            return _JointPredictor._postprocess_predictions(self, df)
        out = df.set_index(pcoords.columns.values.tolist())
        if xr is None:
            return out
        ds = out.to_xarray()
        ts = self.mf.fields[self.i].timestamp
        try:
            np.isnan(ts)
            return ds
        except TypeError:
            return ds.assign_coords(coords={"time": np.datetime64(ts)})

    def cross_validation(self, i: int, max_dist: float = 1e3, partitions: int = None,
                         postprocess: bool = True) -> pd.DataFrame:
        """Leave-one-out at every data location of process ``i`` (src/point_prediction.py:303-346):
        one local prediction per location with the co-located datum withheld."""
        self.cv = True
        names = ["lat", "lon"] if postprocess else ["d1", "d2"]
        f = self.mf.fields[i]
        data = pd.DataFrame(np.hstack((f.coords_main, np.atleast_2d(f.values_main).T)), columns=names + ["data"])
        out = self.__call__(i, data[names], max_dist=max_dist, partitions=partitions, postprocess=postprocess)
        df = out.to_dataframe().reset_index() if hasattr(out, "to_dataframe") else out.reset_index()
        df = df.dropna(subset=["pred"]).merge(data, on=names, how="outer")
        df["residual"] = df["data"] - df["pred"]
        return df[names + ["data", "pred", "residual", "pred_err"]]
