"""Host-side mirror of the parts of the reference's ``fields`` module the hot path
touches (src/fields.py): the attribute bags the predictors read
(``Field.coords_main/values_main/coords/values/timestamp``, ``MultiField.fields``),
``distance_matrix`` and the empirical (cross-)variogram -- with the arithmetic in HIP.

The xarray/OLS pre-processing of src/fields.py:59-121,283-375 is O(n) data
preparation outside the hot path and is not mirrored: build a ``Field`` directly
from arrays (or pass the reference's own ``MultiField`` -- the predictors only duck-type).
"""
from __future__ import annotations

import warnings
from dataclasses import dataclass

import numpy as np
import pandas as pd

from . import native

EARTH_RADIUS = 6371  # km (src/fields.py:17)

_shared = {}


def _shared_handle(device=0):
    if device not in _shared:
        _shared[device] = native.Handle(device)
    return _shared[device]


def metric_of(dist_units, fast_dist) -> int:
    """The reference's branch selection (src/fields.py:332-342)."""
    if fast_dist:
        return native.METRIC_HAVERSINE
    if dist_units is None:
        return native.METRIC_EUCLID
    raise NotImplementedError(
        "geodesic distances (fast_dist=False with units) go through a per-pair geopy callback in the "
        "reference (src/fields.py:337-339) and are outside the HIP path")


def distance_matrix(X1, X2, units: str = "km", fast_dist: bool = False, device: int = 0) -> np.ndarray:
    """Pairwise distances, rows formatted [lat, lon] (src/fields.py:318-342)."""
    h = _shared_handle(device)
    h.set_metric(metric_of(units, fast_dist))
    return h.distance_dense(np.atleast_2d(X1), np.atleast_2d(X2))


class VarioConfig:
    """src/fields.py:20-46."""

    def __init__(self, max_dist, n_bins, n_procs=2, kind="Semivariogram", dist_units="km", fast_dist=True):
        self.max_dist, self.n_bins, self.n_procs, self.kind = max_dist, n_bins, n_procs, kind
        self.dist_units, self.fast_dist = dist_units, fast_dist
        self.covariogram = self.kind == "Covariogram"


@dataclass
class EmpiricalVariogram:
    """src/fields.py:49-56."""
    df: pd.DataFrame
    config: VarioConfig
    timestamp: object
    timedeltas: list


class Field:
    """Values and coordinates of one process at one time (the attributes of
    src/fields.py:59-95 that the predictors and variograms read)."""

    def __init__(self, coords, values, coords_main=None, values_main=None, timestamp=np.nan, attrs=None):
        self.coords = np.ascontiguousarray(coords, dtype=np.float64).reshape(-1, 2)
        self.values = np.ascontiguousarray(values, dtype=np.float64).ravel()
        self.coords_main = self.coords if coords_main is None else np.ascontiguousarray(coords_main, dtype=np.float64).reshape(-1, 2)
        self.values_main = self.values if values_main is None else np.ascontiguousarray(values_main, dtype=np.float64).ravel()
        self.timestamp = timestamp
        self.size = len(self.values)
        # post-processing attributes of src/fields.py:345-375 (scale_fact, spatial_mean, ...)
        self.ds = _Attrs(attrs or {})


class _Attrs:
    def __init__(self, attrs):
        self.attrs = dict(attrs)


class MultiField:
    """A multivariate process: list of Field (src/fields.py:124-190)."""

    def __init__(self, fields, timestamp=np.nan, timedeltas=None):
        self.fields = np.empty(len(fields), dtype=object)
        for k, f in enumerate(fields):
            self.fields[k] = f
        self.n_procs = len(fields)
        self.timestamp = timestamp
        self.timedeltas = timedeltas if timedeltas is not None else [np.nan] * self.n_procs
        self.n_data = int(sum(f.size for f in fields))

    def calc_dist_matrix(self, ids, units, fast_dist, main=False):
        assert len(ids) == 2
        cs = [self.fields[i].coords_main if main else self.fields[i].coords for i in ids]
        return distance_matrix(*cs, units=units, fast_dist=fast_dist)

    # empirical variograms are added by ``variogram.py`` (K5) ------------------------------
    def get_variogram(self, i, j, config):
        from .variogram import get_variogram
        return get_variogram(self, i, j, config)

    def empirical_variograms(self, config):
        """src/fields.py:234-252."""
        vs = [self.get_variogram(i, j, config) for i in range(self.n_procs) for j in range(self.n_procs) if i <= j]
        return EmpiricalVariogram(pd.concat(vs), config, self.timestamp, self.timedeltas)
