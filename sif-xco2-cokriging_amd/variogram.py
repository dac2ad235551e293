"""Empirical (cross-)semivariogram / covariogram with the pair loop on the GPU
(mirror of MultiField.get_variogram, src/fields.py:192-232; bins per src/fields.py:389-403).

Host side (this file): centring the values, the 31 data-dependent bin edges, the per-bin
division, the DataFrame the reference returns and its warnings.  Device side
(csrc/ck_vario.hip): every pair distance, the max_dist filter, bin membership, sums and counts.
"""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd

from . import native
from .fields import metric_of


def construct_bins(lo: float, hi: float, n_bins: int):
    """Bin centres and edges from the smallest positive and the largest retained distance:
    centres = linspace(lo, hi), edges half a width either side, first edge moved to zero
    (src/fields.py:389-403)."""
    centers = np.linspace(lo, hi, n_bins)
    width = centers[1] - centers[0]
    edges = np.arange(lo - 0.5 * width, hi + width, width)
    if len(edges) == n_bins + 1 and not np.allclose((edges[1:] + edges[:-1]) / 2, centers):
        warnings.warn("WARNING: variogram bins are not centered.")
    edges[0] = 0
    return centers, edges


def variogram_arrays(handle, coords_i, values_i, coords_j, values_j, same, max_dist, n_bins, covariogram=False):
    """(centers, edges, means, counts) for one pair of fields."""
    vi = np.asarray(values_i, dtype=np.float64)
    ri = vi - vi.mean()                       # src/fields.py:380
    if same:
        handle.vario_begin(coords_i, ri)
    else:
        vj = np.asarray(values_j, dtype=np.float64)
        handle.vario_begin(coords_i, ri, coords_j, vj - vj.mean())
    try:
        lo, hi, npos = handle.vario_extent(max_dist)
        if not npos:
            raise ValueError("no pair of distinct sites within max_dist")
        centers, edges = construct_bins(lo, hi, n_bins)
        if len(edges) != n_bins + 1:
            # what pd.cut raises in the reference when arange overshoots (src/fields.py:214-216)
            raise ValueError("Bin labels must be one fewer than the number of bin edges")
        sums, counts = handle.vario_bin(max_dist, edges, covariogram)
    finally:
        handle.vario_end()
    with np.errstate(invalid="ignore", divide="ignore"):
        means = sums / counts                 # empty bins: NaN mean, 0 count
    return centers, edges, means, counts


def get_variogram(mf, i: int, j: int, config, device: int = 0) -> pd.DataFrame:
    """DataFrame [bin_center, bin_mean, bin_count] indexed by (i, j, bin) -- src/fields.py:208-232."""
    h = native.Handle(device)
    try:
        h.set_metric(metric_of(config.dist_units, config.fast_dist))
        fi, fj = mf.fields[i], mf.fields[j]
        centers, edges, means, counts = variogram_arrays(h, fi.coords, fi.values, fj.coords, fj.values, i == j,
                                                         config.max_dist, config.n_bins, config.covariogram)
    finally:
        h.close()
    df = pd.DataFrame({"bin_center": centers, "bin_mean": means, "bin_count": counts})
    if (df["bin_count"] < 30).any():
        warnings.warn("WARNING: Fewer than 30 pairs used for at least one bin in variogram calculation.")
    df["i"], df["j"] = i, j
    return df.set_index(["i", "j", df.index])
