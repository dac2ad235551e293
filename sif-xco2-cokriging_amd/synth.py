"""Synthetic inputs of the benchmark configurations (BASELINE.json configs).

The reference's own generator (src/sim.py) factorises a dense (2 * grid.count)^2 matrix on the
CPU and cannot reach n_obs = 5k..50k, so the benchmark inputs are built
here: observation sites half co-located / half not, mirroring the sampling scheme of
src/sim.py:67-82, and values from a sum of random cosine waves with the cross-correlation
imposed by mixing (timing does not depend on the values; parity is checked against the oracle
on the same arrays, whatever they are).
"""
from __future__ import annotations

import numpy as np

# flat parameter order of src/model.py:130,145-152
SET_A = [0.99, 0.81, 0.39, 0.695, 1.0, 460.0, 460.0, 460.0, 0.02, 0.025, -0.19]   # generic nu (headline)
SET_B = [1.0, 1.0, 1.5, 1.5, 1.5, 400.0, 400.0, 400.0, 0.02, 0.02, -0.6]          # closed-form nu
SET_B_UNIT = [1.0, 1.0, 1.5, 1.5, 1.5, 0.2, 0.2, 0.2, 0.02, 0.02, -0.6]           # config 2, unit square


def split_sites(pts: np.ndarray, n: int):
    """floor(1.5 n) distinct sites -> two sets of n: ceil(n/2) shared, the rest disjoint
    (the semi-co-located scheme of src/sim.py:67-82)."""
    n_co = int(np.ceil(n / 2))
    n_mis = n - n_co
    assert len(pts) >= n_co + 2 * n_mis
    co = pts[:n_co]
    return np.vstack([co, pts[n_co:n_co + n_mis]]), np.vstack([co, pts[n_co + n_mis:n_co + 2 * n_mis]])


def lattice_sites(rng, count: int, step: float = 0.05, extents=(-125.0, -65.0, 22.0, 58.0)) -> np.ndarray:
    """`count` distinct cell centres [lat, lon] of the 0.05-degree lattice the reference grids its
    residuals on (l2_north_america/create_residuals.ipynb:417)."""
    nlon = int(round((extents[1] - extents[0]) / step))
    nlat = int(round((extents[3] - extents[2]) / step))
    idx = rng.choice(nlat * nlon, size=count, replace=False)
    lat = extents[2] + step / 2 + (idx // nlon) * step
    lon = extents[0] + step / 2 + (idx % nlon) * step
    return np.column_stack([lat, lon])


def cosine_field_pair(rng, c0: np.ndarray, c1: np.ndarray, scale: float, rho: float, n_waves: int = 256):
    """Two unit-variance fields with co-located correlation ~rho: sums of random cosine waves."""
    k = rng.standard_normal((n_waves, 2)) / scale
    ph = rng.uniform(0, 2 * np.pi, (2, n_waves))
    amp = np.sqrt(2.0 / n_waves)

    def f(c, p):
        return amp * np.cos(c @ k.T + p).sum(axis=1)

    a0, a1 = f(c0, ph[0]), f(c1, ph[0])
    b1 = f(c1, ph[1])
    return a0, rho * a1 + np.sqrt(max(0.0, 1 - rho * rho)) * b1


def conus_problem(n: int, seed: int = 20003, params=SET_A):
    """BASELINE config 3/4 shape: n sites per process on the 0.05-degree CONUS lattice,
    haversine metric, prediction grid = full 0.5-degree rectangle of prediction_coords()
    defaults (73 x 121 = 8 833 points, src/joint_prediction.py:277-283)."""
    rng = np.random.default_rng(seed)
    pts = lattice_sites(rng, int(np.floor(1.5 * n)))
    c0, c1 = split_sites(pts, n)
    vr = np.random.default_rng(seed + 10000)
    z0, z1 = cosine_field_pair(vr, c0, c1, scale=6.0, rho=params[10])
    z0 *= params[0]
    z1 *= params[1]
    lat = np.arange(22.0, 58.0 + 0.25, 0.5)
    lon = np.arange(-125.0, -65.0 + 0.25, 0.5)
    la, lo = np.meshgrid(lat, lon, indexing="ij")
    grid = np.column_stack([la.ravel(), lo.ravel()])
    return dict(coords=[c0, c1], values=[z0, z1], pcoords=grid, params=list(params), metric=0)


def unit_square_problem(n: int, grid_side: int = 100, seed: int = 20002, params=SET_B_UNIT):
    """BASELINE config 2 shape: n sites per process uniform on the unit square, Euclidean metric,
    grid_side^2 prediction grid (as research/simulation_experiment.ipynb:772-775)."""
    rng = np.random.default_rng(seed)
    pts = rng.random((int(np.floor(1.5 * n)), 2))
    c0, c1 = split_sites(pts, n)
    vr = np.random.default_rng(seed + 10000)
    z0, z1 = cosine_field_pair(vr, c0, c1, scale=0.15, rho=params[10])
    g = np.linspace(0, 1, grid_side)
    gx, gy = np.meshgrid(g, g, indexing="ij")
    return dict(coords=[c0, c1], values=[z0, z1], pcoords=np.column_stack([gx.ravel(), gy.ravel()]),
                params=list(params), metric=1)


def residual_tables(n_sif: int = 47562, n_xco2: int = 23080, seed: int = 20001, params=SET_A):
    """BASELINE config 1's input shape: gridded-residual tables with the column schema of the reference's
    l2_north_america CSVs (l2_north_america/empirical_semivariogram.ipynb cell 3: lon, lat, evi, sif, lon_std, lat_std,
    evi_std, ols_mean, sif_residuals, sif_residuals_std) and their row counts (47 562 SIF cells, 23 080 XCO2 cells of the
    0.05-degree lattice, create_residuals.ipynb:732,1758).  The real files are not in the reference's repository
    (.MISSING_LARGE_BLOBS); values here are synthetic (random cosine waves, see cosine_field_pair).  Returns two pandas
    DataFrames (SIF, XCO2); write them with DataFrame.to_csv to rehearse the CSV plumbing."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    pts = lattice_sites(rng, n_sif + n_xco2 // 2)
    c_sif = pts[:n_sif]
    c_x = np.vstack([pts[:n_xco2 - n_xco2 // 2], pts[n_sif:]])          # half of the XCO2 cells are SIF cells too
    z_sif, z_x = cosine_field_pair(np.random.default_rng(seed + 10000), c_sif, c_x, scale=6.0, rho=params[10])

    def table(c, resid, name, sd):
        n = len(c)
        ols = 0.4 + 0.05 * np.cos(np.radians(c[:, 0]) * 3.0)
        cols = {"lon": c[:, 1], "lat": c[:, 0], "evi": 0.3 + 0.1 * rng.standard_normal(n), name: ols + sd * resid,
                "lon_std": np.full(n, 0.0144), "lat_std": np.full(n, 0.0144), "evi_std": 0.02 * np.ones(n), "ols_mean": ols,
                f"{name}_residuals": sd * resid, f"{name}_residuals_std": resid}
        return pd.DataFrame(cols)

    return table(c_sif, z_sif * params[0], "sif", 0.17), table(c_x, z_x * params[1], "xco2", 1.1)
