"""Simulate and sample a bivariate Matern field -- mirror of the reference's ``sim`` module
(src/sim.py) with the dense work on the GPU: the joint covariance of the two processes on the grid
is assembled by the K1 kernel (Euclidean metric), factored by the blocked Cholesky (K3) and the
draw is z = L eps (``ck_sample``).  Random numbers come from the same
``numpy.random.default_rng(seed)`` stream in the same order as the reference (src/sim.py:37,53),
the semi-co-located sampling scheme and its pandas calls follow src/sim.py:56-117, so a seed gives
the same experiment.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from . import native
from .fields import Field, MultiField
from .model import configure_handle


class CartesianGrid:
    """Regular Cartesian grid in Euclidean space (src/sim.py:11-27); node order x-major."""

    def __init__(self, xbounds: tuple = (0, 1), ybounds: tuple = (0, 1), xcount=51, ycount=51) -> None:
        xs = np.linspace(*xbounds, num=xcount)
        ys = np.linspace(*ybounds, num=ycount)
        xx, yy = np.meshgrid(xs, ys, indexing="ij")          # all (x, y) combinations, x varying slowest
        self.coords = pd.DataFrame({"x": xx.ravel(), "y": yy.ravel()})
        self.count = len(self.coords)


class BivariateRandomField:
    """src/sim.py:30-54."""

    def __init__(self, model, grid: CartesianGrid, seed: int = None, device: int = 0) -> None:
        self.seed = seed
        self.rng = np.random.default_rng(seed)
        self.mod, self.grid, self.coords = model, grid, grid.coords
        xy = grid.coords.values
        h = native.Handle(device)
        configure_handle(h, model)
        h.set_metric(native.METRIC_EUCLID)
        h.set_option("site_order", 0)   # z = L eps: the draw for a given eps depends on the order of the sites
        zero = np.zeros(grid.count)
        h.set_data(0, xy, zero)
        h.set_data(1, xy, zero)
        h.assemble_joint()
        info = h.factor()
        if info != 0:
            from numpy.linalg import LinAlgError
            raise LinAlgError(f"{info}-th leading minor of the array is not positive definite")
        noise = self.rng.standard_normal(2 * grid.count)
        z = h.sample(noise)
        h.close()
        self.fields = [pd.DataFrame({"x": xy[:, 0], "y": xy[:, 1], "value": z[k * grid.count:(k + 1) * grid.count]})
                       for k in range(2)]

    def _split_samp_coords(self, size: int, seed: int) -> list:
        """Half the samples co-located, half not (src/sim.py:56-72)."""
        ext = int(np.floor(1.5 * size))
        n_co = int(np.ceil(size / 2))
        n_mis = size - n_co
        assert ext >= n_co + 2 * n_mis
        coords = self.coords.sample(n=ext, random_state=seed, replace=False)
        co = coords.iloc[:n_co, :]
        mis = [coords.iloc[n_co:n_co + n_mis, :], coords.iloc[n_co + n_mis:, :]]
        return [pd.concat((co, mis[i])) for i in range(2)]

    def sample(self, size: int = None, frac: float = None, epsilon: list = [0], seed: int = None) -> list:
        """src/sim.py:74-117."""
        if frac is not None:
            size = int(np.ceil(frac * self.grid.count))
        assert 1.5 * size <= self.grid.count, "Sample size is too large for semi-colocated sampling scheme."
        epsilon = np.array(epsilon)
        if epsilon.size == 1:
            epsilon = np.repeat(epsilon, 2)
        if seed is not None:
            self.rng = np.random.default_rng(seed)
        else:
            seed = self.seed
        coords = self._split_samp_coords(size, seed)
        samples = [pd.merge(self.fields[i], coords[i]) for i in range(2)]
        for i, df in enumerate(samples):
            df["value"] += self.rng.normal(scale=epsilon[i], size=size)
            df.rename(columns={"value": f"Z{i}"}, inplace=True)
        return samples

    def to_fields(self, samples: list, i: int = None) -> MultiField:
        """Samples as a MultiField; each process' sites in (x, y) order, which is what the
        reference's xarray round trip produces (src/sim.py:127-137, src/fields.py:91-94)."""
        fl = []
        for j in ([0, 1] if i is None else [i]):
            s = samples[j].sort_values(["x", "y"])
            fl.append(Field(s[["x", "y"]].values, s[f"Z{j}"].values))
        return MultiField(fl)
