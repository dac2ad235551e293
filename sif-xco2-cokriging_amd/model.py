"""Host-side mirror of the covariance part of the reference's ``model`` module
(src/model.py): parameter containers and ``MultivariateMatern`` with the same
attribute and method names the notebooks use.  The numbers come from the HIP
library (``ck_cov_lags``): distance -> Matern auto/cross-covariance with a
device K_nu; nothing is evaluated in numpy/scipy here.

``fit`` (composite weighted least squares, src/model.py:277-317) keeps
the reference's optimiser call -- scipy L-BFGS-B with its finite-difference gradient -- and
evaluates the model variograms of every cost-function call in ONE device launch
(``ck_model_variogram``).
"""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd
from scipy.optimize import minimize

from . import native


class _Param:
    """An n_procs x n_procs parameter array, NaN where the parameter does not exist
    (cf. MarginalParam / CrossParam / RhoParam, src/model.py:16-106)."""

    def __init__(self, name, default, bounds, n_procs, where):
        self.name, self.default, self.bounds, self.n_procs = name, default, bounds, n_procs
        self._where = where  # "diag" | "triu" | "striu"
        self.values = np.full((n_procs, n_procs), np.nan)
        self.reset_values()

    def _index(self):
        if self._where == "diag":
            return np.diag_indices(self.n_procs)
        return np.triu_indices(self.n_procs, k=0 if self._where == "triu" else 1)

    def get_names(self):
        r, c = self._index()
        return [f"{self.name}_{i + 1}{j + 1}" for i, j in zip(r, c)]

    def get_values(self):
        return self.values[self._index()]

    def set_values(self, x):
        self.values[self._index()] = x
        return self

    def reset_values(self):
        self.values[self._index()] = self.default
        return self

    def count_params(self):
        return len(self._index()[0])

    def to_dataframe(self):
        return pd.DataFrame({"name": self.get_names(), "value": self.get_values(),
                             "bounds": [self.bounds] * self.count_params()})


class MaternParams:
    """sigma, nu, len_scale, nugget, rho with the reference's defaults, bounds and flat
    order sigma_11 sigma_22 nu_11 nu_12 nu_22 len_11 len_12 len_22 nugget_11 nugget_22 rho_12
    (src/model.py:109-169)."""

    def __init__(self, n_procs: int = 2) -> None:
        self.n_procs = n_procs
        self.sigma = _Param("sigma", 1.0, (0.4, 3.5), n_procs, "diag")
        self.nu = _Param("nu", 1.5, (0.2, 3.5), n_procs, "triu")
        self.len_scale = _Param("len_scale", 5e2, (1e2, 2e3), n_procs, "triu")
        self.nugget = _Param("nugget", 0.0, (0.0, 0.2), n_procs, "diag")
        self.rho = _Param("rho", np.nan if n_procs == 1 else 0.0, (-1.0, 1.0), n_procs, "striu")
        self._params = [self.sigma, self.nu, self.len_scale, self.nugget, self.rho]
        self.n_params = sum(p.count_params() for p in self._params)

    def to_dataframe(self):
        return pd.concat([p.to_dataframe() for p in self._params], ignore_index=True)

    def get_names(self):
        return self.to_dataframe()["name"].values

    def get_values(self):
        return self.to_dataframe()["value"].values

    def get_bounds(self):
        return self.to_dataframe()["bounds"].values

    def set_values(self, x):
        x = np.asarray(x, dtype=float)
        if len(x) != self.n_params:
            raise ValueError("Incorrect number of parameters in input array.")
        at = 0
        for p in self._params:
            k = p.count_params()
            p.set_values(x[at:at + k])
            at += k
        return self

    def reset_values(self):
        for p in self._params:
            p.reset_values()
        return self

    def set_bounds(self, **kwargs):
        for name, bounds in kwargs.items():
            if name not in ("sigma", "nu", "len_scale", "nugget", "rho"):
                raise AttributeError(f"`{name}` is not a valid parameter.")
            getattr(self, name).bounds = bounds
        return self


def model_arrays(mod):
    """(n_procs, sigma, nu3, len3, nugget, rho12) from any object with the reference's
    ``mod.params.<name>.values`` layout (ours or the reference's own MultivariateMatern)."""
    p = mod.params
    n = int(mod.n_procs)
    sig = np.diag(np.asarray(p.sigma.values, dtype=float)).copy()
    nug = np.diag(np.asarray(p.nugget.values, dtype=float)).copy()
    nu = np.asarray(p.nu.values, dtype=float)
    ls = np.asarray(p.len_scale.values, dtype=float)
    if n == 1:
        return 1, sig, np.array([nu[0, 0]] * 3), np.array([ls[0, 0]] * 3), nug, 0.0
    if n != 2:
        raise ValueError("the HIP path supports n_procs = 1 or 2")
    rho = float(np.asarray(p.rho.values, dtype=float)[0, 1])
    return 2, sig, np.array([nu[0, 0], nu[0, 1], nu[1, 1]]), np.array([ls[0, 0], ls[0, 1], ls[1, 1]]), nug, rho


def configure_handle(h: "native.Handle", mod):
    n, sig, nu, ls, nug, rho = model_arrays(mod)
    h.set_model(n, sig, nu, ls, nug, rho)


class MultivariateMatern:
    """Multivariate Matern covariance model (Gneiting et al., 2010) -- same public
    surface as src/model.py:172-222 for the covariance functions."""

    def __init__(self, n_procs: int = 2, params: MaternParams = None, device: int = 0) -> None:
        self.n_procs = n_procs
        self.params = MaternParams(n_procs=n_procs) if params is None else params
        self.fit_result = None
        self._device = device
        self._h = None

    def _handle(self):
        if self._h is None:
            self._h = native.Handle(self._device)
        configure_handle(self._h, self)   # parameters may have been edited in place
        return self._h

    def _eval(self, i, j, h, use_nugget):
        h = np.atleast_1d(np.asarray(h, dtype=np.float64))
        return self._handle().cov_lags(i, j, h, use_nugget=use_nugget)

    def covariance(self, i: int, h, use_nugget: bool = True) -> np.ndarray:
        """sigma_i^2 rho_ii(h) + nugget_i [h == 0]   (src/model.py:193-197)."""
        return self._eval(i, i, h, use_nugget)

    def cross_covariance(self, i: int, j: int, h) -> np.ndarray:
        """rho_ij prod(sigma) rho^Matern_ij(h)   (src/model.py:199-207)."""
        if i > j:
            i, j = j, i
        return self._eval(i, j, h, False)

    def correlation(self, i: int, j: int, h) -> np.ndarray:
        """Matern correlation with (nu_ij, len_scale_ij) -- src/model.py:188-191; it does not depend on sigma or
        rho, so it is evaluated with unit amplitudes (rho_12 = 0 is a valid model and must not divide by zero)."""
        if i > j:
            i, j = j, i
        n, sig, nu, ls, nug, rho = model_arrays(self)
        if self._h is None:
            self._h = native.Handle(self._device)
        self._h.set_model(n, np.ones_like(sig), nu, ls, np.zeros_like(nug), 1.0)
        h = np.atleast_1d(np.asarray(h, dtype=np.float64))
        return self._h.cov_lags(i, j, h, use_nugget=False)   # the next _handle() call restores the model's amplitudes

    def semivariance(self, i: int, h) -> np.ndarray:
        """src/model.py:209-213."""
        s2 = self.params.sigma.values[i, i] ** 2
        return s2 - self._eval(i, i, h, False) + self.params.nugget.values[i, i]

    def cross_semivariance(self, i: int, j: int, h) -> np.ndarray:
        """src/model.py:215-222."""
        sill = 0.5 * np.nansum(self.params.sigma.values ** 2 + self.params.nugget.values)
        return sill - self.cross_covariance(i, j, h)

    def get_variogram(self, i: int, j: int, h, kind: str) -> pd.DataFrame:
        """src/model.py:224-237."""
        h = np.asarray(h, dtype=np.float64)
        v = self._handle().model_variogram(i, j, h, kind="covariogram" if kind == "covariogram" else "semivariogram")
        df = pd.DataFrame({"distance": h, "variogram": v, "i": i, "j": j})
        return df.set_index(["i", "j", df.index])

    def variograms(self, h, kind: str = "semivariogram") -> pd.DataFrame:
        """Modelled variograms and cross-variogram(s) at the given lags (src/model.py:239-248)."""
        return pd.concat([self.get_variogram(i, j, h, kind)
                          for i in range(self.n_procs) for j in range(self.n_procs) if i <= j])

    @staticmethod
    def _weighted_least_squares(ydata: np.ndarray, yfit: np.ndarray, bin_counts: np.ndarray) -> float:
        """Cressie (1985) weighted least squares with the reference's handling of fit == 0
        (src/model.py:250-264)."""
        ydata, yfit, bin_counts = (np.asarray(a, dtype=np.float64) for a in (ydata, yfit, bin_counts))
        zero = yfit == 0.0
        wls = np.zeros_like(yfit)
        wls[zero] = bin_counts[zero] * ydata[zero] ** 2
        nz = ~zero
        wls[nz] = bin_counts[nz] * ((ydata[nz] - yfit[nz]) / yfit[nz]) ** 2
        return np.sum(wls)

    def _map_fit(self, df_vario: pd.DataFrame) -> pd.DataFrame:
        """New ``fit`` column: the model semivariogram at ``bin_center`` for every (i, j) group
        (src/model.py:266-275), all groups in one device launch; rows come back grouped by (i, j)
        in sorted order like ``groupby(level=[0, 1]).apply``."""
        df = df_vario.sort_index(level=[0, 1], sort_remaining=False, kind="stable").copy()
        i = df.index.get_level_values(0).values.astype(np.int32)
        j = df.index.get_level_values(1).values.astype(np.int32)
        lo, hi = np.minimum(i, j), np.maximum(i, j)   # cross-semivariance is symmetric (:216-218)
        df["fit"] = self._handle().model_variogram(lo, hi, df["bin_center"].values.astype(np.float64))
        return df

    def _composite_wls(self, p, df_vario: pd.DataFrame) -> float:
        """Composite WLS cost (src/model.py:277-283, 389-391)."""
        self.params.set_values(p)
        df = self._map_fit(df_vario)
        ydata, yfit, counts = df[["bin_mean", "fit", "bin_count"]].T.values.astype(np.float64)
        nz = yfit != 0.0
        return np.sum(counts[nz] * ((ydata[nz] - yfit[nz]) / yfit[nz]) ** 2)

    def fit(self, estimate, guess: MaternParams = None, polish: bool = False):
        """Fit the parameters to the empirical (cross-)semivariograms simultaneously by composite
        weighted least squares -- same flow, optimiser and warning as src/model.py:285-317.
        ``polish`` (not in the reference; off by default, so the default call is the reference's): restart the
        optimiser once from its own answer and keep the better of the two -- L-BFGS-B on finite-difference gradients
        stops early on this cost's flat valley floors (the reference's recorded run: 1618.19, restarted: 1608.19)."""
        if estimate.config.n_procs != self.n_procs:
            raise ValueError("Number of theoretical processes different from empirical processes.")
        if guess is None:
            init_params = self.params.reset_values().get_values()
        else:
            init_params = self.params.get_values()
            self.params.set_bounds(**{p.name: p.bounds for p in guess._params})
        bounds = self.params.get_bounds()
        optim_result = minimize(self._composite_wls, init_params, args=(estimate.df,), method="L-BFGS-B", bounds=bounds)
        if polish:
            again = minimize(self._composite_wls, optim_result.x, args=(estimate.df,), method="L-BFGS-B", bounds=bounds)
            if again.fun <= optim_result.fun:
                optim_result = again
        if optim_result.success == False:   # noqa: E712  (as the reference)
            warnings.warn("ERROR: optimization did not converge.")
        self.params.set_values(optim_result.x)
        self.fit_result = FittedVariogram(self, estimate, optim_result.fun)
        return self


class FittedVariogram:
    """Model parameters and theoretical variogram for the corresponding empirical variogram
    (src/model.py:320-347)."""

    def __init__(self, model: MultivariateMatern, estimate, cost: float) -> None:
        self.config = estimate.config
        self.timestamp = estimate.timestamp
        self.timedeltas = estimate.timedeltas
        self.df_empirical = estimate.df
        h = np.linspace(0, self.df_empirical["bin_center"].max(), 100)
        self.df_theoretical = model.variograms(h)
        self.params = model.params
        self.cost = cost
        self.cs_valid = self.cs_check()

    def cs_check(self):
        """Placeholder in the reference as well (src/model.py:337-347): always None."""
        return None
