"""GPU diagnostic: device-backed MultivariateMatern.fit against the reference's recorded result."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.conftest import load_golden
from tests.test_gpu_fit import _estimate, _fit_groups_oracle
from oracle import cokrige_oracle as orc
from sif_xco2_cokriging_amd import model

g = load_golden("model_fit")
est = _estimate(g)
groups = _fit_groups_oracle(g)
np.set_printoptions(precision=6, suppress=True, linewidth=200)
mod = model.MultivariateMatern(n_procs=2)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mod.fit(est)
x = mod.params.get_values().astype(float)
print("ref  x", g["fit_x"], float(g["fit_cost"]))
print("ours x", x, mod.fit_result.cost)
print("oracle cost at ours", orc.composite_wls(x, groups), " at ref", orc.composite_wls(g["fit_x"], groups))
xo, co, ok = orc.fit(groups, x0=x)
print("oracle restarted from ours:", xo, co, ok)
xo, co, ok = orc.fit(groups, x0=g["fit_x"])
print("oracle restarted from ref :", xo, co, ok)
