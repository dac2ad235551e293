#!/usr/bin/env python3
"""research/simulation_experiment.ipynb (cells 3-5 and 11 of the reference) on the MI355X path:
simulate a bivariate Matern field on the 51 x 51 unit grid, sample 100 semi-co-located sites per
process, cokrige process 1 on the full grid -- every dense operation on the GPU.

The notebook's recorded outputs (research/simulation_experiment.ipynb:762-763):
    pred      1.025 1.129 1.177 1.106 ... -0.3236 -0.2804 -0.2439
    pred_err  0.2072 0.1824 0.1494 0.0871 ... 0.6993 0.7249 0.754
"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sif_xco2_cokriging_amd import joint_prediction as prediction  # noqa: E402
from sif_xco2_cokriging_amd import model, sim  # noqa: E402


def run():
    # notebook configuration (cell 3)
    seed = 1
    param_vals = [1.0, 1.0, 1.5, 1.5, 1.5, 0.2, 0.2, 0.2, 0.0, 0.0, -0.6]
    grid_size, samp_size, meas_err = 51, 100, 0.01
    true_params = model.MaternParams().set_values(param_vals)
    true_mod = model.MultivariateMatern(params=true_params)
    grid = sim.CartesianGrid(xcount=grid_size, ycount=grid_size)
    rf = sim.BivariateRandomField(true_mod, grid, seed=seed)
    samples = rf.sample(size=samp_size, epsilon=np.sqrt(meas_err))
    mf = rf.to_fields(samples)
    cokrig = prediction.Predictor(true_mod, mf, fast_dist=False, dist_units=None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")   # prediction sites on data sites: the reference warns too (:354)
        pred, err = cokrig.predict_arrays(1, grid.coords.values)
    return pred, err


if __name__ == "__main__":
    pred, err = run()
    np.set_printoptions(precision=4, suppress=True)
    print("pred     ", pred[:4], "...", pred[-3:])
    print("pred_err ", err[:4], "...", err[-3:])
