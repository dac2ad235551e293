"""Wall time of joint_prediction.Predictor.__call__ (the drop-in entry, Python included) against the native calls bench.py times."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sif_xco2_cokriging_amd import fields, joint_prediction, model, synth
pb = synth.conus_problem(20000, seed=20003)
mod = model.MultivariateMatern()
mod.params.set_values(pb["params"])
mf = fields.MultiField([fields.Field(pb["coords"][0], pb["values"][0]), fields.Field(pb["coords"][1], pb["values"][1])])
pc = pb["pcoords"]
P = joint_prediction.Predictor(mod, mf)
for verify in (False, True):
  P.verify_model = verify
  print("verify_model =", verify, "(the reference's _verify_model: here the Cholesky of the m x m Schur complement)")
  for r in range(3):
    t0 = time.perf_counter(); a = P(0, pc, postprocess=False); t1 = time.perf_counter()
    b = P(1, pc, postprocess=False); t2 = time.perf_counter()
    P.invalidate()
    print("  " + f"Predictor.__call__: new model {1e3*(t1-t0):.1f} ms, second field on the resident factor {1e3*(t2-t1):.1f} ms", flush=True)
