"""ck_verify_model (the reference's _verify_model as the Cholesky of the m x m Schur complement) after a prediction at the headline size:
wall time per call (52.5 ms: 3.1e12 + 2.3e11 flop); run under rocprofv3 --kernel-trace --stats for its kernels."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sif_xco2_cokriging_amd import native, synth
pb = synth.conus_problem(20000, seed=20003)
pv, pc = pb["params"], pb["pcoords"]
h = native.Handle(0)
h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
h.set_metric(pb["metric"])
for k in range(2):
    h.set_data(k, pb["coords"][k], pb["values"][k])
h.assemble_joint()
assert h.factor() == 0
for r in range(3):
    h.predict(0, pc)
    h.synchronize()
    t0 = time.perf_counter()
    info = h.verify_model()
    print("verify_model", info, (time.perf_counter() - t0) * 1e3, "ms", flush=True)
