"""GPU A/B in one process: factor / solve time of the headline problem for
panel_group,gemm_variant[,lookahead] combinations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
combos = [tuple(int(x) for x in a.split(",")) for a in (sys.argv[2:] or ["1,4", "4,4", "1,5", "4,5", "8,5"])]
pb = synth.conus_problem(n)
pv = pb["params"]
h = native.Handle(0)
h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
h.set_metric(0)
for k in range(2):
    h.set_data(k, pb["coords"][k], pb["values"][k])
h.set_option("time_gemm", 1)
res = {}
for rnd in range(3):
    for c in combos:
        g, v = c[0], c[1]
        la = c[2] if len(c) >= 3 else 0
        h.set_option("panel_group", g)
        h.set_option("gemm_variant", v)
        h.set_option("lookahead", la)
        h.assemble_joint()
        info = h.factor()
        assert info == 0 or os.environ.get("CK_AB_NOCHECK"), info   # kernel-timing experiments produce garbage on purpose
        pred, err = h.predict(0, pb["pcoords"])
        t = h.timings()
        res.setdefault(c, []).append((t["factor_ms"], t["solve_ms"], t["syrk_ms"], t["aux_gemm_ms"], float(pred[17])))
for k, r in res.items():
    a = np.array(r)[1:]
    print(f"group {k[0]:2d} variant {k[1]} lookahead {k[2:]}: factor {a[:,0].mean():7.1f} ms  solve {a[:,1].mean():7.1f} ms  "
          f"syrk {a[:,2].mean():7.1f}  auxgemm {a[:,3].mean():7.1f}  pred[17] {a[0,4]:.12f}")
