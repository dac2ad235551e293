"""Diagnostic: the cooperative panel step against the launch-per-dependency panel step, block by block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native

rng = np.random.default_rng(3)
for n in (200, 700):
    N = 2 * n
    coords = [np.column_stack([rng.uniform(25, 49, n), rng.uniform(-124, -67, n)]) for _ in range(2)]
    values = [rng.standard_normal(n), rng.standard_normal(n)]
    pv = [0.99, 0.81, 0.39, 0.695, 1.0, 460, 460, 460, 0.02, 0.025, -0.19]
    L = {}
    for name, fused in (("plain", 2), ("coop", 18)):
        h = native.Handle(0)
        h.set_option("site_order", 0)
        h.set_option("panel_fused", fused)
        h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        h.set_metric(0)
        for k in range(2):
            h.set_data(k, coords[k], values[k])
        h.assemble_joint()
        info = h.factor()
        t = h.timings()
        L[name] = h.debug_get_lower(N)
        print(f"N={N} {name}: info {info} redone {t['panel_coop_redone']}", flush=True)
        h.close()
    d = np.abs(L["coop"] - L["plain"])
    nb = -(-N // 64)
    print("max |diff| per 64 x 64 block (rows down, cols across):")
    for bi in range(nb):
        print(" ".join(f"{d[64*bi:64*bi+64, 64*bj:64*bj+64].max():8.1e}" for bj in range(bi + 1)))
