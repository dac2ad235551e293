"""GPU diagnostic: table status and assembly timing (exact vs tabulated)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
site_order = 0 if (len(sys.argv) > 2 and sys.argv[2] == "caller") else 1   # "caller": keep the caller's site order

for name, params in (("A", synth.SET_A), ("B", synth.SET_B)):
    pb = synth.conus_problem(n, params=params)
    for exact in (0, 1):
        h = native.Handle(0)
        pv = pb["params"]
        h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        h.set_metric(0)
        for k in range(2):
            h.set_data(k, pb["coords"][k], pb["values"][k])
        h.set_option("exact_cov", exact)
        h.set_option("site_order", site_order)
        h.assemble_joint()
        h.table_fallbacks(reset=True)
        h.assemble_joint()
        t = h.timings()["assemble_sigma_ms"]
        print("fallback entries:", h.table_fallbacks())
        if not exact:
            print(name, [h.table_info(b) for b in range(3)])
        N = 2 * n
        print(f"set {name} exact={exact} site_order={site_order} n={n}: assemble {t:.2f} ms -> {8*N*(N+1)/2/t/1e6:.1f} GB/s")
        h.close()
