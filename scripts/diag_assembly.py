"""GPU diagnostic: table status and assembly timing (exact vs tabulated)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
presort = len(sys.argv) > 2 and sys.argv[2] == "hilbert"


def hilbert_perm(xy):
    lo, hi = xy.min(0), xy.max(0)
    g = np.minimum(((xy - lo) / np.maximum(hi - lo, 1e-300) * 65536).astype(np.int64), 65535)
    x, y = g[:, 0].copy(), g[:, 1].copy()
    d = np.zeros(len(xy), dtype=np.int64)
    s = 32768
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, s - 1 - x, x)
        y = np.where(flip, s - 1 - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s >>= 1
    return np.argsort(d, kind="stable")


for name, params in (("A", synth.SET_A), ("B", synth.SET_B)):
    pb = synth.conus_problem(n, params=params)
    if presort:
        for k in range(2):
            pm = hilbert_perm(pb["coords"][k])
            pb["coords"][k] = np.ascontiguousarray(pb["coords"][k][pm])
            pb["values"][k] = np.ascontiguousarray(pb["values"][k][pm])
    for exact in (0, 1):
        h = native.Handle(0)
        pv = pb["params"]
        h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        h.set_metric(0)
        for k in range(2):
            h.set_data(k, pb["coords"][k], pb["values"][k])
        h.set_option("exact_cov", exact)
        h.assemble_joint()
        h.table_fallbacks(reset=True)
        h.assemble_joint()
        t = h.timings()["assemble_sigma_ms"]
        print("fallback entries:", h.table_fallbacks())
        if not exact:
            print(name, [h.table_info(b) for b in range(3)])
        N = 2 * n
        print(f"set {name} exact={exact} n={n}: assemble {t:.2f} ms -> {8*N*(N+1)/2/t/1e6:.1f} GB/s")
        h.close()
