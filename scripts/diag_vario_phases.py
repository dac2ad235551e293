"""Wall time of the phases of one variogram call at n soundings (config 5): host centring, ck_vario_begin (Hilbert sort on
the host, upload, unit vectors, bounding balls), ck_vario_extent, bin construction, ck_vario_bin, ck_vario_end."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native
from sif_xco2_cokriging_amd.variogram import construct_bins

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(20005)
c0 = np.column_stack([rng.uniform(22, 58, n), rng.uniform(-125, -65, n)])
v0 = rng.standard_normal(n)
h = native.Handle(0)
h.set_metric(0)
for rep in range(3):
    t = [time.perf_counter()]
    r = v0 - v0.mean(); t.append(time.perf_counter())
    h.vario_begin(c0, r); t.append(time.perf_counter())
    lo, hi, npos = h.vario_extent(1500.0); t.append(time.perf_counter())
    centers, edges = construct_bins(lo, hi, 30); t.append(time.perf_counter())
    sums, counts = h.vario_bin(1500.0, edges, False); t.append(time.perf_counter())
    h.vario_end(); t.append(time.perf_counter())
    names = ["centre", "begin", "extent", "bins", "bin", "end"]
    print(json.dumps({"n": n, **{k: round((b - a) * 1e3, 2) for k, a, b in zip(names, t[:-1], t[1:])},
                      "total_ms": round((t[-1] - t[0]) * 1e3, 2), "bin_kernel_ms": h.timings()["vario_bin_ms"]}))
