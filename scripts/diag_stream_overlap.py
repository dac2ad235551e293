"""GPU diagnostic: small dependent kernels on the side stream under a chip-filling trailing update (ck_debug_stream_overlap)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
pb = synth.conus_problem(n, seed=20003)
pv = pb["params"]
h = native.Handle(0)
h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
h.set_metric(pb["metric"])
for k in range(2):
    h.set_data(k, pb["coords"][k], pb["values"][k])
for rows in (512, 2048, 20480):
    for rep in range(2):
        for mode in (0, 1, 2):
            h.assemble_joint()
            o = h.stream_overlap(mode, rows, 6)
            print(f"rows {rows:6d} mode {mode}: update {o[0]:8.3f} ms | side kernels end at " + " ".join(f"{x:7.3f}" for x in o[1:]), flush=True)
