"""A/B, interleaved in one process: ck_factor + ck_predict against ck_factor_predict (the two sweeps overlapped)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pb = synth.conus_problem(n, seed=20003)
pv = pb["params"]
h = native.Handle(0)
h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
h.set_metric(pb["metric"])
for k in range(2):
    h.set_data(k, pb["coords"][k], pb["values"][k])
pc = pb["pcoords"]
ref = None
variants = [("plain", {}), ("fused", {"fused_prio": 0}), ("fusedS", {"fused_prio": 1}), ("fused=", {"fused_prio": 2}),
            ("fusedG2", {"fused_group": 2}), ("fusedG4", {"fused_group": 4}), ("fusedG6", {"fused_group": 6}),
            ("fusedLA", {"fused_la": 1}), ("fusedLAG2", {"fused_la": 1, "fused_group": 2}), ("fusedLAG4", {"fused_la": 1, "fused_group": 4}),
            ("fusedLAG1", {"fused_la": 1, "fused_group": 1})]
if len(sys.argv) > 3:
    variants = [v for v in variants if v[0] in sys.argv[3].split(",")]
for it in range(reps):
    for mode, opts in variants:
        h.set_option("fused_sweeps", 1)   # overlapped whatever the size (the automatic rule stops at 128 panels)
        h.set_option("fused_prio", 0)
        h.set_option("fused_group", 0)
        h.set_option("fused_la", 0)   # variants name the look-ahead explicitly
        for k_, v_ in opts.items():
            h.set_option(k_, v_)
        h.assemble_joint()
        t0 = time.perf_counter()
        if mode == "plain":
            assert h.factor() == 0
            pred, err = h.predict(0, pc)
        else:
            info, pred, err = h.factor_predict(0, pc)
            assert info == 0
        wall = (time.perf_counter() - t0) * 1e3
        t = h.timings()
        if ref is None:
            ref = (pred, err)
        dp = np.max(np.abs(pred - ref[0])) / np.max(np.abs(ref[0]))
        de = np.max(np.abs(err - ref[1])) / np.max(np.abs(ref[1]))
        print(f"N={2*n} m={len(pc)} {mode:6s} wall {wall:7.1f} ms | factor {t['factor_ms']:.1f} solve {t['solve_ms']:.1f} "
              f"both {t['fused_sweeps_ms'] if mode != 'plain' else t['factor_ms'] + t['solve_ms']:.1f} | diff pred {dp:.1e} err {de:.1e}", flush=True)
