"""A/B, interleaved in one process: ck_factor_predict as ONE sweep over the tall matrix [Sigma; c0^T; z^T] (round 4,
option tall_sweep) against round 3's two overlapped sweeps and the plain sequence.

    python scripts/ab_tall.py [n_obs=20000] [reps=3] [variants=a,b,..] [config=2|1]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 2
pb = synth.conus_problem(n, seed=20003) if cfg == 2 else synth.unit_square_problem(n, grid_side=100)
pv = pb["params"]
h = native.Handle(0)
h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
h.set_metric(pb["metric"])
for k in range(2):
    h.set_data(k, pb["coords"][k], pb["values"][k])
pc = pb["pcoords"]
ref = None
variants = [("plain", {}), ("two", {"tall_sweep": 0}), ("tall", {"tall_sweep": 1}), ("tallG3", {"tall_sweep": 1, "fused_group": 3, "panel_group": 3, "group_first": 0}), ("tallT", {"tall_sweep": 1, "time_gemm": 1}),
            ("tallG1", {"tall_sweep": 1, "fused_group": 1}), ("tallG2", {"tall_sweep": 1, "fused_group": 2}),
            ("tallG4", {"tall_sweep": 1, "fused_group": 4}), ("tallG5", {"tall_sweep": 1, "fused_group": 5}),
            ("tallG6", {"tall_sweep": 1, "fused_group": 6}),
            ("tallF1", {"tall_sweep": 1, "group_first": 1}), ("tallG4F1", {"tall_sweep": 1, "fused_group": 4, "group_first": 1}),
            ("tallG4F2", {"tall_sweep": 1, "fused_group": 4, "group_first": 2}),
            ("tallF1T", {"tall_sweep": 1, "group_first": 1, "group_tail": 1, "group_tail_panels": 12}),
            ("tallG4F1T", {"tall_sweep": 1, "fused_group": 4, "group_first": 1, "group_tail": 2, "group_tail_panels": 16}),
            ("tallG4F1T1", {"tall_sweep": 1, "fused_group": 4, "group_first": 1, "group_tail": 1, "group_tail_panels": 12}),
            ("tallG5F1", {"tall_sweep": 1, "fused_group": 5, "group_first": 1}),
            ("tallG4F3", {"tall_sweep": 1, "fused_group": 4, "group_first": 3}), ("tallG5F2", {"tall_sweep": 1, "fused_group": 5, "group_first": 2}),
            ("tallG5F3", {"tall_sweep": 1, "fused_group": 5, "group_first": 3}), ("tallG6F2", {"tall_sweep": 1, "fused_group": 6, "group_first": 2}),
            ("tallG6F3", {"tall_sweep": 1, "fused_group": 6, "group_first": 3}), ("tallG3F2", {"tall_sweep": 1, "fused_group": 3, "group_first": 2}),
            ("tallG4F2T", {"tall_sweep": 1, "fused_group": 4, "group_first": 2, "group_tail": 2, "group_tail_panels": 12}),
            ("tallB2off", {"tall_b2_stream": 0}), ("tallS0", {"tall_split": 0}), ("tallThick", {"tall_thin": 0}), ("tallS1", {"tall_split": 1}), ("tallS2R8", {"tall_split": 2, "tall_split_rows": 8 * 512}),
            ("tallS2R16", {"tall_split": 2, "tall_split_rows": 16 * 512}), ("tallS2R32", {"tall_split": 2, "tall_split_rows": 32 * 512}),
            ("tallG2F1", {"tall_sweep": 1, "fused_group": 2, "group_first": 1}), ("tallG3F1", {"tall_sweep": 1, "fused_group": 3, "group_first": 1}),
            ("tallG3F2", {"tall_sweep": 1, "fused_group": 3, "group_first": 2}),
            ("tallG2T", {"tall_sweep": 1, "fused_group": 2, "group_tail": 1, "group_tail_panels": 4}),
            ("tallG2S1", {"tall_sweep": 1, "fused_group": 2, "tall_split": 1})]
if len(sys.argv) > 3 and sys.argv[3] != "all":
    variants = [v for v in variants if v[0] in sys.argv[3].split(",")]
N = 2 * n
m = len(pc)
flops = N ** 3 / 3 + N ** 2 * m
for it in range(reps):
    for mode, opts in variants:
        h.set_option("fused_sweeps", 1)
        h.set_option("fused_group", 0)
        h.set_option("tall_sweep", 1)
        h.set_option("time_gemm", 0)
        h.set_option("group_first", -1)
        for k_ in ("group_tail", "group_tail_panels"):
            h.set_option(k_, 0)
        h.set_option("tall_split", 2)
        h.set_option("tall_thin", 1)
        h.set_option("tall_b2_stream", 1)
        h.set_option("tall_split_rows", 24 * 512)
        for k_, v_ in opts.items():
            h.set_option(k_, v_)
        h.assemble_joint()
        h.synchronize()
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
        both = t['fused_sweeps_ms'] if mode != 'plain' else t['factor_ms'] + t['solve_ms']
        print(f"N={N} m={m} {mode:7s} wall {wall:7.1f} ms | sweeps {both:7.1f} ms = {flops / both / 1e9 / 78.6:.3f} of peak | "
              f"launches {t['syrk_launches']:.0f} sum {t['syrk_ms']:.1f} ms | diff pred {dp:.1e} err {de:.1e}", flush=True)
