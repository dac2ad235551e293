"""A/B of the panel step, variants INTERLEAVED (the clocks sag under sustained FP64 matrix load: a variant measured later
in a process looks slower): 24 dependent launches per panel (rounds 1-2) against the cooperative single launch
(k_panel_coop, option panel_fused bit 4), for several panel groupings.  Results must agree to rounding."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
which = sys.argv[3].split(",") if len(sys.argv) > 3 else None
pb = synth.conus_problem(n, seed=20003)
pv = pb["params"]
pc = pb["pcoords"][:2048]
variants = [("launches", {"panel_fused": 2}), ("launchesG1", {"panel_fused": 2, "panel_group": 1}),
            ("coop", {"panel_fused": 2 | 16}), ("coopG1", {"panel_fused": 2 | 16, "panel_group": 1}),
            ("coopG2", {"panel_fused": 2 | 16, "panel_group": 2}), ("coopG4", {"panel_fused": 2 | 16, "panel_group": 4}),
            ("coopLA", {"panel_fused": 2 | 16, "lookahead": 1}), ("launchesLA", {"panel_fused": 2, "lookahead": 1}),
            ("default", {}), ("LA0", {"lookahead": 0})]
if which:
    variants = [v for v in variants if v[0] in which]
hs = []
for name, opts in variants:
    h = native.Handle(0)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(pb["metric"])
    for k in range(2):
        h.set_data(k, pb["coords"][k], pb["values"][k])
    h.set_option("time_gemm", 1)
    for k, v in opts.items():
        h.set_option(k, v)
    hs.append(h)
fm = {name: [] for name, _ in variants}
sy = {name: [] for name, _ in variants}
for it in range(reps + 1):
    for (name, _), h in zip(variants, hs):
        h.assemble_joint()
        assert h.factor() == 0
        t = h.timings()
        fm[name].append(t["factor_ms"])
        sy[name].append(t["syrk_ms"])
ref = None
for (name, _), h in zip(variants, hs):
    t = h.timings()
    pcs = pb["pcoords"] if os.environ.get("CK_AB_FULL_GRID") else pc
    sv = []
    for _ in range(3):
        pred, err = h.predict(0, pcs)
        sv.append(h.timings()["solve_ms"])
    pred, err = pred[:len(pc)] if len(pcs) != len(pc) else pred, err[:len(pc)] if len(pcs) != len(pc) else err
    print(f"   solve_ms ({len(pcs)} points) {' '.join(f'{x:.2f}' for x in sv)}")
    if ref is None:
        ref = (pred, err)
    dp = np.max(np.abs(pred - ref[0])) / np.max(np.abs(ref[0]))
    de = np.max(np.abs(err - ref[1])) / np.max(np.abs(ref[1]))
    print(f"N={2*n} {name:8s} factor_ms {' '.join(f'{x:.1f}' for x in fm[name])} | min {min(fm[name][1:]):.2f} | syrk_ms {' '.join(f'{x:.1f}' for x in sy[name][1:])} | "
          f"diff pred {dp:.1e} err {de:.1e} | redone {t['panel_coop_redone']:.0f}", flush=True)
    h.close()
