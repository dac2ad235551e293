"""What the step-wise form of a pass (distributed.DistributedJoint: per-panel ctypes calls from Python, one stream, the form every
rank of a multi-GPU run executes) costs against the single-process form (ck_factor_predict) on ONE GPU, world = 1, no collectives:

    python scripts/bench_stepwise.py [n_obs=20000] [reps=3]

Bounds the per-rank efficiency of the multi-GPU form from above: its kernels per panel are serialised on one stream (the chain
does not run under the bulk of the previous group as in tall_sweeps) and the host issues >= 5 calls per panel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sif_xco2_cokriging_amd import native, synth
from sif_xco2_cokriging_amd.distributed import DistributedJoint

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pb = synth.conus_problem(n, seed=20003)
pv, pc = pb["params"], pb["pcoords"]


def handle():
    h = native.Handle(0)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(pb["metric"])
    for k in range(2):
        h.set_data(k, pb["coords"][k], pb["values"][k])
    return h


h = handle()
ref = None
for r in range(reps + 1):
    h.assemble_joint()
    h.synchronize()
    t0 = time.perf_counter()
    info, pred, err = h.factor_predict(0, pc)
    dt = (time.perf_counter() - t0) * 1e3
    ref = (pred, err)
    if r:
        print(f"N={2 * n} single process (ck_factor_predict, after ck_assemble_joint): {dt:8.2f} ms", flush=True)
h.close()
for G in (1, "1 + chain stream", 3, 4):
    h = handle()
    cs = isinstance(G, str)
    run = DistributedJoint(h, 0, 1, device=torch.device("cuda", 0), panel_group=1 if cs else G, chain_stream=cs).prepare(len(pc))
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pred, err = run.predict(0, pc)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if r:
            d = max(np.max(np.abs(pred - ref[0])) / np.max(np.abs(ref[0])), np.max(np.abs(err - ref[1])) / np.max(np.abs(ref[1])))
            t = run.timings
            print(f"N={2 * n} step-wise G={G} (assembly included): {dt:8.2f} ms  | assemble {t.get('assemble_ms', 0):.2f} panel {t.get('panel_ms', 0):.1f} "
                  f"update {t.get('update_ms', 0):.1f} wait {t.get('bcast_wait_ms', 0):.2f} finish {t.get('finish_ms', 0):.2f} | diff {d:.1e}", flush=True)
    h.close()
