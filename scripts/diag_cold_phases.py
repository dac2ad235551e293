"""GPU diagnostic: where does the cold pass (new handle -> results on the host) spend its time beyond the kernels?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
pb = synth.conus_problem(n, seed=20003)
pv = pb["params"]


def run():
    ph = {}
    t = time.perf_counter()

    def lap(name):
        nonlocal t
        now = time.perf_counter()
        ph[name] = (now - t) * 1e3
        t = now
    h = native.Handle(0); lap("create")
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10]); lap("set_model")
    h.set_metric(pb["metric"])
    for k in range(2):
        h.set_data(k, pb["coords"][k], pb["values"][k])
    lap("set_data")
    h.num_panels(); lap("layout (Hilbert sort, uploads, allocation, tables)")
    h.assemble_joint(); lap(f"assemble_joint (K1 on fresh memory: {h.timings()['assemble_sigma_ms']:.2f} ms by events)")
    info, pred, err = h.factor_predict(0, pb["pcoords"]); lap("factor_predict (aux allocation, K2, sweeps, reduce, results)")
    tm = h.timings()
    h.close(); lap("close")
    return ph, tm


run()
for rep in range(3):
    ph, tm = run()
    dev = tm["assemble_sigma_ms"] + tm["assemble_aux_ms"] + tm["fused_sweeps_ms"] + tm["reduce_ms"]
    print(f"rep {rep}: total {sum(ph.values()):.1f} ms, device stages {dev:.1f} ms | " + " | ".join(f"{k} {v:.1f}" for k, v in ph.items()), flush=True)
