"""GPU benchmark of the local-neighbourhood predictor (point_prediction path, ck_predict_local):
config-3 sites, full 0.5-degree grid, a sweep of max_dist."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
pb = synth.conus_problem(n)
pv = pb["params"]
h = native.Handle(0)
h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
h.set_metric(0)
for k in range(2):
    h.set_data(k, pb["coords"][k], pb["values"][k])
if os.environ.get("CK_LOCAL_TILE_MIN"):   # A/B of the size-class boundary (default 256)
    h.set_option("local_tile_min", int(os.environ["CK_LOCAL_TILE_MIN"]))
if os.environ.get("CK_LOCAL_SLAB_MB"):
    h.set_option("local_slab_mb", int(os.environ["CK_LOCAL_SLAB_MB"]))
if os.environ.get("CK_LOCAL_GROUP"):
    h.set_option("local_group", int(os.environ["CK_LOCAL_GROUP"]))
out = []
if "--reserve" in sys.argv:     # the scratch slab at the automatic budget, once (ck_local_reserve): no call below grows it
    sys.argv.remove("--reserve")
    t0 = time.perf_counter()
    h.local_reserve(0)
    print(json.dumps({"local_reserve_wall_ms": (time.perf_counter() - t0) * 1e3, "alloc_ms": h.timings()["local_alloc_ms"]}), flush=True)
for md in [float(x) for x in (sys.argv[2:] or ["50", "100", "200", "400"])]:
    h.predict_local(0, pb["pcoords"][:64], md)          # warm-up (layout, tables)
    t0 = time.perf_counter()
    res = h.predict_local(0, pb["pcoords"], md)
    dt = time.perf_counter() - t0
    pred = res[0]
    info = res[2] if len(res) > 2 else {}
    out.append({"max_dist_km": md, "points": len(pred), "seconds": dt, "points_per_s": len(pred) / dt,
                "finite": int(np.isfinite(pred).sum()), "info": {k: int(v) for k, v in dict(info).items()} if info else None})
    out[-1]["checksum"] = float(np.nansum(pred))
    t = h.timings()
    out[-1]["device_ms"] = t["local_ms"]               # device work only: counting pass + solve (ck_timings [10])
    out[-1]["scratch_alloc_ms"] = t["local_alloc_ms"]   # host time this call spent growing the scratch slab (hipMalloc); round 3
                                                        # counted it inside device_ms (the 1 262 ms of r03c's 100 km row)
    print(json.dumps(out[-1]), flush=True)
