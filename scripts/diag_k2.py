"""Diagnostic: worklist counts per assembly and bit-identity of the sweeps after the K2 change."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

def mk(pb, **opts):
    pv = pb["params"]
    h = native.Handle(0)
    for k, v in opts.items():
        h.set_option(k, v)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(pb["metric"])
    for k in range(2):
        h.set_data(k, pb["coords"][k], pb["values"][k])
    return h

pb = synth.conus_problem(700, params=synth.SET_A, seed=5)
pc = pb["pcoords"][::9]
h = mk(pb, site_order=1)
h.assemble_joint(); print("K1 fallbacks", h.table_fallbacks())
assert h.factor() == 0
p0 = h.predict(0, pc); print("K2(0) fallbacks", h.table_fallbacks())
p1 = h.predict(1, pc); print("K2(1) fallbacks", h.table_fallbacks())
p0b = h.predict(0, pc); print("K2(0) again fallbacks", h.table_fallbacks(), "same bits", np.array_equal(p0[0], p0b[0]))
cv = h.loocv(0, 700)
h.assemble_joint(); print("K1 again fallbacks", h.table_fallbacks())
assert h.factor() == 0
p0c = h.predict(0, pc); print("K2(0) after refactor fallbacks", h.table_fallbacks(), "same bits", np.array_equal(p0[0], p0c[0]), float(np.max(np.abs(p0[0] - p0c[0]))))
h.set_option("site_order", 0)
h.assemble_joint(); print("K1 site_order 0 fallbacks", h.table_fallbacks())
assert h.factor() == 0
p0d = h.predict(0, pc); print("K2(0) site_order 0 fallbacks", h.table_fallbacks(), "max diff", float(np.max(np.abs(p0[0] - p0d[0]))))
h2 = mk(pb, site_order=0)
h2.assemble_joint(); print("fresh site_order 0: K1 fallbacks", h2.table_fallbacks())
assert h2.factor() == 0
q0 = h2.predict(0, pc); print("fresh K2(0) fallbacks", h2.table_fallbacks(), "max diff vs Hilbert", float(np.max(np.abs(p0[0] - q0[0]))), "vs relaid", float(np.max(np.abs(p0d[0] - q0[0]))))

pb = synth.conus_problem(5000, seed=20003)
pc = pb["pcoords"][:3000]
a = mk(pb); a.assemble_joint(); info, ap, ae = a.factor_predict(0, pc)
b = mk(pb); b.assemble_joint(); assert b.factor() == 0; bp, be = b.predict(0, pc)
c = mk(pb, tall_sweep=0); c.assemble_joint(); info, cp, ce = c.factor_predict(0, pc)
print("N=10000 m=3000: tall vs seq: pred differ at", int((ap != bp).sum()), "err differ at", int((ae != be).sum()), "| two vs seq:", int((cp != bp).sum()), int((ce != be).sum()),
      "| fallbacks", a.table_fallbacks(), b.table_fallbacks(), c.table_fallbacks())
d = mk(pb); d.assemble_joint(); assert d.factor() == 0; dp, de = d.predict(0, pc)
print("seq vs seq (two fresh handles): pred differ at", int((dp != bp).sum()), "err", int((de != be).sum()))
