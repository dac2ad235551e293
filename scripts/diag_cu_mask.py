"""GPU diagnostic: which (XCC, SE, SH, CU) the workgroups of CU-masked streams land on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from collections import Counter
from sif_xco2_cokriging_amd import native

h = native.Handle(0)


def show(tag, mask):
    loc = h.cu_probe(mask, 4096)
    per_xcc = Counter(int(x) for x in loc[:, 0])
    cus = sorted(set(tuple(int(v) for v in r) for r in loc))
    print(f"{tag}: distinct CUs {len(cus)}; workgroups per XCC {dict(sorted(per_xcc.items()))}", flush=True)
    if len(cus) <= 40:
        print("   ", cus, flush=True)


show("unmasked", None)
m = np.zeros(8, dtype=np.uint32)
m[0] = 0xFF               # bits 0..7
show("bits 0-7", m)
m = np.full(8, 0xFFFFFFFF, dtype=np.uint32)
m[0] = 0xFFFFFF00         # everything but bits 0..7
show("all but bits 0-7", m)
m = np.zeros(8, dtype=np.uint32)
m[0] = 0xFFFF             # bits 0..15
show("bits 0-15", m)
