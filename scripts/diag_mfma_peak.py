"""GPU diagnostic: FP64 MFMA issue-rate ceiling and the clock the chip holds under it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sif_xco2_cokriging_amd import native
h = native.Handle(0)
for w in (1, 2, 4):
    for rep in range(2):
        print(f"waves/SIMD={w}:", h.mfma_peak(w, 40000))
