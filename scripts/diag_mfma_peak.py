"""GPU diagnostic: FP64 MFMA issue-rate ceiling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sif_xco2_cokriging_amd import native
h = native.Handle(0)
for w in (1, 2, 4):
    print(f"waves/SIMD={w}: {h.mfma_peak(w, 20000):.1f} TFLOP/s (repeat {h.mfma_peak(w, 20000):.1f})")
