"""phase profile of k_potrf64 (64 x 64 Cholesky + inverse): where the 20-odd microseconds of the panel chain's link go"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sif_xco2_cokriging_amd import native
h = native.Handle(0)
p = h.potrf_profile(300)
print({k: round(v, 2) for k, v in p.items()})
