"""phase profile of k_potrf64 (64 x 64 Cholesky + inverse): where the 20-odd microseconds of the panel chain's link go"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sif_xco2_cokriging_amd import native
h = native.Handle(0)
p = h.potrf_profile(300)
print({k: round(v, 2) for k, v in p.items()})
for rows in (2048, 16384):
    o = h.coop_profile(rows)
    print(f"k_panel_coop on {rows} rows: launch {o[0]:.1f} us; per link (us after the pivot's rows flag): accumulated | inverse flag | solved | rows flag set | "
          "diag handed over | factored | flag set")
    for b in range(1, 8):
        print("  link", b, " ".join(f"{o[8 * b + k]:7.2f}" for k in range(1, 8)))   # (stamps of different links come from different XCDs' clocks)
