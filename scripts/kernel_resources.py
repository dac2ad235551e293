#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel of one csrc/*.hip file (device assembly of the gfx950 build).

    python scripts/kernel_resources.py ck_cov.hip [name-filter]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
path = os.path.join(ROOT, "sif-xco2-cokriging_amd", "csrc", src)
out = f"/tmp/asm/{src}.s"
os.makedirs("/tmp/asm", exist_ok=True)
extra = ["-fno-slp-vectorize"] if src == "ck_vario.hip" else []
extra += os.environ.get("CK_BUILD_DEFS", "").split()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "--cuda-device-only",
                "-S", path, "-o", out] + extra, check=True)
txt = open(out).read()
# the metadata block at the end lists every kernel
kern = re.split(r"\n  - \.agpr_count:", txt)
for k in kern[1:]:
    name = re.search(r"\.name:\s+(\S+)", k).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.split("(")[0]
    if flt and flt not in dem:
        continue
    g = lambda key: (re.search(rf"\.{key}:\s+(\d+)", k) or [None, "?"])[1]
    print(f"{dem:55s} vgpr {g('vgpr_count'):>4s} agpr {k.split()[0]:>3s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} "
          f"lds {g('group_segment_fixed_size'):>6s} spill_v {g('vgpr_spill_count'):>3s}")
