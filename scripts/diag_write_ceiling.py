"""What a pure write stream reaches on this box, next to K1 / K2 (which write 8 bytes per entry and nothing else to HBM):
torch's fill kernel over 6.4 GB (= Sigma's packed panels at N = 40 000) and over 2.83 GB (= the right-hand-side rows), then
scripts/time_assembly.py's measurement of K1 / K2 in the same process.

    python scripts/diag_write_ceiling.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda", 0)
for gb in (6.40016, 2.82656):
    n = int(gb * 1e9 / 8)
    x = torch.empty(n, dtype=torch.float64, device=dev)
    ms = []
    for rep in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        x.fill_(1.5 + rep)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    med = ms[len(ms) // 2]
    print(f"fill of {gb:.2f} GB: median {med:.3f} ms = {gb / med:.2f} TB/s = {gb / med / 8:.3f} of 8 TB/s (min {ms[0]:.3f} ms)", flush=True)
    del x
torch.cuda.empty_cache()
import runpy
sys.argv = [sys.argv[0]]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_assembly.py"), run_name="__main__")
