#!/usr/bin/env python3
"""Where a timed step's wall time goes that no kernel accounts for, from a rocprofv3 --kernel-trace CSV of bench.py:

    python scripts/diag_step_gaps.py <kernel_trace.csv> [steps=5]

Takes the last `steps` steps (a step = from one Sigma assembly to the next), prints per step its period, the time some kernel is
running, and every stretch > 5 us with NO kernel running together with the kernels on either side of it."""
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        if r["Kind"] != "KERNEL_DISPATCH":
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("void k_assemble<") and ", false>" in r[2]]
# only the timed steps: ck_factor_predict's passes carry k_tall_group_d
starts = [i for k, i in enumerate(starts[:-1]) if any("k_tall_group_d" in r[2] for r in rows[i:starts[k + 1]])]
for a, b in list(zip(starts[:-1], starts[1:]))[-steps:]:
    P = rows[a:b]
    t0, t1 = P[0][0], rows[b][0]
    cur, busy, holes = t0, 0, []
    last = P[0][2]
    for s, e, n in P:
        if s > cur:
            if s - cur > 5000:
                holes.append((cur - t0, s - cur, last, n))
            cur = s
        if e > cur:
            busy += e - cur
            cur = e
            last = n
    if t1 - cur > 5000:
        holes.append((cur - t0, t1 - cur, last, "(next step's first kernel)"))
    print(f"step period {(t1 - t0) / 1e6:8.3f} ms, some kernel running {busy / 1e6:8.3f} ms, idle {(t1 - t0 - busy) / 1e6:6.3f} ms in {len(holes)} stretches > 5 us:")
    for at, d, l, n in holes:
        print(f"    at {at / 1e6:8.3f} ms: {d / 1e3:7.1f} us   after {l}  before {n}")
