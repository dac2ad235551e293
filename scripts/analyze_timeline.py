#!/usr/bin/env python3
"""Timeline of one pass from a rocprofv3 --kernel-trace CSV: who occupies the chip, when.

    python scripts/analyze_timeline.py <kernel_trace.csv> [first_kernel_substring]

Takes the LAST pass in the trace (from the last k_assemble<..., false> = the assembly of Sigma to the last k_reduce_pred),
prints per kernel: launches, sum of durations, union of its busy intervals; for the whole pass: span, time with NO kernel
running (gaps), time with only small (< 64 workgroups x ...) kernels, and a coarse timeline (20 slices: GEMM-busy share)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        if r["Kind"] != "KERNEL_DISPATCH":
            continue
        name = r["Kernel_Name"].split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1),
                     int(r["Queue_Id"])))
rows.sort()
# the last pass: from the last Sigma assembly to the last reduction
starts = [i for i, r in enumerate(rows) if "k_assemble<true, false>" in r[2] or "k_assemble<false, false>" in r[2]]
ends = [i for i, r in enumerate(rows) if "k_reduce_pred" in r[2]]
i0 = starts[-1] if starts else 0
i1 = [e for e in ends if e > i0][-1] if [e for e in ends if e > i0] else len(rows) - 1
if not [e for e in ends if e > i0] and len(starts) > 1:
    i0 = starts[-2]
    i1 = [e for e in ends if e > i0][-1]
P = rows[i0:i1 + 1]
t0, t1 = P[0][0], max(r[1] for r in P)
span = (t1 - t0) / 1e6


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


by = defaultdict(list)
for s, e, n, wg, q in P:
    by[n].append((s, e, wg, q))
print(f"pass: {len(P)} dispatches, span {span:.2f} ms")
print(f"{'kernel':60s} {'n':>5s} {'sum ms':>9s} {'union ms':>9s} {'avg ms':>8s} {'queues'}")
for n, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e, _, _ in kv[1])):
    sm = sum(e - s for s, e, _, _ in v) / 1e6
    un = union([(s, e) for s, e, _, _ in v]) / 1e6
    print(f"{n[:60]:60s} {len(v):5d} {sm:9.3f} {un:9.3f} {sm / len(v):8.4f} {sorted(set(q for _, _, _, q in v))}")
busy_any = union([(s, e) for s, e, _, _, _ in P]) / 1e6
big = [(s, e) for s, e, n, wg, _ in P if wg >= 2048]
busy_big = union(big) / 1e6
print(f"some kernel running: {busy_any:.2f} ms of {span:.2f} (idle {span - busy_any:.2f} ms)")
print(f"a chip-filling launch (>= 2048 workgroups) running: {busy_big:.2f} ms; only small launches or nothing: {span - busy_big:.2f} ms")
# where are the stretches without a chip-filling launch?
big.sort()
holes = []
cur = t0
for s, e in big:
    if s > cur:
        holes.append((cur, s))
    cur = max(cur, e)
if cur < t1:
    holes.append((cur, t1))
holes = [(s, e) for s, e in holes if e - s > 20000]
print(f"{len(holes)} stretches > 20 us without a chip-filling launch, total {sum(e - s for s, e in holes) / 1e6:.2f} ms; the ten longest:")
for s, e in sorted(holes, key=lambda h: -(h[1] - h[0]))[:10]:
    inside = sorted(set(n.split('<')[0][:24] for a, b, n, wg, _ in P if a < e and b > s and wg < 2048))
    print(f"   at {(s - t0) / 1e6:8.2f} ms: {(e - s) / 1e3:8.1f} us   running: {', '.join(inside)}")
# optional third argument: list every dispatch of the last so-many ms of the pass (start, duration, workgroups, queue)
if len(sys.argv) > 3:
    tail_ms = float(sys.argv[3])
    print(f"dispatches of the last {tail_ms} ms (and of the first {tail_ms / 2} ms):")
    for s, e, n, wg, q in P:
        if s >= t1 - tail_ms * 1e6 or s <= t0 + tail_ms * 0.5e6:
            print(f"   {(s - t0) / 1e6:9.3f} ms  {(e - s) / 1e3:8.1f} us  {wg:6d} wg  q{q}  {n.split('<')[0][:28]}")
