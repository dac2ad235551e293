"""compact view of bench_local.py output lines (stdin)"""
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        r = json.loads(l)
        print(f'{r["max_dist_km"]:6.0f} km  k_max {r["info"]["k_max"]:5d}  wall {1e3 * r["seconds"]:8.2f} ms  device {r["device_ms"]:8.2f} ms  {r["points_per_s"]:10.0f} pts/s')
