"""GPU diagnostic: how full is the chip during ONE trailing update of the factorisation?

Every workgroup of the selected k_syrk_group_d launch (option gemm_stamps = 2 + K0) leaves its start, its lifetime and the
CU it ran on.  From those: workgroups in flight over time, idle time of the CUs' two slots between workgroups, the
lifetime distribution over the launch, and the time the launch needs against (tiles / 512) x median lifetime."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
K0s = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 36]
pb = synth.conus_problem(n, seed=20003)
pv = pb["params"]
h = native.Handle(0)
h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
h.set_metric(pb["metric"])
for k in range(2):
    h.set_data(k, pb["coords"][k], pb["values"][k])
h.assemble_joint()
Npad = (2 * n + 511) // 512 * 512 if (n % 64 == 0) else None
for K0 in K0s:
    h.set_option("gemm_stamps", 2 + K0)
    for it in range(3):
        h.assemble_joint()
        assert h.factor() == 0
    nwg_max = 4 * 4 * 400 * 400
    raw, grid = h.gemm_stamps(min(nwg_max, 2_000_000))
    gx, gy, J0, npan = [int(x) for x in grid]
    nwg = gx * gy
    st = raw[:nwg].astype(np.int64)
    cyc, ticks, start, hw = st[:, 0], st[:, 1], st[:, 2], st[:, 3]
    ran = ticks > 1000
    came = start > 0
    t0 = start[came].min()
    s_us = (start - t0) / 100.0
    e_us = s_us + ticks / 100.0
    end = e_us[ran].max()
    life = ticks[ran] / 100.0
    print(f"== trailing update behind panels {K0}..{K0+npan-1}: grid {gx} x {gy} = {nwg} workgroups, {ran.sum()} run a tile, "
          f"{came.sum() - ran.sum()} return at once; launch spans {end:.0f} us (first start to last end)")
    print(f"   lifetime us: median {np.median(life):.1f} mean {life.mean():.1f} p05 {np.percentile(life,5):.1f} "
          f"p95 {np.percentile(life,95):.1f} max {life.max():.1f};  tiles/512 x median = {ran.sum()/512*np.median(life):.0f} us, "
          f"sum of lifetimes / 512 = {life.sum()/512:.0f} us")
    # workgroups in flight over time
    ev = np.concatenate([np.stack([s_us[ran], np.ones(ran.sum())], 1), np.stack([e_us[ran], -np.ones(ran.sum())], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    fl = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0], append=ev[-1, 0])
    avg = (fl * dt).sum() / end
    print(f"   workgroups in flight: time-average {avg:.1f} of 512; time with < 480 in flight: {dt[fl < 480].sum():.0f} us, "
          f"< 256: {dt[fl < 256].sum():.0f} us")
    # by tenth of the launch: mean in flight, mean lifetime of the workgroups that started there
    for d in range(10):
        a, b = end * d / 10, end * (d + 1) / 10
        m = (ev[:, 0] >= a) & (ev[:, 0] < b)
        ws = ran & (s_us >= a) & (s_us < b)
        inflight = (fl[m] * dt[m]).sum() / max(dt[m].sum(), 1e-9)
        print(f"     {a:8.0f}-{b:8.0f} us: in flight {inflight:6.1f}  started {ws.sum():6d}  mean lifetime of those "
              f"{(ticks[ws].mean() / 100.0 if ws.any() else 0):.1f} us  returned at once {(came & ~ran & (s_us >= a) & (s_us < b)).sum()}")
    # slots: per CU (xcc, se, sh(0), cu) the gaps between the end of one tile and the start of the next
    xcc = (hw >> 32) & 15
    cu = hw & 15
    se = (hw >> 13) & 7
    sh = (hw >> 12) & 1
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ids = np.unique(cuid[ran])
    print(f"   CUs seen: {len(ids)}")
    busy = 0.0
    gaps = []
    for c in ids[:: max(1, len(ids) // 64)]:
        m = ran & (cuid == c)
        o = np.argsort(s_us[m])
        ss, ee = s_us[m][o], e_us[m][o]
        # two slots: assign greedily
        slot_end = [0.0, 0.0]
        for a, b in zip(ss, ee):
            k = int(np.argmin(slot_end)) if min(slot_end) <= a + 1e-6 else int(np.argmin(slot_end))
            gaps.append(a - slot_end[k])
            slot_end[k] = b
    gaps = np.array(gaps)
    print(f"   slot gaps (end of a tile -> start of the next tile in that slot), sampled CUs: median {np.median(gaps):.1f} us, "
          f"mean {gaps.mean():.1f}, p95 {np.percentile(gaps,95):.1f}, share of slot time {gaps.sum() / (len(ids[::max(1,len(ids)//64)]) * 2 * end):.3f}")
h.close()
