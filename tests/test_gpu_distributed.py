"""The multi-rank HIP code path on ONE GPU: two processes share cuda:0, each with its own
handle partitioned (rank, world = 2) -- block-column-cyclic ownership, receive buffers, the
strided trailing-update launch -- with the panel exchange done over gloo (RCCL refuses two
ranks on one device; the driver's 8-GPU run uses the nccl backend through the same code)."""
import os
import socket

import numpy as np
import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


class _GlooViaHost:
    """torch.distributed facade that stages device tensors through the host for gloo."""

    def __init__(self, dist):
        self.d = dist

    def broadcast(self, t, src, group=None, async_op=False):
        c = t.cpu()
        self.d.broadcast(c, src=src)
        t.copy_(c)
        if async_op:
            class _Done:
                def wait(self):
                    return True
            return _Done()

    def all_gather(self, outs, t, group=None):
        cs = [o.cpu() for o in outs]
        self.d.all_gather(cs, t.cpu())
        for o, c in zip(outs, cs):
            o.copy_(c)

    # point-to-point exchange (exchange="p2p"): gloo sends and receives host tensors only
    isend, irecv = "isend", "irecv"

    class P2POp:
        def __init__(self, op, tensor, peer, group=None):
            self.op, self.tensor, self.peer = op, tensor, peer

    def get_backend(self, group=None):
        return "gloo"

    def batch_isend_irecv(self, ops):
        import torch
        d = self.d
        works = []
        for o in ops:
            if o.op == "isend":
                torch.cuda.synchronize()
                hostbuf = o.tensor.cpu().contiguous()
                w = d.isend(hostbuf, o.peer)
                works.append(_HostWork(w, None, hostbuf))
            else:
                hostbuf = torch.empty(o.tensor.shape, dtype=o.tensor.dtype)
                w = d.irecv(hostbuf, o.peer)
                works.append(_HostWork(w, o.tensor, hostbuf))
        return works


class _HostWork:
    def __init__(self, w, dev_tensor, hostbuf):
        self.w, self.dev, self.hostbuf = w, dev_tensor, hostbuf

    def wait(self):
        self.w.wait()
        if self.dev is not None:
            self.dev.copy_(self.hostbuf)
            self.dev = None
        return True


def _worker(rank, world, port, q, via_host, big=False, exchange="broadcast"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sif_xco2_cokriging_amd import native
        from sif_xco2_cokriging_amd.distributed import DistributedJoint
        g = load_golden("joint_solve")
        # enlarge the problem beyond one 512-panel per rank: 3 panels
        rng = np.random.default_rng(11)
        npts = 4300 if big else 1300         # big: N = 4 600 -> 9 panels, look-ahead over several owners' turns
        lat = rng.uniform(25, 50, npts)
        lon = rng.uniform(-120, -70, npts)
        pts = np.column_stack([lat, lon])
        n0 = 2300 if big else 700
        coords = [pts[:n0], pts[n0 - 200:npts]]
        values = [rng.standard_normal(len(coords[0])), rng.standard_normal(len(coords[1]))]
        pv = g["params_A"]
        h = native.Handle(0)
        h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        h.set_metric(0)
        for k in range(2):
            h.set_data(k, coords[k], values[k])
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        dm = _GlooViaHost(dist) if via_host else dist   # raw gloo on device tensors: really asynchronous broadcasts
        r = DistributedJoint(h, rank, world, dist_module=dm, device=dev, exchange=exchange).prepare(len(g["pcoords_A"]))
        pred, err = r.predict(0, g["pcoords_A"])
        assert r.timings["update_ms"] > 0 and r.timings["bcast_wait_ms"] >= 0
        q.put((rank, pred, err, coords, values))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("via_host,big,world,exchange", [(True, False, 2, "broadcast"), (False, False, 2, "broadcast"),
                                                         (False, True, 2, "broadcast"), (True, True, 3, "p2p")])
def test_two_ranks_one_gpu_matches_oracle(via_host, big, world, exchange):
    """(the last case: three ranks, every panel scattered by its owner and passed on point to point)"""
    import torch.multiprocessing as mp
    from oracle import cokrige_oracle as orc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, via_host, big, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _queue
    import time
    out, t0 = [], time.time()
    while len(out) < world:
        try:
            out.append(q.get(timeout=2))
        except _queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), "a rank died"
            assert time.time() - t0 < 240, "timeout"
    out.sort(key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = load_golden("joint_solve")
    _, pred0, err0, coords, values = out[0]
    p = orc.Params.from_flat(g["params_A"])
    rp, re = orc.joint_predict(p, coords, values, g["pcoords_A"], 0, 0)
    assert np.max(np.abs(pred0 - rp)) / np.max(np.abs(rp)) < 1e-9
    assert np.max(np.abs(err0 ** 2 - re ** 2)) < 1e-10
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


def test_world1_driver_with_arena_matches_direct():
    """the same driver at world = 1 with a torch-owned arena == ck_factor + ck_predict."""
    import torch
    from sif_xco2_cokriging_amd import native
    from sif_xco2_cokriging_amd.distributed import DistributedJoint
    g = load_golden("joint_solve")
    pv = g["params_R"]

    def mk():
        h = native.Handle(0)
        h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        h.set_metric(0)
        h.set_data(0, g["coords0_R"], g["values0_R"])
        h.set_data(1, g["coords1_R"], g["values1_R"])
        return h
    h1 = mk()
    r = DistributedJoint(h1, 0, 1, device=torch.device("cuda", 0)).prepare(len(g["pcoords_R"]))
    p1, e1 = r.predict(1, g["pcoords_R"])
    h2 = mk()
    h2.assemble_joint()
    assert h2.factor() == 0
    p2, e2 = h2.predict(1, g["pcoords_R"])
    assert np.array_equal(p1, p2) and np.array_equal(e1, e2)
    assert np.max(np.abs(p1 - g["pred_R_1"])) / np.max(np.abs(g["pred_R_1"])) < 1e-9


def test_step_wise_driver_reports_the_failing_minor_in_the_callers_order():
    """A Sigma that is not positive definite: the step-wise driver (the form every rank of a multi-GPU run executes)
    raises scipy's message with the minor index in the CALLER's site order -- it sweeps a second time with
    site_order = 0, as ck_factor does for one process -- not in the library's Hilbert order."""
    import torch
    from numpy.linalg import LinAlgError
    from sif_xco2_cokriging_amd import native
    from sif_xco2_cokriging_amd.distributed import DistributedJoint
    g = load_golden("joint_not_pd")
    pv = g["params"]
    h = native.Handle(0)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(0)
    h.set_data(0, g["coords0"], np.zeros(len(g["coords0"])))
    h.set_data(1, g["coords1"], np.zeros(len(g["coords1"])))
    r = DistributedJoint(h, 0, 1, device=torch.device("cuda", 0)).prepare(5)
    with pytest.raises(LinAlgError) as e:
        r.predict(0, g["coords0"][:5])
    assert str(e.value) == str(g["message"])


def _worker_vario_local(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sif_xco2_cokriging_amd import native
        from sif_xco2_cokriging_amd.distributed import DistributedLocal, DistributedVariogram
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        g = load_golden("variogram")
        h = native.Handle(0)
        h.set_metric(0)
        dv = DistributedVariogram(h, rank, world, dist_module=dist, device=dev)
        vres = []
        for (ci, vi, cj, vj, same) in ((g["coords0"], g["values0"], None, None, True),
                                       (g["coords0"], g["values0"], g["coords1"], g["values1"], False)):
            vres.append(dv.variogram_arrays(ci, vi, cj, vj, same, 1500.0, 30))
        h.close()
        # local predictor, points sharded
        s = load_golden("joint_solve")
        pv = s["params_A"]
        h2 = native.Handle(0)
        h2.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        h2.set_metric(0)
        h2.set_data(0, s["coords0_A"], s["values0_A"])
        h2.set_data(1, s["coords1_A"], s["values1_A"])
        pred, err = DistributedLocal(h2, rank, world, dist_module=dist, device=dev).predict(1, s["pcoords_A"], max_dist=600.0)
        ref = h2.predict_local(1, s["pcoords_A"], 600.0)[:2] if rank == 0 else None
        h2.close()
        q.put((rank, vres, pred, err, ref))
    finally:
        dist.destroy_process_group()


def test_sharded_variogram_and_local_predictor_two_ranks_one_gpu():
    """the lag-binning kernels with their pair tiles dealt out over two ranks, and the
    local predictor with its points sharded -- the real HIP kernels, two processes on one GPU over gloo."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_vario_local, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue as _queue
    import time
    out, t0 = [], time.time()
    while len(out) < 2:
        try:
            out.append(q.get(timeout=2))
        except _queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), "a rank died"
            assert time.time() - t0 < 240, "timeout"
    out.sort(key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = load_golden("variogram")
    for rank, vres, pred, err, ref in out:
        for (i, j), (centers, edges, means, counts) in zip(((0, 0), (0, 1)), vres):
            key = f"semi_1500_30_{i}{j}"
            assert np.array_equal(counts, g[key + "_counts"])
            np.testing.assert_allclose(centers, g[key + "_centers"], rtol=1e-12)
            np.testing.assert_allclose(means, g[key + "_means"], rtol=1e-11, atol=1e-14)
    ref_pred, ref_err = out[0][4]
    for rank, vres, pred, err, ref in out:
        assert np.array_equal(np.isnan(pred), np.isnan(ref_pred))
        np.testing.assert_allclose(pred, ref_pred, rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(err, ref_err, rtol=1e-12, equal_nan=True)
