"""world_size-2 (3, 4 and 8) runs of the multi-process joint predictor on the gloo backend (CPU):
covers panel ownership, the broadcast schedule, prediction-point sharding, the result gather
and the not-positive-definite path of sif-xco2-cokriging_amd/distributed.py.  The panel
arithmetic is the numpy stand-in of tests/fake_panel_handle.py (the HIP kernels need a GPU and
are covered by the -m gpu tests, including the same driver at world = 1)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cokrige_oracle as orc
from tests.conftest import load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sif_xco2_cokriging_amd.distributed import DistributedJoint
        from tests.fake_panel_handle import FakePanelHandle
        if case.startswith("solve"):
            g = load_golden("joint_solve")
            h = FakePanelHandle(g["params_A"], [g["coords0_A"], g["coords1_A"]], [g["values0_A"], g["values1_A"]], 0)
            tok = case.split("_")
            exchange = "p2p" if "p2p" in tok else "sag" if "sag" in tok else "auto" if "auto" in tok else "broadcast"
            group = 3 if "g3" in tok else 2 if "g2" in tok else "auto" if "gauto" in tok else 1
            if "cooptimeout" in tok or "rankfails" in tok:
                # ONE rank's handle reports an error that the others do not see (include/cokrige.h: ck_factor_info fails on
                # the rank whose cooperative panel step timed out, once; "rankfails": some other local failure)
                from sif_xco2_cokriging_amd.native import NativeError
                real, state = h.factor_info, {"n": 0}

                def flaky():
                    state["n"] += 1
                    if rank == world - 1 and (state["n"] == 1 or "rankfails" in tok):
                        raise NativeError("cooperative panel step timed out waiting for a pivot block (option panel_fused "
                                          "bit 4 now off): sweep again" if "cooptimeout" in tok else "hipErrorOutOfMemory")
                    return real()
                h.factor_info = flaky
            r = DistributedJoint(h, rank, world, dist_module=dist, lookahead="sequential" not in tok,
                                 exchange=exchange, panel_group=group, rehearse_collectives="rehearse" in tok
                                 ).prepare(len(g["pcoords_A"]))
            if "rankfails" in tok:
                from sif_xco2_cokriging_amd.native import NativeError
                try:
                    r.predict(1, g["pcoords_A"])
                    q.put((rank, "no-raise", None, None))
                except NativeError as e:
                    q.put((rank, "raised: " + str(e).split("\n")[0], None, None))
                return
            if exchange == "auto":
                info = r.calibrate(reps=1, candidates=("broadcast", "sag", "p2p"))
                assert info["exchange"] in ("broadcast", "sag", "p2p") and r.exchange == info["exchange"]
                assert all(info["calibration_ms"][k] is not None for k in ("broadcast", "sag", "p2p")), info   # all ran
            if group == "auto":
                pred, err = r.autotune(1, g["pcoords_A"])
                assert r.G in (1, 3) and r.tune_info["pass_ms"] is not None
            pred, err = r.predict(1, g["pcoords_A"])
            if "again" in tok:     # the factor is resident: a second set of sites costs one solve-only sweep
                p2, e2 = r.predict(0, g["pcoords_A"][::-1], reuse_factor=True)
                q.put((rank, "ok", pred, err, p2[::-1], e2[::-1]))
            else:
                q.put((rank, "ok", pred, err))
        elif case == "vario":
            from sif_xco2_cokriging_amd.distributed import DistributedVariogram
            from tests.fake_panel_handle import FakeVarioHandle
            g = load_golden("variogram")
            dv = DistributedVariogram(FakeVarioHandle(0), rank, world, dist_module=dist)
            out = []
            for (ci, vi, cj, vj, same) in ((g["coords0"], g["values0"], None, None, True),
                                           (g["coords0"], g["values0"], g["coords1"], g["values1"], False)):
                out.append(dv.variogram_arrays(ci, vi, cj, vj, same, 1500.0, 30))
            q.put((rank, "ok", out, None))
        else:
            g = load_golden("joint_not_pd")
            h = FakePanelHandle(g["params"], [g["coords0"], g["coords1"]], [np.zeros(260), np.zeros(260)], 0)
            r = DistributedJoint(h, rank, world, dist_module=dist).prepare(5)
            try:
                r.predict(0, g["coords0"][:5])
                q.put((rank, "no-raise", None, None))
            except np.linalg.LinAlgError as e:
                q.put((rank, str(e), None, None))
    finally:
        dist.destroy_process_group()


def _run(world, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
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
            assert time.time() - t0 < 180, "timeout"
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out, key=lambda t: t[0])


@pytest.mark.parametrize("world,case", [(2, "solve"), (3, "solve"), (2, "solve_sequential"), (3, "solve_p2p"),
                                        (4, "solve_p2p"), (3, "solve_p2p_sequential"),
                                        (8, "solve"), (8, "solve_p2p"),   # the node's size: more ranks than panels, empty pieces
                                        (2, "solve_sag"), (3, "solve_sag_g3"), (8, "solve_sag"), (3, "solve_auto"),
                                        (2, "solve_g3"), (3, "solve_g2"), (4, "solve_p2p_g3"), (8, "solve_g3"),
                                        (2, "solve_gauto"), (3, "solve_g3_again"), (2, "solve_again")])
def test_joint_predict_two_ranks(world, case):
    """look-ahead schedule (asynchronous broadcast of panel K + 1 under the update by panel K) and the
    plain factor -> broadcast -> apply sequence; the panel exchange as one broadcast, as scatter + in-place all-gather
    ("sag") or as scatter + point-to-point all-gather ("p2p": every rank forwards its piece to all the others), chosen
    by calibration ("auto"); the grouped schedule (g2 / g3: trailing updates for 2 / 3 panels at once with the next
    group's panel steps and exchanges in between; gauto: timed against the per-panel one); a second prediction on the
    resident factor (again)"""
    g = load_golden("joint_solve")
    out = _run(world, case)
    for rank, status, pred, err, *more in out:
        assert status == "ok"
        assert np.max(np.abs(pred - g["pred_A_1"])) / np.max(np.abs(g["pred_A_1"])) < 1e-9
        assert np.max(np.abs(err ** 2 - g["pred_err_A_1"] ** 2)) < 1e-10
        if more:
            assert np.max(np.abs(more[0] - g["pred_A_0"])) / np.max(np.abs(g["pred_A_0"])) < 1e-9
            assert np.max(np.abs(more[1] ** 2 - g["pred_err_A_0"] ** 2)) < 1e-10
    # every rank returns the same full-length vectors
    assert np.array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("world,case", [(2, "solve_cooptimeout"), (3, "solve_g3_cooptimeout"), (1, "solve_rehearse"),
                                        (1, "solve_sag_rehearse_g3"), (1, "solve_p2p_rehearse"), (1, "solve_auto_rehearse")])
def test_one_ranks_timeout_is_swept_again_by_all_and_world1_rehearses_the_collectives(world, case):
    """(a) ADVICE r03: a cooperative panel step that timed out on ONE rank used to raise there in front of the result
    gather while the other ranks blocked in it; the outcome now travels in the gathered buffer and every rank repeats the
    pass.  (b) rehearse_collectives: a one-rank group issues every collective of the schedules (what a single GPU can run
    of the nccl path; here on gloo)."""
    g = load_golden("joint_solve")
    out = _run(world, case)
    assert len(out) == world
    for rank, status, pred, err, *more in out:
        assert status == "ok"
        assert np.max(np.abs(pred - g["pred_A_1"])) / np.max(np.abs(g["pred_A_1"])) < 1e-9
        assert np.max(np.abs(err ** 2 - g["pred_err_A_1"] ** 2)) < 1e-10


def test_a_local_failure_on_one_rank_raises_on_every_rank_instead_of_deadlocking():
    out = _run(3, "solve_rankfails")
    assert [st for _, st, *_ in out][2] == "raised: hipErrorOutOfMemory"
    for rank, status, *_ in out[:2]:
        assert status.startswith("raised: rank(s) [2] failed"), status


def test_not_positive_definite_all_ranks_raise():
    g = load_golden("joint_not_pd")
    out = _run(2, "npd")
    for rank, status, *_ in out:
        assert status == str(g["message"]), status


def test_single_rank_fake_matches_oracle():
    """world = 1 through the same driver, no process group."""
    from sif_xco2_cokriging_amd.distributed import DistributedJoint
    from tests.fake_panel_handle import FakePanelHandle
    g = load_golden("joint_solve")
    h = FakePanelHandle(g["params_R"], [g["coords0_R"], g["coords1_R"]], [g["values0_R"], g["values1_R"]], 0)
    pred, err = DistributedJoint(h, 0, 1).prepare(120).predict(0, g["pcoords_R"])
    assert np.max(np.abs(pred - g["pred_R_0"])) / np.max(np.abs(g["pred_R_0"])) < 1e-9


def test_sharded_variogram_two_and_three_ranks():
    """DistributedVariogram: pair tiles dealt out over the ranks, MIN/MAX all-reduce of the extreme
    distances, SUM all-reduce of the per-bin sums and counts -- against the reference's fixture."""
    g = load_golden("variogram")
    for world in (2, 3):
        out = _run(world, "vario")
        for rank, status, res, _ in out:
            assert status == "ok"
            for (i, j), (centers, edges, means, counts) in zip(((0, 0), (0, 1)), res):
                key = f"semi_1500_30_{i}{j}"
                assert np.array_equal(counts, g[key + "_counts"])
                np.testing.assert_allclose(centers, g[key + "_centers"], rtol=1e-12)
                np.testing.assert_allclose(means, g[key + "_means"], rtol=1e-11, atol=1e-14)
