"""RankPool's reply collection (sif-xco2-cokriging_amd/workers.py) without GPUs: one rank fails while another sits in a
collective it will never leave.  The caller must get the error within the grace period, not after one request timeout per
rank, and close() / garbage collection must not wait for ranks that cannot answer."""
import multiprocessing as mp
import time

import numpy as np
import pytest

from sif_xco2_cokriging_amd import workers
from sif_xco2_cokriging_amd.native import NativeError


def _fake_rank(conn, behaviour):
    """ok: answers every request | err: answers with an error | stuck: never answers (a rank inside a collective)"""
    conn.send(("ok", None))                      # start-up handshake
    while True:
        try:
            msg = conn.recv()
        except EOFError:
            return
        if behaviour == "stuck":
            time.sleep(3600)
        if msg[0] == "close":
            conn.send(("ok", None))
            return
        if behaviour == "err":
            conn.send(("err", "NativeError", "hipErrorOutOfMemory\n[rank 1]"))
        elif behaviour == "linalg":
            conn.send(("err", "LinAlgError", "3-th leading minor of the array is not positive definite\n[rank]"))
        else:
            conn.send(("ok", (np.zeros(2), np.ones(2), {}, {})))


def _pool(behaviours, timeout=3600.0):
    ctx = mp.get_context("spawn")
    pool = object.__new__(workers.RankPool)
    pool.devices, pool.world, pool.backend, pool.timeout = list(range(len(behaviours))), len(behaviours), "gloo", timeout
    pool._conns, pool._procs = [], []
    pool.last_timings, pool.last_comm = {}, {}
    for b in behaviours:
        a, c = ctx.Pipe()
        p = ctx.Process(target=_fake_rank, args=(c, b), daemon=True)
        p.start()
        c.close()
        pool._conns.append(a)
        pool._procs.append(p)
    pool._collect(60.0)
    return pool


def test_one_failed_rank_does_not_cost_a_timeout_per_rank(monkeypatch):
    monkeypatch.setenv("CK_RANK_GRACE", "2")
    pool = _pool(["stuck", "err", "stuck"])
    procs = list(pool._procs)
    t0 = time.monotonic()
    with pytest.raises(NativeError) as e:
        pool.predict_joint(0, np.zeros((1, 2)))
    assert "hipErrorOutOfMemory" in str(e.value)
    assert time.monotonic() - t0 < 30.0            # not 3 x 3600 s
    assert pool._procs == [] and all(not p.is_alive() for p in procs)   # the stuck ranks were terminated
    with pytest.raises(workers.RankError):
        pool.predict_joint(0, np.zeros((1, 2)))    # the pool says that it is closed
    t0 = time.monotonic()
    pool.close()
    assert time.monotonic() - t0 < 5.0


def test_errors_raised_by_every_rank_leave_the_pool_usable():
    from numpy.linalg import LinAlgError
    pool = _pool(["linalg", "linalg"])
    with pytest.raises(LinAlgError) as e:
        pool.predict_joint(0, np.zeros((1, 2)))
    assert str(e.value) == "3-th leading minor of the array is not positive definite"
    assert len(pool._procs) == 2 and all(p.is_alive() for p in pool._procs)
    pool.close()
    assert pool._procs == []


def test_close_does_not_wait_for_a_stuck_rank(monkeypatch):
    monkeypatch.setenv("CK_RANK_CLOSE_TIMEOUT", "2")
    pool = _pool(["ok", "stuck"])
    procs = list(pool._procs)
    t0 = time.monotonic()
    pool.close()
    assert time.monotonic() - t0 < 30.0
    assert all(not p.is_alive() for p in procs)


def test_a_dead_rank_is_reported_at_once():
    pool = _pool(["ok", "ok"])
    pool._procs[1].terminate()
    pool._procs[1].join(10)
    with pytest.raises(workers.RankError) as e:
        pool.predict_joint(0, np.zeros((1, 2)))
    assert "rank 1" in str(e.value)
    assert pool._procs == []
