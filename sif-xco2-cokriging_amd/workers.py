"""Multi-GPU entry of the drop-in boundary: ``Predictor(mod, mf, devices=[0, 1, ...])``.

The reference's one parallel entry is a keyword on the predictor (``partitions`` ->
``multiprocessing.Pool``, src/point_prediction.py:45-52, 69-81).  Here the parallel resource is
the node's GPUs: ``devices`` starts one FRESH worker process per entry (spawn: a new interpreter
that has never touched a GPU -- the calling process may long have), each worker owns one GPU, one
``native.Handle`` partitioned (rank, world) and one rank of a ``torch.distributed`` group
(backend "nccl" = RCCL over xGMI when every rank has its own GPU; "gloo" when two ranks share a
device -- the single-GPU rehearsal of the same code).  The workers stay alive between calls:
model and data are shipped once per change, the factor stays resident in the ranks' panels, and
further predictions cost one exchange-and-substitute sweep (``DistributedJoint.predict(...,
reuse_factor=True)``).

    pool = RankPool([0, 1, 2, 3])
    pool.load(model_arrays, metric, coords, values)
    pred, err = pool.predict_joint(i, pcoords)
    pool.close()

Every rank computes the full-length result; rank 0's copy is returned.  An exception in a rank
(``numpy.linalg.LinAlgError`` for a Sigma that is not positive definite, ``NativeError``, ...)
is re-raised in the caller with the same type where the type is known.
"""
from __future__ import annotations

import os
import socket
import traceback

import numpy as np


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def pick_backend(devices):
    """"nccl" (RCCL) when every rank has a GPU of its own, "gloo" when ranks share one; CK_DIST_BACKEND overrides."""
    env = os.environ.get("CK_DIST_BACKEND")
    if env:
        return env
    return "nccl" if len(set(devices)) == len(devices) and len(devices) > 1 else "gloo"


def _rank_main(rank, world, device, port, backend, conn, opts):
    """Body of one worker process (one rank, one GPU)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist = None
    try:
        import torch
        torch.cuda.set_device(device)
        dev = torch.device("cuda", device)
        if world > 1:
            import torch.distributed as dist
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        from . import native
        from .distributed import DistributedJoint, DistributedLocal
        state = {"h": None, "hl": None, "runner": None, "m_cap": -1, "loaded": None}

        def new_handle():
            ld = state["loaded"]
            h = native.Handle(devices=opts["devices"], rank=rank)      # ck_create_partitioned: device = devices[rank]
            h.set_model(*ld["model"])
            h.set_metric(ld["metric"])
            for k, (c, v) in enumerate(zip(ld["coords"], ld["values"])):
                h.set_data(k, c, v)
            return h

        def drop():
            if state["h"] is not None:
                state["h"].close()
            state.update(h=None, runner=None, m_cap=-1)
            if state.get("hl") is not None:
                state["hl"].close()
            state["hl"] = None

        conn.send(("ok", None))
        while True:
            msg = conn.recv()
            op = msg[0]
            try:
                if op == "close":
                    drop()
                    conn.send(("ok", None))
                    break
                if op == "load":
                    drop()
                    state["loaded"] = msg[1]
                    conn.send(("ok", None))
                elif op == "predict_joint":
                    _, i, pcoords, reuse = msg
                    m = len(pcoords)
                    # the arena is sized for a number of prediction points: a larger set needs a new handle (and factor)
                    if state["runner"] is None or m > state["m_cap"]:
                        if state["h"] is not None:
                            state["h"].close()
                        h = new_handle()
                        r = DistributedJoint(h, rank, world, dist_module=dist, device=dev,
                                             exchange=opts.get("exchange", "broadcast") if backend == "nccl" else "broadcast",
                                             panel_group=opts.get("panel_group", 1)).prepare(m)
                        r.calibrate()
                        state.update(h=h, runner=r, m_cap=m)
                        reuse = False
                    r = state["runner"]
                    pred, err = r.predict(i, pcoords, reuse_factor=bool(reuse))
                    conn.send(("ok", (pred, err, dict(r.timings), dict(r.comm_info)) if rank == 0 else None))
                elif op == "predict_local":
                    _, i, pcoords, max_dist, cv = msg
                    if state.get("hl") is None:
                        state["hl"] = new_handle()
                    pred, err, info = DistributedLocal(state["hl"], rank, world, dist_module=dist, device=dev).predict(
                        i, pcoords, max_dist=max_dist, cv=cv, with_info=True)
                    conn.send(("ok", (pred, err, info) if rank == 0 else None))
                else:
                    conn.send(("err", "ValueError", f"unknown request {op!r}"))
            except Exception as e:   # noqa: BLE001 -- shipped to the caller
                conn.send(("err", type(e).__name__, f"{e}\n[rank {rank}]\n{traceback.format_exc()}"))
    except Exception as e:   # noqa: BLE001
        try:
            conn.send(("err", type(e).__name__, f"{e}\n[rank {rank} start-up]\n{traceback.format_exc()}"))
        except Exception:
            pass
    finally:
        if dist is not None and dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:
                pass


class RankError(RuntimeError):
    pass


class RankPool:
    """`len(devices)` worker processes, one rank per entry of `devices` (device ordinals; the same ordinal twice =
    two ranks sharing that GPU over gloo)."""

    def __init__(self, devices, backend: str = None, exchange: str = "broadcast", panel_group=1, start_timeout: float = 600.0):
        # exchange: "broadcast" until an 8-GPU run has recorded "sag" / "p2p" in comm_info ("auto" would send the
        # first real multi-GPU call of a notebook through an exchange that has only ever run on gloo); bench.py calibrates
        import torch.multiprocessing as mp
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("devices must name at least one GPU")
        self.world = len(self.devices)
        self.backend = backend or pick_backend(self.devices)
        self.timeout = float(os.environ.get("CK_RANK_TIMEOUT", "3600"))
        ctx = mp.get_context("spawn")
        port = _free_port()
        self._conns, self._procs = [], []
        opts = {"exchange": exchange, "panel_group": panel_group, "devices": list(self.devices)}
        for r, d in enumerate(self.devices):
            a, b = ctx.Pipe()
            p = ctx.Process(target=_rank_main, args=(r, self.world, d, port, self.backend, b, opts), daemon=True)
            p.start()
            b.close()
            self._conns.append(a)
            self._procs.append(p)
        self._collect(start_timeout)
        self.last_timings, self.last_comm = {}, {}

    # -- plumbing ------------------------------------------------------------------------------------
    def _collect(self, timeout=None, grace=None):
        """One reply per rank; raises if any rank failed or died.

        ONE deadline for the whole request (not one per rank), all pipes polled together.  With more than one rank an error
        on a single rank usually leaves the others inside a collective they will never leave, so the first
        error starts a short grace period (CK_RANK_GRACE, default 15 s: long enough for errors every rank raises together --
        a Sigma that is not positive definite -- to arrive from all of them); ranks still silent after it are taken to be
        stuck and the pool is killed.  A pool in which every rank answered, with errors or not, stays usable."""
        import time
        from multiprocessing.connection import wait as conn_wait
        timeout = self.timeout if timeout is None else timeout
        grace = float(os.environ.get("CK_RANK_GRACE", "15")) if grace is None else grace
        n = len(self._conns)
        replies, errors, lost = [None] * n, [], False
        pending = set(range(n))
        deadline = time.monotonic() + timeout
        err_deadline = None
        while pending:
            ready = conn_wait([self._conns[r] for r in pending], timeout=0.5)
            for r in sorted(pending):
                c, p = self._conns[r], self._procs[r]
                if c in ready:
                    pending.discard(r)
                    try:
                        msg = c.recv()
                    except (EOFError, OSError):
                        errors.append(("RankError", f"rank {r} closed its pipe (exit code {p.exitcode})"))
                        lost = True
                        continue
                    if msg[0] == "err":
                        errors.append((msg[1], msg[2]))
                    else:
                        replies[r] = msg[1]
                elif not p.is_alive():
                    pending.discard(r)
                    errors.append(("RankError", f"rank {r} died (exit code {p.exitcode})"))
                    lost = True
            now = time.monotonic()
            if errors and err_deadline is None:
                err_deadline = now + grace
            if pending and err_deadline is not None and now > err_deadline:
                errors.append(("RankError", f"rank(s) {sorted(pending)} did not answer within {grace:.0f} s of another rank's failure "
                                            "(stuck in a collective): pool terminated"))
                lost = True
                break
            if pending and now > deadline:
                errors.append(("RankError", f"rank(s) {sorted(pending)} did not answer within {timeout:.0f} s"))
                lost = True
                break
        if errors:
            if lost:
                self._kill()
            name, text = errors[0]
            if name == "LinAlgError":
                from numpy.linalg import LinAlgError
                raise LinAlgError(text.split("\n")[0])
            if name == "NativeError":
                from .native import NativeError
                raise NativeError(text)
            if name == "ValueError":
                raise ValueError(text)
            raise RankError(f"{name}: {text}")
        return replies

    def _request(self, *msg):
        if not self._procs:
            raise RankError("the rank pool is closed")
        for c in self._conns:
            try:
                c.send(msg)
            except (BrokenPipeError, OSError):
                pass            # a rank that is gone: _collect names it
        return self._collect()

    def _kill(self):
        for p in self._procs:
            if p.is_alive():
                p.terminate()
        for p in self._procs:
            p.join(timeout=10)
        self._procs, self._conns = [], []

    # -- requests -------------------------------------------------------------------------------------
    def load(self, model, metric, coords, values):
        """model = (n_procs, sigma, nu3, len3, nugget, rho12) (model.model_arrays); coords / values: per process."""
        payload = {"model": tuple(np.asarray(x, dtype=np.float64) if not np.isscalar(x) else x for x in model),
                   "metric": int(metric),
                   "coords": [np.ascontiguousarray(c, dtype=np.float64) for c in coords],
                   "values": [np.ascontiguousarray(v, dtype=np.float64) for v in values]}
        self._request("load", payload)

    def predict_joint(self, i, pcoords, reuse_factor=True):
        pc = np.ascontiguousarray(np.atleast_2d(np.asarray(pcoords, dtype=np.float64)))
        out = self._request("predict_joint", int(i), pc, bool(reuse_factor))[0]
        pred, err, self.last_timings, self.last_comm = out
        return pred, err

    def predict_local(self, i, pcoords, max_dist=1e3, cv=False):
        pc = np.ascontiguousarray(np.atleast_2d(np.asarray(pcoords, dtype=np.float64)))
        return self._request("predict_local", int(i), pc, float(max_dist), bool(cv))[0]

    def close(self):
        """Ask the ranks to leave; ranks that do not answer within a few seconds (stuck in a collective after another
        rank's failure) are terminated -- close() and garbage collection never wait for the request timeout."""
        if not self._procs:
            return
        try:
            for c in self._conns:
                try:
                    c.send(("close",))
                except (BrokenPipeError, OSError):
                    pass
            self._collect(timeout=float(os.environ.get("CK_RANK_CLOSE_TIMEOUT", "10")), grace=3.0)
        except Exception:
            pass
        for p in self._procs:
            p.join(timeout=5)
        self._kill()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
