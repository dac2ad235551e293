"""Multi-GPU joint cokriging: one process per GPU, ``torch.distributed`` for the single exchange
step the path has (the Cholesky panel broadcast; backend "nccl" = RCCL over xGMI).

Partition (include/cokrige.h, "step-wise form"):
  * Sigma: 1-D block-column-cyclic over ranks, panel width NB = 512 -- rank r assembles and owns
    the block columns K with K % world == r (no communication: every tile depends only on two
    coordinate slices, which are replicated);
  * prediction points: contiguous shards, one per rank -- each rank assembles its own rows of
    c0^T and forward-substitutes them with every panel it sees, so the solve rides on the
    panel broadcasts and needs no collective of its own;
  * per Cholesky step K: every rank holds panel K (the owner's storage or a receive buffer); the
    owner of K + 1 updates block column K + 1 first, factors it and the broadcast of panel K + 1
    ((Npad - (K + 1) NB) x NB doubles, contiguous) starts at once -- asynchronously, into the
    other of the two receive buffers -- while every rank updates the rest of its trailing block
    columns and its right-hand-side rows with panel K (look-ahead of depth 1: the panel step and
    the transfer run under the update instead of in front of it);
  * at the end: all-gather of the 2 m result values, MIN-reduce of the LAPACK-style info flag.

The torch tensors are plumbing: device memory (one uint8 arena the library carves its panels
from), the stream, and the collective.  All arithmetic is in the HIP library.

The two other paths of SURVEY.md section 8(e) shard without any exchange inside the computation:
  * DistributedVariogram -- the pair tiles of the lag-binning kernels are dealt out round-robin
    (tile t to rank t mod world); the ranks' extreme distances and per-bin sums / counts are
    combined by one MIN/MAX and one SUM all-reduce;
  * DistributedLocal -- the prediction points of the local-neighbourhood predictor are split into
    contiguous shards (the observations are replicated), results gathered.
"""
from __future__ import annotations

import numpy as np

from . import native

_BIG = np.iinfo(np.int64).max


class _Marks:
    """Time stamps on the rank's stream (HIP events through torch) or, for the CPU stand-in of the tests, on the
    host clock.  ms(a, b) is valid after the stream has been synchronised."""

    def __init__(self, device):
        self.cuda = device is not None and str(device).startswith("cuda")
        self.device = device

    def mark(self):
        if self.cuda:
            import torch
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.device))
            return e
        import time
        return time.perf_counter()

    def wait(self, e):
        if self.cuda:
            e.synchronize()

    def ms(self, a, b):
        return float(a.elapsed_time(b)) if self.cuda else (b - a) * 1e3


class _Works:
    def __init__(self, works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()
        return True


class _StreamWork:
    """An exchange that ran on a side stream: wait() makes the CURRENT stream wait for its end (no host block)."""

    def __init__(self, done_event, device):
        self.done, self.device = done_event, device

    def wait(self):
        import torch
        torch.cuda.current_stream(self.device).wait_event(self.done)
        return True


EXCHANGES = ("broadcast", "sag", "p2p")


class DistributedJoint:
    """One rank of the multi-GPU joint predictor.

    exchange -- how panel K travels from its owner to every rank:
      "broadcast"  one collective broadcast per panel (RCCL picks ring / tree);
      "sag"        scatter + in-place all-gather (two RCCL collectives): the owner hands piece r of the panel to rank r,
                   then all ranks gather the pieces -- every link carries panel_bytes / world per phase;
      "p2p"        the same data movement spelled as point-to-point sends (batch_isend_irecv): piece r to rank r over the
                   owner's seven xGMI links at once, then every rank forwards its piece to the others over its own links;
      "auto"       calibrate(): one mid-size panel through each of them at warm-up, all ranks take the one whose slowest
                   rank was fastest (timings and choice are kept in `comm_info`).
    panel_group -- 1: per-panel schedule with look-ahead of depth 1 (two receive slots); G > 1: the trailing updates
      are applied for G panels at once (K = 512 G, a G-th of the C traffic: _sweep_grouped, 2 G receive slots); "auto":
      storage for G = 3 and autotune() times one pass of either schedule and keeps the faster."""

    def __init__(self, handle, rank: int, world: int, dist_module=None, device=None, group=None, lookahead=True,
                 exchange: str = "broadcast", panel_group=1, rehearse_collectives: bool = False, chain_stream: bool = False):
        self.h, self.rank, self.world = handle, int(rank), int(world)
        self.dist, self.device, self.group = dist_module, device, group
        self.lookahead = bool(lookahead)
        # rehearse_collectives: issue every collective of the multi-rank schedules even at world == 1 (a one-rank
        # communicator: each call goes through the backend -- RCCL on a single GPU -- and moves nothing).  What a box with
        # one GPU can execute of the nccl path: tests/test_gpu_distributed.py::test_rccl_single_rank_*.
        self._comm = self.world > 1 or (bool(rehearse_collectives) and dist_module is not None)
        self._rehearse = bool(rehearse_collectives)
        if exchange not in EXCHANGES + ("auto",):
            raise ValueError("exchange must be one of " + ", ".join(EXCHANGES + ("auto",)))
        self.exchange = exchange
        if panel_group != "auto" and not (isinstance(panel_group, int) and 1 <= panel_group <= 8):
            raise ValueError("panel_group must be 1..8 or 'auto'")
        self.panel_group = panel_group
        self.G = 1 if panel_group == "auto" else int(panel_group)   # the schedule in force
        self.slots = 2 if panel_group == 1 else 2 * (3 if panel_group == "auto" else int(panel_group))
        self.arena = None
        # this rank's breakdown of the last predict(): panel_ms (panel steps it owned, incl. the look-ahead column
        # update in front of them), update_ms (trailing + right-hand-side updates, collective enqueue), bcast_wait_ms
        # (its stream waiting for a panel after its own updates: exposed communication), assemble_ms, finish_ms
        self.timings = {}
        self.comm_info = {"exchange": None if exchange == "auto" else exchange, "calibration_ms": None}
        self.tune_info = {"panel_group": self.G, "pass_ms": None}
        self._marks = _Marks(device)
        self._steps = []
        self._caller_order = False   # set once a not-positive-definite Sigma has been re-swept in the caller's order
        self._resident = False       # the factor of the last predict() is still in the panels
        self._xstream = None
        # chain_stream (per-panel schedule, GPU only; off by default until a multi-GPU box has run it): the next panel's column
        # update, panel step and exchange go to a second, high-priority stream and run UNDER the current panel's bulk update
        # instead of in front of it (_sweep_chain_stream) -- the look-ahead of the single-process tall sweep in the step-wise form
        self.chain_stream = bool(chain_stream)
        self._cstream = None

    # -- setup ------------------------------------------------------------------------------------
    def shard(self, m_total: int):
        """[lo, hi) of this rank's prediction points and the common (padded) shard length."""
        chunk = -(-m_total // self.world)
        lo = min(self.rank * chunk, m_total)
        return lo, min(lo + chunk, m_total), chunk

    def prepare(self, m_total: int):
        """Partition the handle and give it one torch-owned arena sized for `m_total` prediction
        points (call once, after set_model / set_data)."""
        import torch
        h = self.h
        h.set_partition(self.rank, self.world)
        if self.world > 1 and self.slots != 2:
            h.set_option("recv_slots", self.slots)
        _, _, chunk = self.shard(m_total)
        nbytes = h.estimate_bytes(chunk)
        if hasattr(h, "make_arena"):           # CPU stand-in used by the gloo tests
            self.arena = h.make_arena(nbytes)
        else:
            self.arena = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            h.set_arena(self.arena.data_ptr(), nbytes, keepalive=self.arena)
            h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        return self

    def _panel_tensor(self, K, pad_to=0):
        """Panel K as a float64 tensor; pad_to > 0: lengthened to that many elements into the slack behind every panel
        buffer (include/cokrige.h: CK_PANEL_SLACK_BYTES)."""
        if hasattr(self.h, "panel_tensor"):
            return self.h.panel_tensor(K, pad_to) if pad_to else self.h.panel_tensor(K)
        import torch
        ptr, nbytes = self.h.panel_buffer(K)
        if pad_to:
            assert 0 <= 8 * pad_to - nbytes <= native.PANEL_SLACK_BYTES, "padding beyond the panel's slack"
            nbytes = 8 * pad_to
        off = ptr - self.arena.data_ptr()
        assert 0 <= off and off + nbytes <= self.arena.numel(), "panel outside the arena"
        return self.arena[off:off + nbytes].view(torch.float64)

    def _piece(self, n):
        """Length of one of the world equal pieces of an n-element panel (a multiple of 512 elements = 4 KB)."""
        return -(-(-(-n // self.world)) // 512) * 512

    def _pieces(self, n):
        """[lo, hi) of the world pieces of an n-element panel (multiples of 512 elements, the last takes the rest)."""
        step = self._piece(n)
        return [(min(r * step, n), min((r + 1) * step, n)) for r in range(self.world)]

    def _backend(self):
        try:
            return self.dist.get_backend(self.group)
        except Exception:
            return None

    def _on_side_stream(self, fn):
        """Run fn() -- which enqueues communication and may wait on it -- on a side stream that first waits for everything
        enqueued on the current stream so far; returns a work whose wait() makes the current stream wait for fn's end.
        nccl only (stream-ordered works); elsewhere fn() simply runs."""
        if self._backend() != "nccl" or self.device is None:
            return _Works(fn() or [])
        import torch
        if self._xstream is None:
            self._xstream = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        done = torch.cuda.Event()
        with torch.cuda.stream(self._xstream):
            self._xstream.wait_event(ready)
            for w in (fn() or []):
                w.wait()            # stream-ordered: blocks the side stream only
            done.record(self._xstream)
        return _StreamWork(done, self.device)

    def _exchange(self, K, src, async_op=False, how=None):
        """Panel K from its owner `src` into every rank's buffer for it."""
        dist = self.dist
        how = how or self.exchange
        if how == "auto":
            how = "broadcast"       # not calibrated (yet)
        if how == "p2p" and ((self.world < 3 and not self._rehearse) or not hasattr(dist, "P2POp")):
            how = "broadcast"
        if how == "sag" and ((self.world < 2 and not self._rehearse) or not hasattr(dist, "all_gather_into_tensor")):
            how = "broadcast"
        if how == "broadcast":
            t = self._panel_tensor(K)
            return dist.broadcast(t, src=src, group=self.group, async_op=async_op)
        me = self.rank
        if how == "sag":
            n = self._panel_tensor(K).numel()
            piece = self._piece(n)
            tp = self._panel_tensor(K, pad_to=piece * self.world)
            mine = tp[me * piece:(me + 1) * piece]
            ordered = self._backend() == "nccl"

            def run():
                lst = [tp[r * piece:(r + 1) * piece] for r in range(self.world)] if me == src else None
                w1 = dist.scatter(mine, lst, src=src, group=self.group, async_op=True)
                if not ordered:
                    w1.wait()       # gloo: the piece must have arrived before it is gathered
                    w2 = dist.all_gather_into_tensor(tp, mine.clone(), group=self.group, async_op=True)
                    return [w2]
                # RCCL: both collectives run in order on the communicator's stream; in-place gather (input = own slice)
                w2 = dist.all_gather_into_tensor(tp, mine, group=self.group, async_op=True)
                return [w1, w2]
            w = self._on_side_stream(run)
            if async_op:
                return w
            w.wait()
            return None
        # "p2p"
        t = self._panel_tensor(K)
        pc = self._pieces(t.numel())
        ops1, ops2 = [], []
        if me == src:                                   # phase 1: scatter, piece r -> rank r
            for r in range(self.world):
                if r != src and pc[r][1] > pc[r][0]:
                    ops1.append(dist.P2POp(dist.isend, t[pc[r][0]:pc[r][1]], r, self.group))
            lo, hi = pc[src]                            # the owner's own piece goes to everybody in phase 2
            for r in range(self.world):
                if r != src and hi > lo:
                    ops2.append(dist.P2POp(dist.isend, t[lo:hi], r, self.group))
        else:
            lo, hi = pc[me]
            if hi > lo:
                ops1.append(dist.P2POp(dist.irecv, t[lo:hi], src, self.group))
            for r in range(self.world):                 # phase 2: all-gather of the pieces by point-to-point sends
                if r == me:
                    continue
                if r != src and hi > lo:
                    ops2.append(dist.P2POp(dist.isend, t[lo:hi], r, self.group))
                if pc[r][1] > pc[r][0]:
                    ops2.append(dist.P2POp(dist.irecv, t[pc[r][0]:pc[r][1]], r, self.group))

        def run():
            works = dist.batch_isend_irecv(ops1) if ops1 else []
            for w in works:          # my piece must have arrived before I pass it on: explicit, whatever the backend
                w.wait()             # (on the side stream this blocks neither the host nor the update kernels)
            return dist.batch_isend_irecv(ops2) if ops2 else []
        w = self._on_side_stream(run)
        if async_op:
            return w
        w.wait()
        return None

    # -- calibration ------------------------------------------------------------------------------
    def _reduce_max(self, x):
        import torch
        dev = self.device if (self.device is not None and self._backend() == "nccl") else "cpu"
        t = torch.tensor([float(v) for v in x], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return [float(v) for v in t.cpu().tolist()]

    def _sync(self):
        if self._marks.cuda:
            import torch
            torch.cuda.synchronize(self.device)

    def calibrate(self, reps: int = 2, candidates=None):
        """exchange = "auto": send one mid-size panel through every candidate exchange (one warm-up, `reps`
        timed repetitions, host clock around device synchronisation), MAX over the ranks of each, and keep the
        fastest -- every rank computes the same choice from the same reduced numbers.  The panel's content is
        whatever the buffers hold (call before or between passes, not inside one).
        candidates: default ("broadcast", "sag") -- two forms built from standard collectives only; "p2p" (hand-rolled
        batch_isend_irecv phases) competes when asked for (CK_PANEL_EXCHANGE_CANDIDATES=broadcast,sag,p2p for bench.py):
        it has run on gloo and through a host-staged facade, never on RCCL, and the first multi-GPU run should not
        depend on it."""
        import time
        if not self._comm or self.exchange != "auto":
            return self.comm_info
        nK = self.h.num_panels()[0]
        K = nK // 2
        src = K % self.world
        want = tuple(candidates) if candidates else ("broadcast", "sag")
        cands = ["broadcast"]
        if "sag" in want and hasattr(self.dist, "all_gather_into_tensor"):
            cands.append("sag")
        if "p2p" in want and (self.world >= 3 or self._rehearse) and hasattr(self.dist, "P2POp"):
            cands.append("p2p")
        ms = []
        for how in cands:
            # Every rank issues the SAME sequence of collectives whatever happens inside its own try block (a
            # rank that swallowed an exception used to skip the barrier and the timed exchanges the others still issued).
            # A failure is agreed on collectively -- one MAX all-reduce of a flag is the only collective that follows a try
            # block -- and a candidate that failed anywhere is dropped everywhere; a failure inside the timed part, after
            # everybody agreed the warm-up worked, propagates: the communicator is not to be trusted any more.
            bad = 0.0
            try:
                self._exchange(K, src, how=how)
                self._sync()
            except Exception:      # an exchange this backend cannot run is simply not a candidate
                bad = 1.0
            if self._reduce_max([bad])[0] > 0.0:
                ms.append(float("inf"))
                continue
            self.dist.barrier(group=self.group)
            t0 = time.perf_counter()
            for _ in range(reps):
                self._exchange(K, src, how=how)
            self._sync()
            ms.append((time.perf_counter() - t0) / reps * 1e3)
        ms = self._reduce_max(ms)
        best = min(range(len(cands)), key=lambda k: ms[k])
        self.exchange = cands[best]
        n = self._panel_tensor(K).numel()
        self.comm_info = {"exchange": self.exchange, "calibration_panel": K, "calibration_bytes": 8 * n,
                          "calibration_ms": {c: (None if v == float("inf") else round(v, 3)) for c, v in zip(cands, ms)}}
        return self.comm_info

    def autotune(self, i: int, pcoords):
        """panel_group = "auto": one full pass with the per-panel schedule, one with the grouped one (G = 3), host clock,
        MAX over the ranks; keeps the faster.  Returns the result of the last pass."""
        import time
        if self.panel_group != "auto":
            return self.predict(i, pcoords)
        res, ms = None, []
        for G in (1, 3):
            self.G = G
            self.predict(i, pcoords)        # untimed: first use of a schedule (allocations, clocks)
            self._sync()
            if self._comm:
                self.dist.barrier(group=self.group)
            t0 = time.perf_counter()
            res = self.predict(i, pcoords)
            self._sync()
            ms.append((time.perf_counter() - t0) * 1e3)
        if self._comm:
            ms = self._reduce_max(ms)
        self.G = 1 if ms[0] <= ms[1] else 3
        self.tune_info = {"panel_group": self.G, "pass_ms": {"G=1": round(ms[0], 2), "G=3": round(ms[1], 2)}}
        return res

    # -- schedules ----------------------------------------------------------------------------------
    def _sweep_lookahead(self, nK):
        """Factor / broadcast / apply with the next panel's factorisation and transfer under the
        current panel's update.  The collective is enqueued BEFORE the big update kernels, so its
        kernel holds its few CUs before they fill the chip; it reads panel K + 1 and writes the
        receive buffer (K + 1) % slots, the update reads panel K (buffer K % slots) and writes columns
        beyond K + 1 -- disjoint."""
        h, mk = self.h, self._marks.mark
        t0 = mk()
        if self.rank == 0:
            h.panel_factor(0)
        t1 = mk()
        self._exchange(0, 0)
        self._steps.append((t0, t1, t1, mk()))
        for K in range(nK):
            nxt, work = K + 1, None
            t0 = mk()
            if nxt < nK:
                if nxt % self.world == self.rank:
                    h.panel_apply_sigma(K, nxt, nxt)
                    h.panel_factor(nxt)
                t1 = mk()
                work = self._exchange(nxt, nxt % self.world, async_op=True)
            else:
                t1 = t0
            h.panel_apply_sigma(K, nxt + 1, nK - 1)
            h.panel_apply(K, native.APPLY_AUX)
            t2 = mk()
            if work is not None:
                work.wait()
            self._steps.append((t0, t1, t2, mk()))

    def _sweep_chain_stream(self, nK):
        """_sweep_lookahead with the chain on its own stream.  Per panel K, stream M (the handle's): the bulk update by panel K
        (block columns beyond K + 1, right-hand sides); stream C (high priority): block column K + 1 by panel K, its panel step,
        its exchange.  C starts behind an event on M that stands behind M's wait for panel K AND behind the bulk update of panel
        K - 1 -- which wrote block column K + 1 and was the last reader of the receive slot the exchange of panel K + 1 writes --;
        M waits for C's end (and the exchange) before it touches panel K + 1.  Same kernels, same order per block column: same
        bits as the other schedules."""
        import torch
        h, mk, world, me = self.h, self._marks.mark, self.world, self.rank
        M = torch.cuda.current_stream(self.device)
        if self._cstream is None:
            self._cstream = torch.cuda.Stream(device=self.device, priority=-1)
        C = self._cstream
        t0 = mk()
        if me == 0:
            h.panel_factor(0)
        t1 = mk()
        if self._comm:
            self._exchange(0, 0)
        self._steps.append((t0, t1, t1, mk()))
        for K in range(nK):
            nxt, work, done, solved = K + 1, None, None, None
            t0 = mk()
            if nxt < nK:
                ready = torch.cuda.Event()
                ready.record(M)
                C.wait_event(ready)
                with torch.cuda.stream(C):
                    h.set_stream(C.cuda_stream)
                    try:
                        # the right-hand-side rows' walk through panel K (140 small workgroups) rides on C as well: it only
                        # needs panel K and what the bulk update of panel K - 1 left in block column K (behind `ready`)
                        h.panel_aux_solve(K)
                        solved = torch.cuda.Event()
                        solved.record(C)
                        if nxt % world == me:
                            h.panel_apply_sigma(K, nxt, nxt)
                            h.panel_factor(nxt)
                        if self._comm:
                            work = self._exchange(nxt, nxt % world, async_op=True)
                        done = torch.cuda.Event()
                        done.record(C)
                    finally:
                        h.set_stream(M.cuda_stream)
            t1 = mk()
            h.panel_apply_sigma(K, nxt + 1, nK - 1)
            if solved is not None:
                M.wait_event(solved)
                h.panel_apply_group(K, 1, native.APPLY_AUX, nxt, nK - 1)
            else:
                h.panel_apply(K, native.APPLY_AUX)
            t2 = mk()
            if done is not None:
                M.wait_event(done)
            if work is not None:
                work.wait()
            self._steps.append((t0, t1, t2, mk()))

    def _sweep_grouped(self, nK, G):
        """Trailing updates for G panels at once (contraction dimension 512 G: a G-th of the read-modify-write traffic
        of the per-panel schedule), with the look-ahead kept: while the group K0 .. K0 + G - 1 is resident on every
        rank, the NEXT group's panels are factored one after the other -- their owners update their block column first
        (by the resident group, then by the panels of the next group that have already arrived), factor it and start
        its exchange -- and between those steps every rank applies one of G pieces of the resident group's update to
        the rest of its block columns and right-hand-side rows.  2 G receive slots: G panels in use, G arriving."""
        h, mk, world, me = self.h, self._marks.mark, self.world, self.rank
        SIG, AUX = native.APPLY_SIGMA, native.APPLY_AUX
        works = {}

        def need(K):            # panel K must have arrived before a kernel that reads it is enqueued
            w = works.pop(K, None)
            if w is not None:
                w.wait()

        def chain(N0, Gn, K0, Gc):
            """panels N0 .. N0 + Gn - 1; between them the pieces of the update by the resident group K0 .. K0 + Gc - 1"""
            for g in range(Gn):
                J = N0 + g
                t0 = mk()
                if J % world == me:
                    if g > 0:
                        for q in range(N0, J):
                            need(q)
                        h.panel_apply_group(N0, g, SIG, J, J)
                    h.panel_factor(J)
                t1 = mk()
                if self._comm:
                    works[J] = self._exchange(J, J % world, async_op=True)
                if Gc > 0:
                    h.panel_apply_group(K0, Gc, SIG, N0 + Gn, nK - 1, g, Gn)
                    h.panel_apply_group(K0, Gc, AUX, N0, nK - 1, g, Gn)
                t2 = mk()
                self._steps.append((t0, t1, t2, t2))

        chain(0, min(G, nK), 0, 0)          # the first group has nothing to hide under
        K0 = 0
        while K0 < nK:
            Gc = min(G, nK - K0)
            N0 = K0 + Gc
            Gn = min(G, nK - N0)
            t0 = mk()
            for q in range(K0, N0):
                need(q)
            t1 = mk()
            self._steps.append((t0, t0, t0, t1))           # exposed communication: the stream waited here
            t0 = mk()
            for g in range(Gc):             # right-hand sides through the resident group (local work only)
                if g > 0:
                    h.panel_apply_group(K0, g, AUX, K0 + g, K0 + g)
                h.panel_aux_solve(K0 + g)
            if Gn > 0:                      # this rank's block columns of the next group come first
                h.panel_apply_group(K0, Gc, SIG, N0, N0 + Gn - 1)
            t1 = mk()
            self._steps.append((t0, t0, t1, t1))
            if Gn > 0:
                chain(N0, Gn, K0, Gc)
            K0 = N0
        for q in list(works):
            need(q)

    def _sweep_solve_only(self, nK):
        """The factor is resident (panel K in its owner's storage): every rank needs each panel once more for its own
        right-hand-side rows -- exchange of panel K + 1 under the substitution with panel K."""
        h, mk = self.h, self._marks.mark
        work = self._exchange(0, 0, async_op=True) if self._comm else None
        for K in range(nK):
            t0 = mk()
            if work is not None:
                work.wait()
            t1 = mk()
            nxt = K + 1
            work = self._exchange(nxt, nxt % self.world, async_op=True) if (self._comm and nxt < nK) else None
            h.panel_apply(K, native.APPLY_AUX)
            t2 = mk()
            self._steps.append((t0, t0, t2, t2, t0, t1))

    # -- one pass of the hot path ---------------------------------------------------------------------
    def predict(self, i: int, pcoords, reuse_factor: bool = False):
        """assemble -> (factor + exchange + apply) per panel -> reduce; returns the full-length
        (pred, pred_err) on every rank.  Raises numpy.linalg.LinAlgError like scipy's cho_factor
        when Sigma is not positive definite.  reuse_factor: the factor of the previous predict() is still resident
        (same model, same data) -- only the right-hand sides are assembled and swept."""
        import torch
        h, dist = self.h, self.dist
        pc = np.ascontiguousarray(np.atleast_2d(np.asarray(pcoords, dtype=np.float64)))
        m = pc.shape[0]
        lo, hi, chunk = self.shard(m)
        mk = self._marks.mark
        self._steps = []
        solve_only = bool(reuse_factor and self._resident)
        self._resident = False
        ta = mk()
        if not solve_only:
            h.assemble_joint()
        h.aux_begin(i, pc[lo:hi])
        tb = mk()
        nK, _, _ = h.num_panels()
        if solve_only:
            self._sweep_solve_only(nK)
        elif self.G > 1:
            self._sweep_grouped(nK, self.G)
        elif self.chain_stream and self.lookahead and self._marks.cuda:
            self._sweep_chain_stream(nK)
        elif self._comm and self.lookahead:
            self._sweep_lookahead(nK)
        else:
            for K in range(nK):
                owner = K % self.world
                t0 = mk()
                if owner == self.rank:
                    h.panel_factor(K)
                t1 = mk()
                if self._comm:
                    self._exchange(K, owner)
                t2 = mk()
                h.panel_apply(K, native.APPLY_SIGMA | native.APPLY_AUX)
                t3 = mk()
                self._steps.append((t0, t1, t3, t3, t1, t2))   # plain schedule: the wait sits between t1 and t2
        tc = mk()
        # The outcome is COLLECTIVE: an error that only this rank sees -- a timed-out cooperative panel step
        # (ck_factor_info reports it and switches the step off for this handle), an allocation failure, ... -- must not
        # keep this rank out of the gather below while the others block in it.  It travels in the gathered buffer:
        # status 1 = "my panel step timed out: sweep again" (every rank then repeats the pass together, as ck_factor
        # does for one process), status 2 = any other failure (every rank raises).
        status, local_err = 0, None
        try:
            pred_l, err_l = h.aux_finish()
            info = h.factor_info()
        except native.NativeError as e:
            local_err = e
            status = 1 if "cooperative panel step timed out" in str(e) else 2
            pred_l, err_l, info = np.zeros(hi - lo), np.zeros(hi - lo), 0
        td = mk()
        self._marks.wait(td)
        ms = self._marks.ms
        tm = {"assemble_ms": ms(ta, tb), "finish_ms": ms(tc, td), "panel_ms": 0.0, "update_ms": 0.0, "bcast_wait_ms": 0.0}
        for st in self._steps:
            if len(st) == 4:
                tm["panel_ms"] += ms(st[0], st[1])
                tm["update_ms"] += ms(st[1], st[2])
                tm["bcast_wait_ms"] += ms(st[2], st[3])
            else:
                tm["panel_ms"] += ms(st[0], st[1])
                tm["bcast_wait_ms"] += ms(st[4], st[5])
                tm["update_ms"] += ms(st[5], st[2])
        self.timings = tm
        statuses = [status]
        if not self._comm:
            pred, err = pred_l, err_l
        else:
            dev = self.device if self.device is not None else "cpu"
            buf = torch.zeros(2 * chunk + 2, dtype=torch.float64, device=dev)
            buf[:hi - lo] = torch.from_numpy(pred_l).to(dev)
            buf[chunk:chunk + hi - lo] = torch.from_numpy(err_l).to(dev)
            buf[2 * chunk] = float(info if info != 0 else 2 ** 62)
            buf[2 * chunk + 1] = float(status)
            allb = [torch.empty_like(buf) for _ in range(self.world)]
            dist.all_gather(allb, buf, group=self.group)
            allb = [b.cpu().numpy() for b in allb]
            pred = np.concatenate([b[:chunk] for b in allb])[:m]
            err = np.concatenate([b[chunk:2 * chunk] for b in allb])[:m]
            infos = [int(b[2 * chunk]) for b in allb]
            info = min(infos)
            info = 0 if info >= 2 ** 62 else info
            statuses = [int(b[2 * chunk + 1]) for b in allb]
        if max(statuses) >= 2:
            if local_err is not None and status >= 2:
                raise local_err
            bad = [r for r, st in enumerate(statuses) if st >= 2]
            raise native.NativeError(f"rank(s) {bad} failed in this pass (their own error is raised in their process)")
        if max(statuses) == 1:
            # some rank's cooperative panel step timed out; its handle now runs the panel step launch by launch.  Everybody
            # sweeps again (bounded: a handle reports this at most once -- the step is off afterwards)
            self._coop_resweeps = getattr(self, "_coop_resweeps", 0) + 1
            if self._coop_resweeps > self.world + 1:
                raise native.NativeError("cooperative panel step keeps timing out")
            return self.predict(i, pcoords)
        if info != 0:
            # every rank sees the same info.  scipy names the failing minor in the CALLER's site order: sweep once more
            # in that order (as ck_factor does for one process); the handle then stays in it
            if not self._caller_order and hasattr(h, "set_option"):
                self._caller_order = True
                h.set_option("site_order", 0)
                return self.predict(i, pcoords)
            from numpy.linalg import LinAlgError
            raise LinAlgError(f"{info}-th leading minor of the array is not positive definite")
        self._resident = True
        return pred, err


def _to_dev(x, device):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64).to(device if device is not None else "cpu")


class DistributedVariogram:
    """Empirical (cross-)variogram with the pair tiles sharded over ranks (src/fields.py:192-232).
    Every rank passes the SAME coordinates / values and receives the same result."""

    def __init__(self, handle, rank: int, world: int, dist_module=None, device=None, group=None):
        self.h, self.rank, self.world = handle, int(rank), int(world)
        self.dist, self.device, self.group = dist_module, device, group
        handle.set_partition(self.rank, self.world)

    def variogram_arrays(self, coords_i, values_i, coords_j, values_j, same, max_dist, n_bins, covariogram=False):
        """(centers, edges, means, counts) -- as variogram.variogram_arrays on one device."""
        from .variogram import construct_bins
        h, dist = self.h, self.dist
        vi = np.asarray(values_i, dtype=np.float64)
        ri = vi - vi.mean()                       # src/fields.py:380
        if same:
            h.vario_begin(coords_i, ri)
        else:
            vj = np.asarray(values_j, dtype=np.float64)
            h.vario_begin(coords_i, ri, coords_j, vj - vj.mean())
        try:
            lo, hi, npos = h.vario_extent(max_dist)
            ext = np.array([lo if npos else np.inf, -(hi if npos else -np.inf), -float(bool(npos))])
            if self.world > 1:                    # one MIN all-reduce: min lo, max hi, "anybody found a pair"
                t = _to_dev(ext, self.device)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
                ext = t.cpu().numpy()
            if ext[2] == 0.0:
                raise ValueError("no pair of distinct sites within max_dist")
            lo, hi = float(ext[0]), float(-ext[1])
            centers, edges = construct_bins(lo, hi, n_bins)
            if len(edges) != n_bins + 1:
                raise ValueError("Bin labels must be one fewer than the number of bin edges")
            sums, counts = h.vario_bin(max_dist, edges, covariogram)
            if self.world > 1:
                import torch
                ts = _to_dev(sums, self.device)
                tc = torch.as_tensor(np.ascontiguousarray(counts), dtype=torch.int64).to(self.device if self.device is not None else "cpu")
                dist.all_reduce(ts, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(tc, op=dist.ReduceOp.SUM, group=self.group)
                sums, counts = ts.cpu().numpy(), tc.cpu().numpy()
        finally:
            h.vario_end()
        with np.errstate(invalid="ignore", divide="ignore"):
            means = sums / counts
        return centers, edges, means, counts


class DistributedLocal:
    """Local-neighbourhood predictor with the prediction points sharded over ranks
    (src/point_prediction.py:45-96 uses Pool.starmap over partitions of the points)."""

    def __init__(self, handle, rank: int, world: int, dist_module=None, device=None, group=None):
        self.h, self.rank, self.world = handle, int(rank), int(world)
        self.dist, self.device, self.group = dist_module, device, group

    def predict(self, i: int, pcoords, max_dist: float = 1e3, cv: bool = False, with_info: bool = False):
        """(pred, pred_err) of all points on every rank; with_info: also the counters of ck_predict_local summed
        (n_empty, n_not_pd) / maximised (k_max) over the shards -- what the caller's warnings are made from."""
        import torch
        pc = np.ascontiguousarray(np.atleast_2d(np.asarray(pcoords, dtype=np.float64)))
        m = pc.shape[0]
        chunk = -(-m // self.world)
        lo = min(self.rank * chunk, m)
        hi = min(lo + chunk, m)
        none = dict(n_empty=0, n_not_pd=0, k_max=0)
        pred_l, err_l, info = self.h.predict_local(i, pc[lo:hi], max_dist, cv) if hi > lo else (np.empty(0), np.empty(0), none)
        info = dict(none) if info is None else info
        if self.world == 1:
            return (pred_l, err_l, info) if with_info else (pred_l, err_l)
        dev = self.device if self.device is not None else "cpu"
        buf = torch.full((2 * chunk + 3,), float("nan"), dtype=torch.float64, device=dev)
        buf[:hi - lo] = torch.from_numpy(np.ascontiguousarray(pred_l)).to(dev)
        buf[chunk:chunk + hi - lo] = torch.from_numpy(np.ascontiguousarray(err_l)).to(dev)
        buf[2 * chunk:] = torch.tensor([float(info["n_empty"]), float(info["n_not_pd"]), float(info["k_max"])], dtype=torch.float64).to(dev)
        allb = [torch.empty_like(buf) for _ in range(self.world)]
        self.dist.all_gather(allb, buf, group=self.group)
        allb = [b.cpu().numpy() for b in allb]
        pred = np.concatenate([b[:chunk] for b in allb])[:m]
        err = np.concatenate([b[chunk:2 * chunk] for b in allb])[:m]
        if not with_info:
            return pred, err
        info = dict(n_empty=int(sum(b[2 * chunk] for b in allb)), n_not_pd=int(sum(b[2 * chunk + 1] for b in allb)),
                    k_max=int(max(b[2 * chunk + 2] for b in allb)))
        return pred, err, info
