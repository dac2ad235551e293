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


class DistributedJoint:
    """exchange = "broadcast": one collective broadcast per panel (RCCL picks ring / tree: every hop carries the whole
    panel, so a step costs panel_bytes / one link).  exchange = "p2p": the owner SCATTERS the panel -- piece r to rank r,
    seven pieces leaving on seven xGMI links at once -- and the ranks then exchange their pieces all-to-all by
    point-to-point sends (every rank sends its eighth to the six others over its own links): both phases move an
    eighth of the panel per link, 2 x panel_bytes / 8 per step instead of panel_bytes.  Same bytes land in the same
    receive buffer; chosen with CK_PANEL_EXCHANGE=p2p (bench.py) until it has been timed on an 8-GPU node."""

    def __init__(self, handle, rank: int, world: int, dist_module=None, device=None, group=None, lookahead=True,
                 exchange: str = "broadcast"):
        self.h, self.rank, self.world = handle, int(rank), int(world)
        self.dist, self.device, self.group = dist_module, device, group
        self.lookahead = bool(lookahead)
        if exchange not in ("broadcast", "p2p"):
            raise ValueError("exchange must be 'broadcast' or 'p2p'")
        self.exchange = exchange
        self.arena = None
        # this rank's breakdown of the last predict(): panel_ms (panel steps it owned, incl. the look-ahead column
        # update in front of them), update_ms (trailing + right-hand-side updates, collective enqueue), bcast_wait_ms
        # (its stream waiting for a panel after its own updates: exposed communication), assemble_ms, finish_ms
        self.timings = {}
        self._marks = _Marks(device)
        self._steps = []
        self._caller_order = False   # set once a not-positive-definite Sigma has been re-swept in the caller's order

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
        _, _, chunk = self.shard(m_total)
        nbytes = h.estimate_bytes(chunk)
        if hasattr(h, "make_arena"):           # CPU stand-in used by the gloo tests
            self.arena = h.make_arena(nbytes)
        else:
            self.arena = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            h.set_arena(self.arena.data_ptr(), nbytes, keepalive=self.arena)
            h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        return self

    def _panel_tensor(self, K):
        if hasattr(self.h, "panel_tensor"):
            return self.h.panel_tensor(K)
        import torch
        ptr, nbytes = self.h.panel_buffer(K)
        off = ptr - self.arena.data_ptr()
        assert 0 <= off and off + nbytes <= self.arena.numel(), "panel outside the arena"
        return self.arena[off:off + nbytes].view(torch.float64)

    def _pieces(self, n):
        """[lo, hi) of the world pieces of an n-element panel (multiples of 512 elements, the last takes the rest)."""
        step = -(-(-(-n // self.world)) // 512) * 512
        return [(min(r * step, n), min((r + 1) * step, n)) for r in range(self.world)]

    def _stream_ordered(self):
        try:
            return self.dist.get_backend(self.group) == "nccl"
        except Exception:
            return False

    def _exchange(self, K, src, async_op=False):
        """Panel K from its owner `src` into every rank's buffer for it."""
        t = self._panel_tensor(K)
        dist = self.dist
        if self.exchange == "broadcast" or self.world < 3 or not hasattr(dist, "P2POp"):
            return dist.broadcast(t, src=src, group=self.group, async_op=async_op)
        pc = self._pieces(t.numel())
        me = self.rank
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
        works = []
        if ops1:
            works += dist.batch_isend_irecv(ops1)
        if me != src and works and not self._stream_ordered():
            for w in works:                             # my piece must have arrived before I pass it on (RCCL: both
                w.wait()                                # batches run in order on the communicator's stream)
            works = []
        if ops2:
            works += dist.batch_isend_irecv(ops2)
        w = _Works(works)
        if async_op:
            return w
        w.wait()
        return None

    def _sweep_lookahead(self, nK):
        """Factor / broadcast / apply with the next panel's factorisation and transfer under the
        current panel's update.  The collective is enqueued BEFORE the big update kernels, so its
        kernel holds its few CUs before they fill the chip; it reads panel K + 1 and writes the
        receive buffer (K + 1) & 1, the update reads panel K (buffer K & 1) and writes columns
        beyond K + 1 -- disjoint."""
        h, dist, mk = self.h, self.dist, self._marks.mark
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

    # -- one pass of the hot path ---------------------------------------------------------------------
    def predict(self, i: int, pcoords):
        """assemble -> (factor + broadcast + apply) per panel -> reduce; returns the full-length
        (pred, pred_err) on every rank.  Raises numpy.linalg.LinAlgError like scipy's cho_factor
        when Sigma is not positive definite."""
        import torch
        h, dist = self.h, self.dist
        pc = np.ascontiguousarray(np.atleast_2d(np.asarray(pcoords, dtype=np.float64)))
        m = pc.shape[0]
        lo, hi, chunk = self.shard(m)
        mk = self._marks.mark
        self._steps = []
        ta = mk()
        h.assemble_joint()
        h.aux_begin(i, pc[lo:hi])
        tb = mk()
        nK, _, _ = h.num_panels()
        if self.world > 1 and self.lookahead:
            self._sweep_lookahead(nK)
        else:
            for K in range(nK):
                owner = K % self.world
                t0 = mk()
                if owner == self.rank:
                    h.panel_factor(K)
                t1 = mk()
                if self.world > 1:
                    self._exchange(K, owner)
                t2 = mk()
                h.panel_apply(K, native.APPLY_SIGMA | native.APPLY_AUX)
                t3 = mk()
                self._steps.append((t0, t1, t3, t3, t1, t2))   # plain schedule: the wait sits between t1 and t2
        tc = mk()
        pred_l, err_l = h.aux_finish()
        info = h.factor_info()
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
        if self.world == 1:
            pred, err = pred_l, err_l
        else:
            dev = self.device if self.device is not None else "cpu"
            buf = torch.zeros(2 * chunk + 1, dtype=torch.float64, device=dev)
            buf[:hi - lo] = torch.from_numpy(pred_l).to(dev)
            buf[chunk:chunk + hi - lo] = torch.from_numpy(err_l).to(dev)
            buf[2 * chunk] = float(info if info != 0 else 2 ** 62)
            allb = [torch.empty_like(buf) for _ in range(self.world)]
            dist.all_gather(allb, buf, group=self.group)
            allb = [b.cpu().numpy() for b in allb]
            pred = np.concatenate([b[:chunk] for b in allb])[:m]
            err = np.concatenate([b[chunk:2 * chunk] for b in allb])[:m]
            infos = [int(b[2 * chunk]) for b in allb]
            info = min(infos)
            info = 0 if info >= 2 ** 62 else info
        if info != 0:
            # every rank sees the same info.  scipy names the failing minor in the CALLER's site order: sweep once more
            # in that order (as ck_factor does for one process); the handle then stays in it
            if not self._caller_order and hasattr(h, "set_option"):
                self._caller_order = True
                h.set_option("site_order", 0)
                return self.predict(i, pcoords)
            from numpy.linalg import LinAlgError
            raise LinAlgError(f"{info}-th leading minor of the array is not positive definite")
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

    def predict(self, i: int, pcoords, max_dist: float = 1e3, cv: bool = False):
        import torch
        pc = np.ascontiguousarray(np.atleast_2d(np.asarray(pcoords, dtype=np.float64)))
        m = pc.shape[0]
        chunk = -(-m // self.world)
        lo = min(self.rank * chunk, m)
        hi = min(lo + chunk, m)
        pred_l, err_l, _ = self.h.predict_local(i, pc[lo:hi], max_dist, cv) if hi > lo else (np.empty(0), np.empty(0), None)
        if self.world == 1:
            return pred_l, err_l
        dev = self.device if self.device is not None else "cpu"
        buf = torch.full((2 * chunk,), float("nan"), dtype=torch.float64, device=dev)
        buf[:hi - lo] = torch.from_numpy(np.ascontiguousarray(pred_l)).to(dev)
        buf[chunk:chunk + hi - lo] = torch.from_numpy(np.ascontiguousarray(err_l)).to(dev)
        allb = [torch.empty_like(buf) for _ in range(self.world)]
        self.dist.all_gather(allb, buf, group=self.group)
        allb = [b.cpu().numpy() for b in allb]
        return (np.concatenate([b[:chunk] for b in allb])[:m], np.concatenate([b[chunk:] for b in allb])[:m])
