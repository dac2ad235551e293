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
"""
from __future__ import annotations

import numpy as np

from . import native

_BIG = np.iinfo(np.int64).max


class DistributedJoint:
    def __init__(self, handle, rank: int, world: int, dist_module=None, device=None, group=None, lookahead=True):
        self.h, self.rank, self.world = handle, int(rank), int(world)
        self.dist, self.device, self.group = dist_module, device, group
        self.lookahead = bool(lookahead)
        self.arena = None
        self.timings = {}

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

    def _sweep_lookahead(self, nK):
        """Factor / broadcast / apply with the next panel's factorisation and transfer under the
        current panel's update.  The collective is enqueued BEFORE the big update kernels, so its
        kernel holds its few CUs before they fill the chip; it reads panel K + 1 and writes the
        receive buffer (K + 1) & 1, the update reads panel K (buffer K & 1) and writes columns
        beyond K + 1 -- disjoint."""
        h, dist = self.h, self.dist
        if self.rank == 0:
            h.panel_factor(0)
        dist.broadcast(self._panel_tensor(0), src=0, group=self.group)
        for K in range(nK):
            nxt, work = K + 1, None
            if nxt < nK:
                if nxt % self.world == self.rank:
                    h.panel_apply_sigma(K, nxt, nxt)
                    h.panel_factor(nxt)
                work = dist.broadcast(self._panel_tensor(nxt), src=nxt % self.world, group=self.group, async_op=True)
            h.panel_apply_sigma(K, nxt + 1, nK - 1)
            h.panel_apply(K, native.APPLY_AUX)
            if work is not None:
                work.wait()

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
        h.assemble_joint()
        h.aux_begin(i, pc[lo:hi])
        nK, _, _ = h.num_panels()
        if self.world > 1 and self.lookahead:
            self._sweep_lookahead(nK)
        else:
            for K in range(nK):
                owner = K % self.world
                if owner == self.rank:
                    h.panel_factor(K)
                if self.world > 1:
                    dist.broadcast(self._panel_tensor(K), src=owner, group=self.group)
                h.panel_apply(K, native.APPLY_SIGMA | native.APPLY_AUX)
        pred_l, err_l = h.aux_finish()
        info = h.factor_info()
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
            from numpy.linalg import LinAlgError
            raise LinAlgError(f"{info}-th leading minor of the array is not positive definite")
        return pred, err
