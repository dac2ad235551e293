"""TEST-ONLY CPU stand-in for native.Handle's step-wise interface (include/cokrige.h,
"step-wise form"), so that the multi-process orchestration in
sif-xco2-cokriging_amd/distributed.py -- panel ownership, broadcast order, sharding of the
prediction points, result gather -- can be exercised with world_size > 1 on the gloo backend
without a GPU.  It keeps the SAME data layout contract (packed block-column panels, remote
panels in a receive buffer) and does the panel algebra in numpy with covariances from the
oracle.  Never imported by the product."""
import numpy as np
import torch

from oracle import cokrige_oracle as orc


class FakePanelHandle:
    def __init__(self, params, coords, values, metric, NB=64):
        self.p = orc.Params.from_flat(params)
        self.coords, self.values, self.metric, self.NB = coords, values, metric, NB
        self.rank, self.world = 0, 1
        self.N = sum(len(v) for v in values)
        self.Npad = -(-self.N // NB) * NB
        self.nK = self.Npad // NB
        self.info = 0
        self.slots = 2

    # -- configuration ---------------------------------------------------------------------------
    def set_partition(self, rank, world):
        self.rank, self.world = rank, world

    def set_option(self, name, value):
        if name == "recv_slots":
            self.slots = int(value)
        # "site_order": the stand-in always works in the caller's order

    def estimate_bytes(self, m):
        return 8 * (self.Npad * self.NB * self.slots)

    SLACK = 64 * 512        # elements behind every panel buffer (include/cokrige.h: CK_PANEL_SLACK_BYTES)

    def make_arena(self, nbytes):
        self.recv = [torch.zeros(self.Npad * self.NB + self.SLACK, dtype=torch.float64) for _ in range(self.slots)]
        self.store, self.sig = {}, {}
        for K in range(self.rank, self.nK, self.world):   # owned panels: storage now (like the library's layout), content later
            n = (self.Npad - K * self.NB) * self.NB
            self.store[K] = torch.zeros(n + self.SLACK, dtype=torch.float64)
            self.sig[K] = self.store[K][:n]
        return torch.zeros(8, dtype=torch.uint8)

    def num_panels(self):
        return self.nK, self.NB, self.Npad

    # -- hot path ------------------------------------------------------------------------------------
    def assemble_joint(self):
        S = np.eye(self.Npad)
        S[:self.N, :self.N] = orc.joint_cov(self.p, self.coords, self.metric)
        NB = self.NB
        for K in range(self.rank, self.nK, self.world):
            self.sig[K].copy_(torch.from_numpy(np.ascontiguousarray(S[K * NB:, K * NB:(K + 1) * NB]).ravel()))
        self.info = 0

    def aux_begin(self, i, pcoords):
        self.i, self.m = i, len(pcoords)
        c0 = orc.pred_cross_cov(self.p, self.coords, pcoords, i, self.metric) if self.m else np.zeros((self.N, 0))
        X = np.zeros((self.m + 1, self.Npad))
        X[:self.m, :self.N] = c0.T
        X[self.m, :self.N] = np.hstack(self.values)
        self.X = X

    def panel_tensor(self, K, pad_to=0):
        rows = self.Npad - K * self.NB
        n = rows * self.NB
        if pad_to:
            assert 0 <= pad_to - n <= self.SLACK
            n = pad_to
        if K in self.sig:
            return self.store[K][:n]
        return self.recv[K % self.slots][:n]

    def panel_factor(self, K):
        NB = self.NB
        P = self.sig[K].numpy().reshape(-1, NB)
        D = np.tril(P[:NB]) + np.tril(P[:NB], -1).T
        try:
            L = np.linalg.cholesky(D)
        except np.linalg.LinAlgError:
            # first non-positive pivot, LAPACK style
            if self.info == 0:
                from scipy.linalg import lapack
                _, inf = lapack.dpotrf(D, lower=1)
                self.info = K * NB + int(inf)
            L = np.eye(NB)
        P[:NB] = L
        P[NB:] = np.linalg.solve(L, P[NB:].T).T

    def panel_apply_sigma(self, K, J_lo, J_hi):
        NB = self.NB
        P = self.panel_tensor(K).numpy().reshape(-1, NB)
        for J in range(max(J_lo, K + 1), min(J_hi, self.nK - 1) + 1):
            if J % self.world != self.rank:
                continue
            C = self.sig[J].numpy().reshape(-1, NB)
            A = P[(J - K) * NB:]
            C -= A @ A[:NB].T

    def panel_apply_group(self, K0, np_, what, J_lo, J_hi, phase=0, n_phase=1):
        """include/cokrige.h: ck_panel_apply_group -- same column / piece selection as the library"""
        NB = self.NB
        J_lo, J_hi = max(J_lo, K0 + np_), min(J_hi, self.nK - 1)
        if J_lo > J_hi:
            return
        Ps = [self.panel_tensor(K0 + p).numpy().reshape(-1, NB) for p in range(np_)]
        if what & 1:
            J0 = J_lo
            while J0 <= J_hi and J0 % self.world != self.rank:
                J0 += 1
            J0 += phase * self.world
            for J in range(J0, J_hi + 1, self.world * n_phase):
                C = self.sig[J].numpy().reshape(-1, NB)
                for p, P in enumerate(Ps):
                    A = P[(J - K0 - p) * NB:]
                    C -= A @ A[:NB].T
        if what & 2:
            tot = J_hi - J_lo + 1
            per = -(-tot // n_phase)
            a = J_lo + phase * per
            b = min(J_hi, a + per - 1)
            for J in range(a, b + 1):
                for p, P in enumerate(Ps):
                    K = K0 + p
                    self.X[:, J * NB:(J + 1) * NB] -= self.X[:, K * NB:(K + 1) * NB] @ P[(J - K) * NB:(J - K + 1) * NB].T

    def panel_aux_solve(self, K):
        NB = self.NB
        P = self.panel_tensor(K).numpy().reshape(-1, NB)
        L = np.tril(P[:NB])
        self.X[:, K * NB:(K + 1) * NB] = np.linalg.solve(L, self.X[:, K * NB:(K + 1) * NB].T).T

    def panel_apply(self, K, what):
        NB = self.NB
        P = self.panel_tensor(K).numpy().reshape(-1, NB)
        if what & 1:
            self.panel_apply_sigma(K, K + 1, self.nK - 1)
        if what & 2:
            L = np.tril(P[:NB])
            XK = np.linalg.solve(L, self.X[:, K * NB:(K + 1) * NB].T).T
            self.X[:, K * NB:(K + 1) * NB] = XK
            if K + 1 < self.nK:
                self.X[:, (K + 1) * NB:] -= XK @ P[NB:].T

    def aux_finish(self):
        y = self.X[self.m]
        V = self.X[:self.m]
        pred = V @ y
        c0 = self.p.sigma[self.i] ** 2 + self.p.nugget[self.i]
        with np.errstate(invalid="ignore"):
            err = np.nan_to_num(np.sqrt(c0 - np.einsum("ij,ij->i", V, V)))
        return pred, err

    def factor_info(self):
        return self.info


class FakeVarioHandle:
    """TEST-ONLY stand-in for the variogram part of native.Handle (ck_vario_*), sharded like the device
    kernels: with set_partition(rank, world) the two passes see only this rank's share of the pairs (here:
    the rows i = rank mod world; the device deals out 256 x 1024 pair tiles).  Distances and cloud values
    follow the oracle."""

    def __init__(self, metric):
        self.metric, self.rank, self.world = metric, 0, 1

    def set_partition(self, rank, world):
        self.rank, self.world = rank, world

    def vario_begin(self, coords_i, resid_i, coords_j=None, resid_j=None):
        self.same = coords_j is None
        ci, ri = np.asarray(coords_i, float), np.asarray(resid_i, float)
        cj, rj = (ci, ri) if self.same else (np.asarray(coords_j, float), np.asarray(resid_j, float))
        rows = np.arange(self.rank, len(ci), self.world)
        d = orc.distance_matrix(ci[rows], cj, self.metric)
        a, b = ri[rows], rj
        self._semi = 0.5 * np.subtract.outer(a, b) ** 2
        self._prod = np.multiply.outer(a, b)
        mask = np.ones_like(d, dtype=bool)
        if self.same:
            mask = np.arange(len(cj))[None, :] > rows[:, None]      # strict upper triangle
        self._d, self._mask = d, mask

    def vario_extent(self, max_dist):
        keep = self._mask & (self._d <= max_dist)
        d = self._d[keep]
        pos = d[d > 0]
        if pos.size == 0:
            return float("nan"), float("nan"), 0
        return float(pos.min()), float(d.max()), 1

    def vario_bin(self, max_dist, edges, covariogram=False):
        keep = self._mask & (self._d <= max_dist)
        d = self._d[keep]
        cloud = (self._prod if covariogram else self._semi)[keep]
        ids = np.searchsorted(edges, d, side="left")
        ids[d == edges[0]] = 1
        ok = (ids >= 1) & (ids <= len(edges) - 1)
        nb = len(edges) - 1
        counts = np.bincount(ids[ok] - 1, minlength=nb).astype(np.int64)
        sums = np.bincount(ids[ok] - 1, weights=cloud[ok], minlength=nb)
        return sums, counts

    def vario_end(self):
        self._d = self._mask = self._semi = self._prod = None
