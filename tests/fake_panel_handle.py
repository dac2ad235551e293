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

    # -- configuration ---------------------------------------------------------------------------
    def set_partition(self, rank, world):
        self.rank, self.world = rank, world

    def estimate_bytes(self, m):
        return 8 * (self.Npad * self.NB * 2)

    def make_arena(self, nbytes):
        self.recv = [torch.zeros(self.Npad * self.NB, dtype=torch.float64) for _ in range(2)]
        return torch.zeros(8, dtype=torch.uint8)

    def num_panels(self):
        return self.nK, self.NB, self.Npad

    # -- hot path ------------------------------------------------------------------------------------
    def assemble_joint(self):
        S = np.eye(self.Npad)
        S[:self.N, :self.N] = orc.joint_cov(self.p, self.coords, self.metric)
        NB = self.NB
        self.sig = {}
        for K in range(self.rank, self.nK, self.world):
            self.sig[K] = torch.from_numpy(np.ascontiguousarray(S[K * NB:, K * NB:(K + 1) * NB]).ravel().copy())
        self.info = 0

    def aux_begin(self, i, pcoords):
        self.i, self.m = i, len(pcoords)
        c0 = orc.pred_cross_cov(self.p, self.coords, pcoords, i, self.metric) if self.m else np.zeros((self.N, 0))
        X = np.zeros((self.m + 1, self.Npad))
        X[:self.m, :self.N] = c0.T
        X[self.m, :self.N] = np.hstack(self.values)
        self.X = X

    def panel_tensor(self, K):
        if K in self.sig:
            return self.sig[K]
        rows = self.Npad - K * self.NB
        return self.recv[K & 1][:rows * self.NB]

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
