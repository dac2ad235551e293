"""Parity at the HEADLINE sizes against something that is neither the oracle's dense path (which stops at a few
thousand sites) nor the library itself: the reference's lines src/joint_prediction.py:67-78 replayed with

  * Sigma and c0 entry by entry from the library's EXACT per-entry evaluator (ck_cov_dense -- the device K_nu, held to
    the reference's covariance blocks by tests/test_gpu_joint.py::test_cov_dense_blocks and to scipy's kv by the kv_grid
    fixture), NOT the tabulated assembly path the product runs, and
  * the vendor's dense solver (torch.linalg.cholesky / cholesky_solve = rocSOLVER) as a TEST cross-check (SURVEY.md
    appendix B allows exactly that use), NOT the hand-written blocked factorisation.

So the product's whole chain at N = 40 000 -- Hilbert layout, tables + exact pass, grouped (G = 4) MFMA Cholesky, fused
forward sweep, reductions -- is compared with an independent chain on the same inputs.  At N = 100 000 the dense matrix
(80 GB) is not formed: sampled rows of the factor are checked through (L L^T)[r, c] = Sigma[r, c].
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _handle(pb, **opts):
    from sif_xco2_cokriging_amd import native
    pv = pb["params"]
    h = native.Handle(0)
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(pb["metric"])
    for k in range(2):
        h.set_data(k, pb["coords"][k], pb["values"][k])
    for k, v in opts.items():
        h.set_option(k, v)
    return h


def _exact_sigma_gpu(h, coords, chunk=4000):
    """Dense Sigma = [[C11, C12], [C12^T, C22]] (src/joint_prediction.py:124-153) as a torch fp64 CUDA tensor, every
    entry from the exact evaluator (ck_cov_dense), row chunks staged through the host."""
    import torch
    n0, n1 = len(coords[0]), len(coords[1])
    N = n0 + n1
    S = torch.empty((N, N), dtype=torch.float64, device="cuda")
    for (i, j, r0, c0) in ((0, 0, 0, 0), (0, 1, 0, n0), (1, 1, n0, n0)):
        A, B = coords[i], coords[j]
        for a in range(0, len(A), chunk):
            blk = torch.from_numpy(h.cov_dense(i, j, A[a:a + chunk], B)).cuda()
            S[r0 + a:r0 + a + blk.shape[0], c0:c0 + len(B)] = blk
            if i != j:
                S[c0:c0 + len(B), r0 + a:r0 + a + blk.shape[0]] = blk.T       # :150-151: the transposed copy
    return S


def test_n40000_prediction_vs_exact_entries_and_vendor_solver():
    import torch
    from sif_xco2_cokriging_amd import synth
    pb = synth.conus_problem(20000, seed=20003)          # the bench's workload (BASELINE configs[2])
    coords, pc = pb["coords"], pb["pcoords"]
    N = 40000
    h = _handle(pb)
    h.assemble_joint()
    assert h.factor() == 0
    cols = np.linspace(0, len(pc) - 1, 64).astype(int)   # 64 of the 8 833 grid points, spread over the grid
    got, full = {}, {}
    for i in (0, 1):
        p, e = h.predict(i, pc)                          # the FULL grid, as the bench does; compared at the sampled points
        got[i] = (p[cols], e[cols])
        full[i] = (p, e)
    assert h.num_panels()[0] == 79                       # grouped sweeps (G = 4, a first group of two) are the default from 40 panels on
    # THE PRODUCT SCHEDULE at this size: what Predictor.__call__, smoke() and the bench's timed step
    # run -- ck_factor_predict with its DEFAULT options: 79 panels, groups of four behind a first group of two, one sweep over the tall matrix
    # [Sigma; c0^T; z^T] with the look-ahead on two streams -- on a fresh handle: the same bits as ck_factor + ck_predict
    # above on all 8 833 points, hence the same distance to the independent chain below.
    hp = _handle(pb)
    hp.assemble_joint()
    info, pp, pe = hp.factor_predict(0, pc)
    tp = hp.timings()
    assert info == 0 and hp.num_panels()[0] == 79 and tp["fused_sweeps_ms"] > 0 and tp["panel_coop_redone"] == 0
    assert np.array_equal(pp, full[0][0]) and np.array_equal(pe, full[0][1])
    pp1, pe1 = hp.predict(1, pc)                         # on the factor the product schedule left resident
    assert np.array_equal(pp1, full[1][0]) and np.array_equal(pe1, full[1][1])
    # ... and round 3's form of the same call (two overlapped sweeps, three streams) for the record
    hp.set_option("tall_sweep", 0)
    hp.assemble_joint()
    info, pq, pqe = hp.factor_predict(0, pc)
    assert info == 0 and hp.timings()["fused_sweeps_ms"] > 0
    assert np.array_equal(pq, full[0][0]) and np.array_equal(pqe, full[0][1])
    print(f"N = 40 000: ck_factor_predict (defaults: tall sweep, G = 4 behind a first group of 2, look-ahead; {tp['fused_sweeps_ms']:.1f} ms) == "
          "ck_factor + ck_predict bit for bit on 8 833 points, both processes")
    hp.close()
    S = _exact_sigma_gpu(h, coords)
    z = torch.from_numpy(np.concatenate(pb["values"])).cuda()
    L = torch.linalg.cholesky(S)                         # :69 cho_factor
    del S
    pv = pb["params"]
    for i in (0, 1):
        # c0 (:104-122): covariance of process i (nugget where h == 0) against its own sites, cross-covariance otherwise
        c0 = np.vstack([h.cov_dense(i, j, coords[j], pc[cols], use_nugget=True) if j == i
                        else h.cov_dense(min(i, j), max(i, j), coords[j], pc[cols]) if i < j
                        else h.cov_dense(min(i, j), max(i, j), pc[cols], coords[j]).T
                        for j in (0, 1)])
        assert c0.shape == (N, len(cols))
        c0 = torch.from_numpy(np.ascontiguousarray(c0)).cuda()
        W = torch.cholesky_solve(c0, L)                  # :68-73 cho_solve -> Sigma^-1 c0
        pred = (W.T @ z).cpu().numpy()                   # :77
        var = (pv[i] ** 2 + pv[8 + i]) - (W * c0).sum(0)  # :74 diagonal of C_pp - W c0
        err = np.nan_to_num(np.sqrt(var.cpu().numpy()))  # :78
        gp, ge = got[i]
        rel_p = np.max(np.abs(gp - pred)) / np.max(np.abs(pred))
        rel_e = np.max(np.abs(ge - err)) / np.max(np.abs(err))
        print(f"N = 40 000, process {i}: 64 grid points, max rel. diff pred {rel_p:.2e}, pred_err {rel_e:.2e}")
        assert rel_p < 1e-8 and rel_e < 1e-8
    h.close()


def test_n40000_table_path_entries_on_the_lattice_vs_exact_evaluator():
    """3 x 10^6 sampled entries of the ASSEMBLED Sigma (the tabulated path the product runs, on config 3's 0.05-degree
    lattice geometry: thousands of pairs share a distance, neighbours sit 5.5 km apart) against the exact evaluator:
    5e-13 max(|C|, 1e-6 sigma^2), the bound of DESIGN.md section 6 -- at the headline size, not on a few hundred sites."""
    from sif_xco2_cokriging_amd import synth
    pb = synth.conus_problem(20000, seed=20003)
    coords = pb["coords"]
    n0 = len(coords[0])
    h = _handle(pb)
    h.assemble_joint()
    assert all(h.table_info(b)["enabled"] for b in range(3))
    rng = np.random.default_rng(7)
    pv = pb["params"]
    amp = {(0, 0): pv[0] ** 2, (0, 1): abs(pv[10]) * pv[0] * pv[1], (1, 1): pv[1] ** 2}
    worst = 0.0
    for (i, j) in ((0, 0), (0, 1), (1, 1)):
        # a random 1000 x 1000 cross product, plus 1000 rows against their 1000 NEAREST-index neighbours (short lags)
        for near in (False, True):
            ri = rng.choice(len(coords[i]), 1000, replace=False)
            ci = (ri[:1] + np.arange(1000)) % len(coords[j]) if near else rng.choice(len(coords[j]), 1000, replace=False)
            exact = h.cov_dense(i, j, coords[i][ri], coords[j][ci], use_nugget=True)
            R, C = np.meshgrid(ri + (n0 if i == 1 else 0), ci + (n0 if j == 1 else 0), indexing="ij")
            got = h.debug_get_entries(R.ravel(), C.ravel()).reshape(1000, 1000)
            tol = 5e-13 * np.maximum(np.abs(exact), 1e-6 * amp[i, j])
            assert np.all(np.abs(got - exact) <= tol), (i, j, near, float(np.max(np.abs(got - exact) / tol)))
            worst = max(worst, float(np.max(np.abs(got - exact) / tol)))
    print(f"table-path Sigma on the lattice, 6e6 entries: worst |diff| = {worst:.3f} of the 5e-13 bound; "
          f"exact-pass entries: {h.table_fallbacks()}")
    h.close()


def test_n100000_factor_rows_vs_exact_entries():
    """BASELINE configs[3]'s size on one GPU: 24 sampled rows of L (from all parts of the matrix, incl. the last
    panel) against Sigma through (L L^T)[r, c] = Sigma[r, c] for every pair of sampled rows -- each row of L depends on
    every column to its left, i.e. on the whole factorisation before it."""
    from sif_xco2_cokriging_amd import synth
    n = 50000
    pb = synth.conus_problem(n, seed=20004)
    coords = pb["coords"]
    N = 2 * n
    h = _handle(pb)
    h.assemble_joint()
    assert h.factor() == 0
    rng = np.random.default_rng(11)
    rows = np.sort(np.concatenate([rng.choice(N, 20, replace=False), [0, n - 1, n, N - 1]]))
    # full rows of L in the caller's order: L[r, c] for all c (entries above the diagonal of the INTERNAL order come back
    # as L[c', r'] -- mask them out by asking in internal order instead: use the identity (L L^T) = Sigma on caller indices)
    # (L L^T)[r, s] = sum_c L[r, c] L[s, c] over internal columns c <= min(r', s'): rows are fetched against ALL sites and
    # entries with internal column beyond the row's own position are zeroed via the factor's triangular structure.
    perm = [h.debug_site_order(k, len(coords[k])) for k in range(2)]
    n0p = -(-n // 64) * 64
    inv = np.empty(N, dtype=np.int64)           # caller's stacked index -> internal position
    for k in range(2):
        inv[(k * n) + perm[k]] = (0 if k == 0 else n0p) + np.arange(len(perm[k]))
    allc = np.arange(N, dtype=np.int64)
    Lrows = []
    for r in rows:
        v = h.debug_get_entries(np.full(N, r, dtype=np.int64), allc)   # L[max(r', c'), min(r', c')]
        v[inv[allc] > inv[r]] = 0.0                                     # keep the row: internal column <= internal row
        Lrows.append(v)
    Lrows = np.array(Lrows)
    G = Lrows @ Lrows.T
    stacked = np.vstack(coords)
    proc = (rows >= n).astype(int)
    want = np.empty_like(G)
    for a, (ra, pa) in enumerate(zip(rows, proc)):
        for b, (rb, pb_) in enumerate(zip(rows, proc)):
            i, j = min(pa, pb_), max(pa, pb_)
            A, B = (stacked[ra:ra + 1], stacked[rb:rb + 1]) if pa <= pb_ else (stacked[rb:rb + 1], stacked[ra:ra + 1])
            want[a, b] = h.cov_dense(i, j, A, B, use_nugget=True)[0, 0]
    scale = np.sqrt(np.outer(np.diag(want), np.diag(want)))
    rel = np.max(np.abs(G - want) / scale)
    print(f"N = 100 000: (L L^T)[r, s] vs exact Sigma[r, s] on {len(rows)} sampled rows, max |diff| / sqrt(S_rr S_ss) = {rel:.2e}")
    assert rel < 1e-10
    h.close()
