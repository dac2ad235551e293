"""The host-side Hilbert ordering of the sites (ck_hilbert_order: no device needed) against a numpy restatement:
Hilbert key of order 16 on the sites' bounding box, stable sort.  Covers the comparison-sort path (n < 4 096) and the
threaded three-pass radix sort above it, duplicate sites, and degenerate boxes."""
import numpy as np
import pytest


def _hilbert_key(x, y):
    x = x.astype(np.uint64).copy()
    y = y.astype(np.uint64).copy()
    d = np.zeros_like(x)
    s = np.uint64(32768)
    while s > 0:
        rx = ((x & s) > 0).astype(np.uint64)
        ry = ((y & s) > 0).astype(np.uint64)
        d += s * s * ((np.uint64(3) * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, s - np.uint64(1) - x, x)
        y = np.where(flip, s - np.uint64(1) - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s = s >> np.uint64(1)
    return d


def _reference_order(c):
    lo, hi = c.min(axis=0), c.max(axis=0)
    f = np.zeros_like(c)
    for k in range(2):
        sc = 65536.0 / (hi[k] - lo[k]) if hi[k] > lo[k] else 0.0
        f[:, k] = np.clip((c[:, k] - lo[k]) * sc, 0.0, 65535.0)
    # the library computes (x - lo) * s in double and truncates: same here
    key = _hilbert_key(f[:, 0].astype(np.uint32), f[:, 1].astype(np.uint32))
    return np.argsort(key, kind="stable")


@pytest.mark.parametrize("n", [1, 2, 100, 4095, 4096, 100_000, 250_001])
def test_hilbert_order_matches_numpy(n):
    from sif_xco2_cokriging_amd import native
    rng = np.random.default_rng(n)
    c = np.column_stack([rng.uniform(22, 58, n), rng.uniform(-125, -65, n)])
    if n > 10:
        c[n // 3: n // 3 + n // 10] = c[: n // 10]                 # duplicate sites: the caller's order within a cell
        c[-(n // 20):] = np.round(c[-(n // 20):] * 4) / 4          # and lattice points sharing cells
    perm = native.hilbert_order(c)
    assert sorted(perm.tolist()) == list(range(n)) if n <= 4096 else np.array_equal(np.sort(perm), np.arange(n))
    assert np.array_equal(perm, _reference_order(c))


def test_hilbert_order_degenerate_boxes():
    from sif_xco2_cokriging_amd import native
    n = 5000
    same = np.tile([[40.0, -100.0]], (n, 1))
    assert np.array_equal(native.hilbert_order(same), np.arange(n))          # one cell: the caller's order
    line = np.column_stack([np.full(n, 40.0), np.linspace(-120, -70, n)])
    assert np.array_equal(native.hilbert_order(line), _reference_order(line))
    assert native.hilbert_order(np.empty((0, 2))).shape == (0,)
