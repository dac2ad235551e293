"""The workgroup -> tile map of the Cholesky trailing updates (csrc/ck_tilemap.h, host side through ck_debug_tile_map):
every launch must cover exactly the tiles the rectangular grid of rounds 1-3 did not skip -- on or below the diagonal of
its block column, rows and columns in front of the identity padding -- once each, column by column, row by row."""
import numpy as np
import pytest

from sif_xco2_cokriging_amd import native


def brute(nvalid, J0, Jstep, nJ):
    Npad = (nvalid + 511) // 512 * 512
    out = []
    for y in range(nJ):
        J = J0 + y * Jstep
        M = Npad - J * 512
        for tm in range(max(M, 0) // 128):
            for tn in range(4):
                r0, c0 = tm * 128, tn * 128
                if r0 + 127 < c0:
                    continue
                if J * 512 + r0 >= nvalid or J * 512 + c0 >= nvalid:
                    continue
                out.append((J, tm, tn))
    return np.array(out, dtype=np.int32).reshape(-1, 3)


CASES = []
for nvalid in (1, 100, 128, 129, 300, 384, 385, 512, 513, 700, 1024, 4600, 9984, 10000, 40000, 40064, 100096):
    nK = (nvalid + 511) // 512
    for J0 in sorted({0, 1, 2, nK // 2, nK - 2, nK - 1}):
        if J0 < 0 or J0 >= nK:
            continue
        for Jstep in (1, 2, 3, 8):
            full = (nK - 1 - J0) // Jstep + 1
            for nJ in sorted({1, 2, full - 1, full}):
                if 1 <= nJ <= full:
                    CASES.append((nvalid, J0, Jstep, nJ))


def test_tile_map_covers_the_lower_triangle_exactly_once():
    assert len(CASES) > 300
    for nvalid, J0, Jstep, nJ in CASES:
        got = native.tile_map(nvalid, J0, Jstep, nJ)
        want = brute(nvalid, J0, Jstep, nJ)
        assert got.shape == want.shape, (nvalid, J0, Jstep, nJ, got.shape, want.shape)
        assert np.array_equal(got, want), (nvalid, J0, Jstep, nJ)


def brute_tall(nvalid, J0, nJ, axr):
    """the triangle tiles of every block column followed by the axr x 4 tiles of the right-hand-side block below it (tile
    columns inside the identity padding dropped: their update is exactly zero)"""
    tri = brute(nvalid, J0, 1, nJ)
    out = []
    for J in range(J0, J0 + nJ):
        grp = tri[tri[:, 0] == J]
        out += [(J, int(tm), int(tn), 0) for _, tm, tn in grp]
        if len(grp):
            out += [(J, tm, tn, 1) for tm in range(axr) for tn in range(4) if J * 512 + tn * 128 < nvalid]
    return np.array(out, dtype=np.int32).reshape(-1, 4)


def test_tall_map_appends_the_right_hand_side_tiles_of_every_block_column():
    """ck_factor_predict's launches over the tall matrix [Sigma; c0^T; z^T] (k_tall_group_d): exactly the tiles of
    k_syrk_group_d and k_aux_group_d together, once each, column by column."""
    n = 0
    for nvalid in (1, 100, 129, 385, 512, 513, 700, 1024, 4600, 9984, 10000, 40000, 40064):
        nK = (nvalid + 511) // 512
        for J0 in sorted({0, 1, 2, nK // 2, nK - 2, nK - 1}):
            if J0 < 0 or J0 >= nK:
                continue
            full = nK - J0
            for nJ in sorted({1, 2, 3, full - 1, full}):
                if not 1 <= nJ <= full:
                    continue
                for axr in (0, 1, 2, 70):
                    got = native.tall_map(nvalid, J0, nJ, axr)
                    want = brute_tall(nvalid, J0, nJ, axr)
                    assert got.shape == want.shape and np.array_equal(got, want), (nvalid, J0, nJ, axr)
                    n += 1
    assert n > 400


def test_tile_map_rejects_a_first_column_inside_the_padding():
    with pytest.raises(native.NativeError):
        native.tile_map(1000, 2, 1, 1)


def test_run_map_of_the_batched_local_systems():
    """Every (system, unit) with unit < counts[system] exactly once; padding only at the end of runs (< 8 systems' worth
    each); all units of a system share the workgroup id modulo 8 (one XCD)."""
    rng = np.random.default_rng(5)
    cases = [[5], [3, 3, 3], [0], [], [7, 7, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 0, 0],
             sorted(rng.integers(0, 46, 3000).tolist(), reverse=True),
             sorted((rng.integers(1, 300, 500) * 7).tolist(), reverse=True),   # more distinct counts than runs in the table
             sorted([t * (t + 1) // 2 for t in rng.integers(0, 12, 9000)], reverse=True)]
    for counts in cases:
        counts = np.asarray(counts, dtype=np.int32)
        got = native.run_map(counts)
        real = got[got[:, 0] >= 0]
        ok = real[:, 1] < counts[real[:, 0]] if len(real) else np.zeros(0, bool)
        pairs = real[ok]
        want = np.array([(y, t) for y in range(len(counts)) for t in range(counts[y])], dtype=np.int32).reshape(-1, 2)
        a = pairs[np.lexsort((pairs[:, 1], pairs[:, 0]))]
        assert np.array_equal(a, want), counts[:10]
        if len(want):
            ids = np.nonzero(got[:, 0] >= 0)[0][ok]
            for y in np.unique(pairs[:, 0])[:50]:
                assert len(set(ids[pairs[:, 0] == y] % 8)) == 1
            if len(np.unique(counts[counts > 0])) < 40:
                assert len(got) <= len(want) + 8 * int(counts[counts > 0].astype(np.int64).max()) * len(np.unique(counts[counts > 0]))
    with pytest.raises(native.NativeError):
        native.run_map([1, 2])
