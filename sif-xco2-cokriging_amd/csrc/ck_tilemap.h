// ck_tilemap.h -- linear workgroup id -> 128 x 128 tile of a Cholesky trailing update, WITHOUT empty workgroups.
//
// A trailing update (k_syrk_group_d) touches, in every block column J = J0 + u * Jstep (u < nJ; NB = 512 columns = 4 tile
// columns), the tiles (tm, tn) on or below the diagonal whose rows and columns lie in front of the identity padding:
// tm < tv(J) = ceil(nvalid / 128) - 4 J, tn <= min(tm, 3).  Rounds 1-3 launched a rectangle (tile rows of column J0 x 4,
// one grid row per block column) and let the workgroups above the diagonal, below a shorter column's end or inside the
// padding return at once: HALF the grid.  Stamps of the workgroups' lifetimes (scripts/diag_gemm_occupancy.py) showed what
// that costs: the dispatcher needs ~0.15 us per workgroup whether it works or not, the slots a finished tile frees wait
// for it, and at N = 40 000 only 455 of the chip's 512 slots were occupied on average (429 in the middle of the
// factorisation).  With this map the grid is exactly the number of tiles.
//
// Column u holds T(u) = a - d u tiles (a = 4 tv(J0) - 6, d = 16 Jstep) as long as it has at least four tile rows; only the
// matrix's last block column can be shorter (tv = 1, 2, 3 -> 1, 3, 6 tiles).  Tiles are numbered column by column, inside a
// column row by row (the order of the old grid), so S(u) = u a - (d / 2) u (u - 1) tiles precede column u and u follows
// from t by a square root, corrected in integers.
//
// Host and device use the same code (tests/test_tilemap.py enumerates whole launches through ck_debug_tile_map).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CK_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define CK_HD static inline
#endif

struct CkTileMap {
    int J0, Jstep;
    int nfull;        // leading columns with >= 4 valid tile rows
    int tv_last;      // valid tile rows (1..3) of the one short column behind them, 0 if there is none
    int a, d;         // tiles of the first column; a full column has d tiles fewer than the one before it
    long long nfull_tiles, total;
};

CK_HD long long ck_tilemap_before(const CkTileMap& m, long long u) { return u * m.a - (long long)(m.d / 2) * u * (u - 1); }

// nvalid: rows / columns from here on are identity padding.  Columns without a valid row are dropped.
CK_HD CkTileMap ck_tilemap_make(long long nvalid, int J0, int Jstep, int nJ) {
    CkTileMap m;
    m.J0 = J0;
    m.Jstep = Jstep;
    m.nfull = 0;
    m.tv_last = 0;
    const long long R0v = (nvalid + 127) / 128;
    const long long tv0 = R0v - 4LL * J0;
    m.a = (int)(4 * tv0 - 6);
    m.d = 16 * Jstep;
    for (int u = 0; u < nJ; ++u) {
        const long long tv = tv0 - 4LL * Jstep * u;
        if (tv >= 4)
            m.nfull = u + 1;
        else {
            if (tv > 0) m.tv_last = (int)tv;
            break;
        }
    }
    m.nfull_tiles = ck_tilemap_before(m, m.nfull);
    m.total = m.nfull_tiles + (long long)m.tv_last * (m.tv_last + 1) / 2;
    return m;
}

// tile t (0 <= t < m.total) -> column index u (block column J0 + u Jstep), tile row tm and tile column tn inside it
CK_HD void ck_tilemap_get(const CkTileMap& m, long long t, int& u, int& tm, int& tn) {
    long long local;
    if (t >= m.nfull_tiles) {   // the short last column: a plain triangle
        u = m.nfull;
        local = t - m.nfull_tiles;
        long long r = (long long)((sqrt(8.0 * (double)local + 1.0) - 1.0) * 0.5);
        while ((r + 1) * (r + 2) / 2 <= local) ++r;
        while (r * (r + 1) / 2 > local) --r;
        tm = (int)r;
        tn = (int)(local - r * (r + 1) / 2);
        return;
    }
    const double h = 0.5 * m.d, b = (double)m.a + h;
    double disc = b * b - 2.0 * (double)m.d * (double)t;
    if (disc < 0) disc = 0;
    long long uu = (long long)((b - sqrt(disc)) / (double)m.d);
    if (uu < 0) uu = 0;
    if (uu > m.nfull - 1) uu = m.nfull - 1;
    while (uu + 1 < m.nfull && ck_tilemap_before(m, uu + 1) <= t) ++uu;
    while (uu > 0 && ck_tilemap_before(m, uu) > t) --uu;
    u = (int)uu;
    local = t - ck_tilemap_before(m, uu);
    if (local < 1) {
        tm = 0;
        tn = 0;
    } else if (local < 3) {
        tm = 1;
        tn = (int)local - 1;
    } else if (local < 6) {
        tm = 2;
        tn = (int)local - 3;
    } else {
        tm = 3 + (int)((local - 6) >> 2);
        tn = (int)((local - 6) & 3);
    }
}
