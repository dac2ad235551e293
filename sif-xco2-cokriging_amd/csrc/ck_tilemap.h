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
    int axr;          // "tall" launches (round 4): tile rows of the right-hand-side block that hangs below every block column
                      // ([Sigma; c0^T; z^T] is ONE tall matrix: k_tall_group_d), 0 = the triangle alone
    long long nfull_tiles, total;
};

// tiles in front of column u: the triangle's a - d u' plus 4 axr right-hand-side tiles per column u' < u
CK_HD long long ck_tilemap_before(const CkTileMap& m, long long u) {
    return u * (m.a + 4LL * m.axr) - (long long)(m.d / 2) * u * (u - 1);
}

// nvalid: rows / columns from here on are identity padding.  Columns without a valid row are dropped.
// aux_tile_rows > 0: every column is followed by its aux_tile_rows x 4 right-hand-side tiles (x tv_last in the short column:
// the tile columns inside the identity padding have an exactly zero update)
CK_HD CkTileMap ck_tilemap_make(long long nvalid, int J0, int Jstep, int nJ, int aux_tile_rows = 0) {
    CkTileMap m;
    m.J0 = J0;
    m.Jstep = Jstep;
    m.nfull = 0;
    m.tv_last = 0;
    m.axr = aux_tile_rows;
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
    m.total = m.nfull_tiles + (long long)m.tv_last * (m.tv_last + 1) / 2 + (long long)m.tv_last * m.axr;
    return m;
}

// tile t (0 <= t < m.total) -> column index u (block column J0 + u Jstep), tile row tm and tile column tn inside it;
// returns true for a tile of the column's right-hand-side block (tm, tn count inside that block then)
CK_HD bool ck_tilemap_get(const CkTileMap& m, long long t, int& u, int& tm, int& tn) {
    long long local;
    if (t >= m.nfull_tiles) {   // the short last column: a plain triangle (+ tv_last tile columns of right-hand-side tiles)
        u = m.nfull;
        local = t - m.nfull_tiles;
        const long long tri = (long long)m.tv_last * (m.tv_last + 1) / 2;
        if (local >= tri) {
            local -= tri;
            tm = (int)(local / m.tv_last);
            tn = (int)(local - (long long)tm * m.tv_last);
            return true;
        }
        long long r = (long long)((sqrt(8.0 * (double)local + 1.0) - 1.0) * 0.5);
        while ((r + 1) * (r + 2) / 2 <= local) ++r;
        while (r * (r + 1) / 2 > local) --r;
        tm = (int)r;
        tn = (int)(local - r * (r + 1) / 2);
        return false;
    }
    const double h = 0.5 * m.d, b = (double)m.a + 4.0 * m.axr + h;
    double disc = b * b - 2.0 * (double)m.d * (double)t;
    if (disc < 0) disc = 0;
    long long uu = (long long)((b - sqrt(disc)) / (double)m.d);
    if (uu < 0) uu = 0;
    if (uu > m.nfull - 1) uu = m.nfull - 1;
    while (uu + 1 < m.nfull && ck_tilemap_before(m, uu + 1) <= t) ++uu;
    while (uu > 0 && ck_tilemap_before(m, uu) > t) --uu;
    u = (int)uu;
    local = t - ck_tilemap_before(m, uu);
    const long long tsig = (long long)m.a - (long long)m.d * uu;   // the column's triangle tiles come first
    if (local >= tsig) {
        local -= tsig;
        tm = (int)(local >> 2);
        tn = (int)(local & 3);
        return true;
    }
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
    return false;
}

// ---------------------------------------------------------------------------------------------------------------
// Batched systems of different sizes (the local predictor's tiled path): workgroup id -> (system, unit inside it)
// ---------------------------------------------------------------------------------------------------------------
// The systems of a batch are sorted by size, largest first, so the number of work units a launch has for system y
// (128 x 128 tiles of its trailing matrix, 64-row chunks below a column group) never grows with y and takes few
// distinct values: runs of systems with the same count.  Rounds 1-3 launched (units of the LARGEST system) x (systems)
// and let the surplus return at once -- at 400 km most of the grid.  Here a launch has, per run, exactly
// count x (systems of the run, rounded up to 8) workgroups; the table of runs travels as a kernel argument.
// Inside a run the workgroups are dealt out so that all units of a system have the same id modulo 8, i.e. run on one
// XCD and share its L2 (workgroups go to the XCDs round-robin by linear id; run offsets are multiples of 8).
#define CK_RUN_MAX 48
struct CkRunMap {
    int nruns;
    int off[CK_RUN_MAX + 1];   // first workgroup of run r; off[nruns] = the grid
    int y0[CK_RUN_MAX];        // first system of run r
    int n[CK_RUN_MAX];         // systems in run r
    int c[CK_RUN_MAX];         // workgroups per system in run r (the largest count inside the run)
};

// count(y) for y < n_sys, non-increasing; systems with count 0 (a suffix) get no workgroups.  More than CK_RUN_MAX
// distinct counts: the last run takes the rest with its largest count (the kernels keep their own bound checks).
template <class F>
static inline CkRunMap ck_runmap_make(int n_sys, F count) {
    CkRunMap m;
    m.nruns = 0;
    m.off[0] = 0;
    int y = 0;
    while (y < n_sys) {
        const int c = count(y);
        if (c <= 0) break;
        int e = y + 1;
        if (m.nruns == CK_RUN_MAX - 1)
            while (e < n_sys && count(e) > 0) ++e;
        else
            while (e < n_sys && count(e) == c) ++e;
        const int r = m.nruns++;
        m.y0[r] = y;
        m.n[r] = e - y;
        m.c[r] = c;
        m.off[r + 1] = m.off[r] + c * ((e - y + 7) / 8 * 8);
        y = e;
    }
    return m;
}

// workgroup b -> system y and unit t; false: a padding workgroup at the end of a run (at most 7 systems' worth)
CK_HD bool ck_runmap_get(const CkRunMap& m, int b, int& y, int& t) {
    int r = 0;
    while (r + 1 < m.nruns && b >= m.off[r + 1]) ++r;
    const int L = b - m.off[r], c = m.c[r];
    const int yl = 8 * (L / (8 * c)) + (L & 7);
    t = (L >> 3) % c;
    y = m.y0[r] + yl;
    return yl < m.n[r];
}
