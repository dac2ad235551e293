// ck_cov.hip -- covariance assembly kernels (pairwise distance -> Matern auto/cross-covariance).
//
// Replaces, entry for entry, the reference's
//   fields.distance_matrix                     src/fields.py:318-342
//   MultivariateMatern.covariance / cross_covariance  src/model.py:193-207 (+ :354-385)
//   Predictor._joint_cov / _pred_cross_cov     src/joint_prediction.py:104-153
// without ever materialising a distance matrix: every output tile is produced from the two
// coordinate slices it depends on and written exactly once, straight into the packed
// block-column layout the factorisation works on.
#include "ck_internal.h"

typedef double d2_t __attribute__((ext_vector_type(2)));

// per-site transform -------------------------------------------------------------------
// c0/c1/c2: (lat_rad, lon_rad, cos lat) | (x, y, 0) for the exact distance formulas;
// u (optional, 3 x n SoA): unit vector on the sphere | (x, y, 0) -- the chord vectors of the table path.
// one site: (a, b) = the caller's coordinates -> c = exact-formula form, v = chord vector
__device__ __forceinline__ void prep_site(int metric, double a, double b, double (&c)[3], double (&v)[3]) {
    if (metric == CK_METRIC_HAVERSINE) {
        const double lat = a * CK_DEG2RAD, lon = b * CK_DEG2RAD;   // numpy.radians (fields.py:334-335)
        const double cl = cos(lat);
        c[0] = lat;
        c[1] = lon;
        c[2] = cl;
        v[0] = cl * cos(lon);
        v[1] = cl * sin(lon);
        v[2] = sin(lat);
    } else {
        c[0] = a;
        c[1] = b;
        c[2] = 0.0;
        v[0] = a;
        v[1] = b;
        v[2] = 0.0;
    }
}

__global__ void k_prep_sites(const double* __restrict__ coords, long n, int metric, double* __restrict__ c0,
                             double* __restrict__ c1, double* __restrict__ c2, double* __restrict__ u) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double c[3], v[3];
    prep_site(metric, coords[2 * i], coords[2 * i + 1], c, v);
    c0[i] = c[0];
    c1[i] = c[1];
    c2[i] = c[2];
    if (u) {
        u[i] = v[0];
        u[n + i] = v[1];
        u[2 * n + i] = v[2];
    }
}

void ck_launch_prep_sites(hipStream_t s, const double* coords, int64_t n, int metric, double* c0, double* c1,
                          double* c2, double* u) {
    if (n <= 0) return;
    k_prep_sites<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(coords, n, metric, c0, c1, c2, u);
}

__device__ __forceinline__ double pair_dist(int metric, double a0, double a1, double a2, double b0, double b1,
                                            double b2) {
    return metric == CK_METRIC_HAVERSINE ? ck_haversine_km(a0, a1, a2, b0, b1, b2) : ck_euclid(a0, a1, b0, b1);
}

// =========================================================================================
// Assembly of Sigma block columns and of the right-hand-side rows
// =========================================================================================
// Internal site order (CkLayout): process 0 occupies [0, n0), padded to n0p = roundup(n0, 64);
// process 1 occupies [n0p, nend); everything else up to npad is padding.  Because n0p is a
// multiple of the 64 x 64 tile, every tile belongs to exactly ONE Matern block (11, 12 or 22):
// no tile ever needs a per-entry block choice.  Entries that involve a padding index form an
// identity (Sigma) or are zero (right-hand sides), so the factorisation of the padded matrix is
// the factorisation of the real one.
//
// One launch covers all panels: a workgroup (256 threads) owns one 64-row x 512-column strip of
// a panel and walks its eight 64 x 64 sub-tiles (assemble_subtile).
struct CkSiteRef {   // SoA views of one site set
    const double *c0, *c1, *c2;   // exact-formula coordinates (lat_rad, lon_rad, cos lat | x, y, 0)
    const double *u0, *u1, *u2;   // chord vectors (table path)
};

__device__ __forceinline__ bool site_valid(const CkLayout& L, long g) {
    return g < L.n0 || (g >= L.n0p && g < L.nend);
}

// the exact formulas as an out-of-line call: rare in the table kernels (pairs closer than the
// table's lower end or beyond its upper end), and out of line it does not inflate their code
__device__ __noinline__ double exact_entry_call(const CkMatern* m, int metric, int nug, double ac0, double ac1,
                                                double ac2, double bc0, double bc1, double bc2) {
    return ck_cov_entry(*m, pair_dist(metric, ac0, ac1, ac2, bc0, bc1, bc2), nug);
}

// append this thread's deferred entries (bit k of `mask` = entry k of the sub-tile) to the worklist
__device__ __noinline__ void worklist_append(const CkWorklist& wl, unsigned mask, int r0, int rstride, int c0) {
    const int lane = threadIdx.x & 63;
    for (int k = 0; k < 16; ++k) {
        const bool f = (mask >> k) & 1u;
        const unsigned long long sl = __builtin_amdgcn_ballot_w64(f);
        if (sl == 0ULL) continue;
        const int leader = __builtin_ctzll(sl);
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(wl.count, (unsigned)__builtin_popcountll(sl));
        base = __shfl(base, leader);
        if (f) {
            const unsigned slot = base + (unsigned)__builtin_popcountll(sl & ((1ULL << lane) - 1ULL));
            // entry k = (b, a, e): column 32 b + e, row a * rstride
            const int b = k >> 3, a = (k >> 1) & 3, e = k & 1;
            if (slot < wl.cap) wl.items[slot] = make_int2(r0 + a * rstride, c0 + 32 * b + e);
        }
    }
}

// One 64 x 64 sub-tile.  thread (ty, tx) computes rows ty + 16 a (a = 0..3), cols 2 tx + {0, 1} +
// 32 b (b = 0, 1): every store instruction of a wave writes 4 rows x 256 contiguous bytes.
//
// Table path: one entry = squared chord, interval index + centre from its bit pattern, 8 LDS
// reads, 7 FMA.  Pairs outside the table's range (closer than its lower end -- which includes
// h == 0 and with it the nugget -- or beyond its upper end) are NOT evaluated here: their bit goes
// into the returned mask and the caller appends (row, col) to a worklist that k_assemble_fix
// evaluates with the exact formulas afterwards; what this kernel stores for them is overwritten.
// Keeping the Bessel code out of this kernel halves its register count.
//
// Interior sub-tiles (no padding row or column; all but a few per panel), table path only.
// ru*: the chord vectors of this thread's four rows.
// ZROWS (right-hand sides, the one row tile per block column that holds row m): rows with 16 a >= zrel are not
// prediction sites -- row m (16 a == zrel) carries the data values z, the rows behind it zeros; they are computed
// like any other row (their coordinates are zero padding, finite) and replaced in front of the store, so that
// this tile runs the same code as the interior ones instead of the entry-by-entry edge path (which made these
// 79 tiles the tail of the launch).
template <bool ZROWS>
__device__ __forceinline__ unsigned assemble_subtile_interior(const double* lcoef, int tbase, unsigned tn,
                                                              const double (&ru0)[4], const double (&ru1)[4],
                                                              const double (&ru2)[4], const CkSiteRef& S, long ct,
                                                              double* __restrict__ obase, int ty, int tx,
                                                              int zrel = 0, const double* __restrict__ z = nullptr) {
    unsigned slowmask = 0;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const long c = ct + 2 * tx + 32 * b;
        const d2_t cu0 = *reinterpret_cast<const d2_t*>(S.u0 + c), cu1 = *reinterpret_cast<const d2_t*>(S.u1 + c),
                   cu2 = *reinterpret_cast<const d2_t*>(S.u2 + c);
        // Two rows x two columns at a time: first the four squared chords, interval indices and
        // offsets from the interval centres; then all 32 coefficient reads (k-major, so they return
        // in the order Horner consumes them); then the four Horner chains interleaved.  The
        // scheduling barriers keep hipcc from sinking each read next to its FMA, which would expose
        // one LDS latency per Horner step.
#pragma unroll
        for (int ap = 0; ap < 2; ++ap) {
            double y[4], cf[CK_TAB_DEG + 1][4];
            const double* lp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int a = 2 * ap + (j >> 1), e = j & 1;
                const double dx = ru0[a] - cu0[e], dy = ru1[a] - cu1[e], dz = ru2[a] - cu2[e];
                const double q = dx * dx + dy * dy + dz * dz;
                int iv;
                y[j] = ck_table_y(q, &iv, tbase);
                bool slow = (unsigned)iv >= tn;                  // q == 0 (and with it the nugget) lands here
                if (ZROWS) slow = slow && 16 * a < zrel;         // the z row and the zero rows never go to the exact pass
                slowmask |= slow ? (1u << ((b * 4 + a) * 2 + e)) : 0u;
                lp[j] = lcoef + min(max(iv, 0), (int)tn - 1);    // keep the lookup inside the table
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = CK_TAB_DEG; k >= 0; --k)
#pragma unroll
                for (int j = 0; j < 4; ++j) cf[k][j] = lp[j][k * CK_TAB_STRIDE];
            __builtin_amdgcn_sched_barrier(0);
            double p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = cf[CK_TAB_DEG][j];
#pragma unroll
            for (int k = CK_TAB_DEG - 1; k >= 0; --k)
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] = fma(p[j], y[j], cf[k][j]);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                d2_t v;
                v[0] = p[2 * h2];
                v[1] = p[2 * h2 + 1];
                if (ZROWS && 16 * (2 * ap + h2) >= zrel) {
                    const d2_t zv = *reinterpret_cast<const d2_t*>(z + c);
                    const d2_t zero = {0.0, 0.0};
                    v = (16 * (2 * ap + h2) == zrel) ? zv : zero;
                }
                // non-temporal: 6.4 GB of panels stream out and are next read by the factorisation, long after they
                // have left every cache (measured -3 % on the kernel against plain stores)
                __builtin_nontemporal_store(v, reinterpret_cast<d2_t*>(obase + (ty + 16 * (2 * ap + h2)) * CK_NB + 2 * tx + 32 * b));
            }
        }
    }
    return slowmask;
}

// Sub-tiles that touch padding rows or columns (identity / z row / zeros), and every sub-tile of
// the exact path: entry by entry, loops kept rolled so that this rarely-run code stays small in
// registers.  AUX: rows are prediction sites (row m = data values z, rows > m zero).
template <bool FAST, bool AUX>
__device__ __forceinline__ unsigned assemble_subtile_edge(const CkMatern& mb, int metric, const double* lcoef,
                                                          int tbase, unsigned tn, const CkSiteRef& R,
                                                          const CkSiteRef& S, const double* __restrict__ z,
                                                          const CkLayout& L, long rt, long ct, long m, int nug,
                                                          double* __restrict__ obase, int ty, int tx,
                                                          const double* rowu = nullptr /* LDS: chord vectors of the strip's
                                                          64 rows when the launch transforms them itself (R.u* is then being
                                                          WRITTEN by the strips of block column 0 of the same launch) */) {
    unsigned slowmask = 0;
#pragma unroll 1
    for (int ba = 0; ba < 8; ++ba) {
        const int b = ba >> 2, a = ba & 3;
        const long r = rt + ty + 16 * a;
        const long c = ct + 2 * tx + 32 * b;
        const bool rv = AUX ? (r < m) : site_valid(L, r);
        d2_t v;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool cv = site_valid(L, c + e);
            const bool valid = rv && cv;
            double val = 0.0;
            if (FAST) {
                const double r0_ = rowu ? rowu[ty + 16 * a] : R.u0[r], r1_ = rowu ? rowu[64 + ty + 16 * a] : R.u1[r],
                             r2_ = rowu ? rowu[128 + ty + 16 * a] : R.u2[r];
                const double dx = r0_ - S.u0[c + e], dy = r1_ - S.u1[c + e], dz = r2_ - S.u2[c + e];
                const double q = dx * dx + dy * dy + dz * dz;
                int iv;
                const double y = ck_table_y(q, &iv, tbase);
                const bool slow = valid && (unsigned)iv >= tn;   // padding lanes never ask for the exact pass
                slowmask |= slow ? (1u << (ba * 2 + e)) : 0u;
                val = ck_table_poly(lcoef, min(max(iv, 0), (int)tn - 1), y);
            } else if (__builtin_amdgcn_ballot_w64(valid) != 0ULL) {
                if (valid)
                    val = exact_entry_call(&mb, metric, nug, R.c0[r], R.c1[r], R.c2[r], S.c0[c + e], S.c1[c + e],
                                           S.c2[c + e]);
            }
            if (AUX) {
                // rows: prediction sites | z | zero padding; padded columns are zero
                if (!rv) val = (r == m) ? z[c + e] : 0.0;
                if (!cv) val = 0.0;
            } else {
                if (!valid) val = (r == c + e) ? 1.0 : 0.0;   // padding: identity
            }
            v[e] = val;
        }
        *reinterpret_cast<d2_t*>(obase + (ty + 16 * a) * CK_NB + 2 * tx + 32 * b) = v;
    }
    return slowmask;
}

__device__ __forceinline__ bool range_has_padding(const CkLayout& L, long g0) {   // [g0, g0 + 64)
    return !(g0 + 64 <= L.n0 || (g0 >= L.n0p && g0 + 64 <= L.nend));
}

// FAST: table path (every block's table enabled) | exact per-entry Bessel evaluation.
// AUX:  rows are prediction sites (row m = data values z, rows > m zero) | rows are data sites.
template <bool FAST, bool AUX>
__global__ __launch_bounds__(256, FAST ? 3 : 1) void k_assemble(const CkMatern* __restrict__ blk,
                                                                 const CkTable* __restrict__ tabs,
                                                                 const double* const* __restrict__ coefs, int metric,
                                                                 int i_pred, CkSiteRef R, long m, CkSiteRef S,
                                                                 const double* __restrict__ z, CkLayout L,
                                                                 CkPanelMap pm, CkWorklist wl,
                                                                 const double* __restrict__ raw = nullptr, long mpad = 0,
                                                                 double* __restrict__ pc_out = nullptr,
                                                                 double* __restrict__ pu_out = nullptr,
                                                                 unsigned* __restrict__ queue = nullptr, int n_strips = 0,
                                                                 int csubs = CK_NB / 64) {
    __shared__ double lcoef[FAST ? (CK_TAB_DEG + 1) * CK_TAB_STRIDE : 1];
    __shared__ double srow[(FAST && AUX) ? 3 * 64 : 1];   // chord vectors of this strip's 64 rows (raw != nullptr)
    __shared__ unsigned s_next;
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    int loaded = -1, tbase = 0;
    unsigned tn = 1;
    // Work units: chunks of csubs sub-tiles of a 64 x 512 strip.  queue == nullptr: one strip per workgroup (blockIdx.x).
    // queue != nullptr (option "assemble_queue"): a resident set of workgroups takes chunks from a counter -- the first one by
    // its own id, the next one fetched while the current one is computed -- so that the 48 KB table is loaded once per
    // workgroup instead of once per strip and a small launch is balanced to a chunk, not to a strip.
    const int per = (CK_NB / 64) / csubs;
    const unsigned n_chunks = queue ? (unsigned)n_strips * (unsigned)per : gridDim.x;
    unsigned cur = blockIdx.x;
    while (cur < n_chunks) {
    if (queue) {
        __syncthreads();   // the previous chunk's readers of srow / s_next are done
        if (t == 0) s_next = gridDim.x + atomicAdd(queue, 1u);
    }
    const unsigned strip = queue ? cur / (unsigned)per : cur;
    const int sub_first = queue ? (int)(cur % (unsigned)per) * csubs : 0;
    const unsigned n_all = queue ? (unsigned)n_strips : gridDim.x;
    // `strip` enumerates the 64-row tiles of ALL panels of this launch; a workgroup walks the
    // eight 64 x 64 sub-tiles of its 64 x 512 strip
    long row0, col0;
    double* out;
    long tile;
    if (AUX) {   // every panel has mpad / 64 row tiles; row-tile-major, LAST row tile first: the tiles that hold the z row
                 // (and, with padding rows behind them, take the edge path) are dispatched first instead of forming the tail
        // (work-queue form: block column OUTER, so that the chunks a workgroup takes one after the other -- ids a grid apart --
        // walk through the block columns once and with them through the Matern blocks: the 48 KB table is reloaded twice, not
        // at every other chunk)
        const long tr = queue ? strip % pm.aux_tiles : strip / pm.n_panels;
        const long j = queue ? strip / pm.aux_tiles : strip - tr * pm.n_panels;
        tile = pm.aux_tiles - 1 - tr;
        row0 = 0;
        col0 = j * CK_NB;
        out = pm.aux + j * pm.aux_tiles * 64 * CK_NB;
    } else {     // owned panels, sizes differ: tile0[j] = first tile of the j-th owned panel.  Dispatched back to front: the
                 // last panels are short and mostly padding (edge path), the launch should not end on them
        const int bid = (queue && pm.order) ? pm.order[strip] : (int)(n_all - 1 - strip);
        int j = 0;
        while (j + 1 < pm.n_panels && pm.tile0[j + 1] <= bid) ++j;
        const int K = pm.panel_of[j];
        tile = bid - pm.tile0[j];
        row0 = col0 = (long)K * CK_NB;
        out = pm.sigptr[K];
    }
    const long rt = row0 + tile * 64;
    const int pr = AUX ? i_pred : (int)(rt >= L.n0p);
    const bool row_pad = AUX ? (rt + 64 > m) : range_has_padding(L, rt);
    const int zrel = AUX ? (int)(m - rt) - ty : 0;   // this thread's row a is r = m + (16 a - zrel)
    double ru0[4], ru1[4], ru2[4];   // table path: chord vectors of this thread's four rows
    if (FAST && AUX && raw) {
        // The prediction sites arrive untransformed (round 4): one wave transforms the strip's 64 rows -- the arithmetic of
        // k_prep_sites, the same bits -- instead of the whole launch waiting for a transform launch in front of it; the strips of
        // block column 0 also store the result for the later users (exact pass, _verify_model, further sweeps).
        if (t < 64) {
            const long r = rt + t;
            double c[3], v[3];
            prep_site(metric, raw[2 * r], raw[2 * r + 1], c, v);
            srow[t] = v[0];
            srow[64 + t] = v[1];
            srow[128 + t] = v[2];
            if (col0 == 0) {
                pc_out[r] = c[0];
                pc_out[mpad + r] = c[1];
                pc_out[2 * mpad + r] = c[2];
                pu_out[r] = v[0];
                pu_out[mpad + r] = v[1];
                pu_out[2 * mpad + r] = v[2];
            }
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            ru0[a] = srow[ty + 16 * a];
            ru1[a] = srow[64 + ty + 16 * a];
            ru2[a] = srow[128 + ty + 16 * a];
        }
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const long r = rt + ty + 16 * a;
            ru0[a] = FAST ? R.u0[r] : 0.0;
            ru1[a] = FAST ? R.u1[r] : 0.0;
            ru2[a] = FAST ? R.u2[r] : 0.0;
        }
    }
    for (int sub = sub_first; sub < sub_first + csubs; ++sub) {
        const long ct = col0 + sub * 64;
        const int pc = (int)(ct >= L.n0p);
        const int bidx = pr + pc;
        const int nug = (pr == pc);
        const CkMatern& mb = blk[bidx];
        double* obase = out + (rt - row0) * CK_NB + (ct - col0);
        if (FAST && loaded != bidx) {
            __syncthreads();
            const double* src = coefs[bidx];
            for (int k = t; k < (CK_TAB_DEG + 1) * CK_TAB_STRIDE; k += 256) lcoef[k] = src[k];
            tbase = tabs[bidx].base;
            tn = (unsigned)tabs[bidx].n_int;
            __syncthreads();
            loaded = bidx;
        }
        unsigned slowmask;
        // right-hand sides: the row tile that holds row m runs the interior code with its rows >= m replaced in front of the
        // store (ZROWS; thread row a is row m + (16 a - zrel))
        if (!FAST || (row_pad && !AUX) || range_has_padding(L, ct))
            slowmask = assemble_subtile_edge<FAST, AUX>(mb, metric, lcoef, tbase, tn, R, S, z, L, rt, ct, m, nug, obase, ty, tx,
                                                        (FAST && AUX && raw) ? srow : nullptr);
        else if (AUX && row_pad)
            slowmask = assemble_subtile_interior<true>(lcoef, tbase, tn, ru0, ru1, ru2, S, ct, obase, ty, tx, zrel, z);
        else
            slowmask = assemble_subtile_interior<false>(lcoef, tbase, tn, ru0, ru1, ru2, S, ct, obase, ty, tx);
        if (FAST && __builtin_amdgcn_ballot_w64(slowmask != 0u) != 0ULL)   // rare: hand the pairs to the exact pass
            worklist_append(wl, slowmask, (int)(rt + ty), 16, (int)(ct + 2 * tx));
    }
    if (!queue) break;
    __syncthreads();
    cur = s_next;
    }
}

// Exact evaluation of the worklist entries (see assemble_subtile).  Sigma: entry (r, c) lives in panel
// K = c / NB at sigptr[K] + (r - K NB) NB + (c - K NB); right-hand sides: aux + K mpad NB + r NB + ...
template <bool AUX>
__global__ __launch_bounds__(256) void k_assemble_fix(const CkMatern* __restrict__ blk, int metric, int i_pred,
                                                       CkSiteRef R, CkSiteRef S, CkLayout L, CkWorklist wl,
                                                       double* const* __restrict__ sigptr, double* __restrict__ aux,
                                                       long mpad) {
    const unsigned n = *wl.count < wl.cap ? *wl.count : wl.cap;
    if (wl.reset && blockIdx.x == 0 && threadIdx.x == 0) {   // the next assembly's counters (not read by this launch)
        wl.reset[0] = 0u;   // its worklist count
        wl.reset[1] = 0u;   // its work queue (k_assemble, option "assemble_queue")
    }
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int2 it = wl.items[e];
        const long r = it.x, c = it.y;
        const int pc = c >= L.n0p;
        const int pr = AUX ? i_pred : (int)(r >= L.n0p);
        const double val = ck_cov_entry(blk[pr + pc], pair_dist(metric, R.c0[r], R.c1[r], R.c2[r], S.c0[c], S.c1[c], S.c2[c]),
                                        pr == pc);
        const long K = c / CK_NB;
        if (AUX)
            aux[K * mpad * CK_NB + r * CK_NB + (c - K * CK_NB)] = val;
        else
            sigptr[K][(r - K * CK_NB) * CK_NB + (c - K * CK_NB)] = val;
    }
}

// all owned Sigma panels in one launch
void ck_launch_assemble_sigma(hipStream_t s, bool fast, const CkMatern* blk, const CkTable* tabs,
                              const double* const* coefs, int metric, const double* c, const double* u, CkLayout L,
                              CkPanelMap pm, int total_tiles, CkWorklist wl, int queue_slots) {
    if (total_tiles <= 0) return;
    const int64_t np = L.npad;
    CkSiteRef S{c, c + np, c + 2 * np, u, u + np, u + 2 * np};
    dim3 grid((unsigned)total_tiles);
    // automatic (-1): OFF for Sigma.  Measured (scripts/ab_assembly.py, and bench.py with CK_BENCH_OPTIONS=assemble_queue=...): the
    // work-queue form of K1 is 3 % faster in isolation on one box (1.31 -> 1.27 ms) and 7 % slower inside the bench's steps on
    // another (1.266 -> 1.354 ms) -- not a robust gain; K2's is (right-hand sides below).  An explicit number still selects it.
    if (queue_slots < 0) queue_slots = 0;
    if (fast && queue_slots > 0) {
        // a resident set of workgroups on a work queue (wl.count + 1); small launches are balanced in half or quarter strips
        const int csubs = total_tiles >= 16 * queue_slots ? 8 : (total_tiles >= 4 * queue_slots ? 4 : 2);
        const long chunks = (long)total_tiles * (8 / csubs);
        k_assemble<true, false><<<dim3((unsigned)std::min<long>(chunks, queue_slots)), dim3(256), 0, s>>>(
            blk, tabs, coefs, metric, 0, S, 0, S, nullptr, L, pm, wl, nullptr, 0, nullptr, nullptr, wl.count + 1, total_tiles, csubs);
    } else if (fast)
        k_assemble<true, false><<<grid, dim3(256), 0, s>>>(blk, tabs, coefs, metric, 0, S, 0, S, nullptr, L, pm, wl);
    else
        k_assemble<false, false><<<grid, dim3(256), 0, s>>>(blk, tabs, coefs, metric, 0, S, 0, S, nullptr, L, pm, wl);
}

// all right-hand-side panels in one launch
void ck_launch_assemble_aux(hipStream_t s, bool fast, const CkMatern* blk, const CkTable* tabs,
                            const double* const* coefs, int metric, int i_pred, double* pc, double* pu,
                            int64_t m, int64_t mpad, const double* c, const double* u, const double* z, CkLayout L,
                            int n_panels, double* aux, CkWorklist wl, const double* raw, int queue_slots) {
    if (mpad <= 0 || n_panels <= 0) return;
    const int64_t np = L.npad;
    CkSiteRef P{pc, pc + mpad, pc + 2 * mpad, pu, pu + mpad, pu + 2 * mpad};
    CkSiteRef S{c, c + np, c + 2 * np, u, u + np, u + 2 * np};
    CkPanelMap pm{nullptr, nullptr, nullptr, n_panels, aux, (long)(mpad / 64), nullptr};
    dim3 grid((unsigned)(n_panels * (mpad / 64)));
    // automatic (-1): 768 resident workgroups (three per CU) on a work queue from 8 strips per workgroup on: the 48 KB table is
    // loaded once per workgroup instead of once per strip and the launch is balanced to half a strip.  N = 40 000: K2 0.605 -> 0.56 ms
    // in isolation, 0.618 -> 0.561 ms inside the bench's steps (two boxes); at N = 10 000, where a chunk would have to be a quarter
    // strip, the per-chunk latencies cost what the balance buys: small launches keep one strip per workgroup.
    if (queue_slots < 0) queue_slots = (long)n_panels * (mpad / 64) >= 8 * 768 ? 768 : 0;
    if (fast && queue_slots > 0) {
        const long strips = (long)n_panels * (mpad / 64);
        const int csubs = strips >= 16 * queue_slots ? 8 : (strips >= 4 * queue_slots ? 4 : 2);
        const long chunks = strips * (8 / csubs);
        k_assemble<true, true><<<dim3((unsigned)std::min<long>(chunks, queue_slots)), dim3(256), 0, s>>>(
            blk, tabs, coefs, metric, i_pred, P, m, S, z, L, pm, wl, raw, (long)mpad, pc, pu, wl.count + 1, (int)strips, csubs);
    } else if (fast)
        k_assemble<true, true><<<grid, dim3(256), 0, s>>>(blk, tabs, coefs, metric, i_pred, P, m, S, z, L, pm, wl, raw, (long)mpad, pc, pu);
    else
        k_assemble<false, true><<<grid, dim3(256), 0, s>>>(blk, tabs, coefs, metric, i_pred, P, m, S, z, L, pm, wl);
}

// after all table-path panels of one assembly: evaluate what they deferred
void ck_launch_assemble_fix(hipStream_t s, bool aux_rows, const CkMatern* blk, int metric, int i_pred,
                            const double* pc, int64_t mpad, const double* c, CkLayout L, CkWorklist wl,
                            double* const* sigptr, double* aux) {
    const int64_t np = L.npad;
    CkSiteRef S{c, c + np, c + 2 * np, nullptr, nullptr, nullptr};
    if (aux_rows) {
        CkSiteRef P{pc, pc + mpad, pc + 2 * mpad, nullptr, nullptr, nullptr};
        k_assemble_fix<true><<<dim3(256), dim3(256), 0, s>>>(blk, metric, i_pred, P, S, L, wl, sigptr, aux, mpad);
    } else {
        k_assemble_fix<false><<<dim3(256), dim3(256), 0, s>>>(blk, metric, i_pred, S, S, L, wl, sigptr, aux, mpad);
    }
}

// dense a x b block (element-wise parity surface) ------------------------------------------
__global__ __launch_bounds__(256) void k_cov_dense(const CkMatern* __restrict__ m, int metric, int add_nugget,
                                                    int mode, const double* __restrict__ a0,
                                                    const double* __restrict__ a1, const double* __restrict__ a2,
                                                    long a, const double* __restrict__ b0,
                                                    const double* __restrict__ b1, const double* __restrict__ b2,
                                                    long b, double* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a * b) return;
    const long r = idx / b, c = idx - r * b;
    const double d = pair_dist(metric, a0[r], a1[r], a2[r], b0[c], b1[c], b2[c]);
    out[r * b + c] = mode == 1 ? d : ck_cov_entry(*m, d, add_nugget);
}

void ck_launch_cov_dense(hipStream_t s, const CkMatern* blk_ij, int metric, int add_nugget, int mode,
                         const double* a0, const double* a1, const double* a2, int64_t a, const double* b0,
                         const double* b1, const double* b2, int64_t b, double* out) {
    if (a <= 0 || b <= 0) return;
    k_cov_dense<<<dim3((unsigned)((a * b + 255) / 256)), dim3(256), 0, s>>>(blk_ij, metric, add_nugget, mode,
                                                                                     a0, a1, a2, a, b0, b1, b2, b, out);
}

__global__ void k_cov_lags(const CkMatern* __restrict__ m, int add_nugget, const double* __restrict__ lags, long n,
                           double* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = ck_cov_entry(*m, fabs(lags[i]), add_nugget);   // model.py:373 takes |h|
}

void ck_launch_cov_lags(hipStream_t s, const CkMatern* blk_ij, int add_nugget, const double* lags, int64_t n,
                        double* out) {
    if (n <= 0) return;
    k_cov_lags<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(blk_ij, add_nugget, lags, n, out);
}


// Model (cross-)variograms row by row (src/model.py:209-237): row r has process pair (pi[r], pj[r])
// and lag h[r].  kind 0: semivariance sigma_i^2 (1 - rho) + nugget_i (i == j), sill - C_ij (i != j,
// sill = mean of the two marginal sills, :219-221); kind 1: covariance incl. nugget / cross-covariance.
__global__ void k_model_variogram(const CkMatern* __restrict__ blk, double sill, int kind,
                                  const int* __restrict__ pi, const int* __restrict__ pj,
                                  const double* __restrict__ lags, long n, double* __restrict__ out) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int i = pi[r], j = pj[r];
    const CkMatern& m = blk[i + j];
    const double h = fabs(lags[r]);   // model.py:373 takes |h|
    double v;
    if (kind == 1) {
        v = ck_cov_entry(m, h, i == j);
    } else if (i == j) {
        const double rho = h == 0.0 ? 1.0 : ck_matern_rho_scaled(m, m.sqrt2nu * (h / m.len_scale));
        v = m.amp * (1.0 - rho) + m.nugget;
    } else {
        v = sill - ck_cov_entry(m, h, 0);
    }
    out[r] = v;
}

void ck_launch_model_variogram(hipStream_t s, const CkMatern* blk, double sill, int kind, const int* pi,
                               const int* pj, const double* lags, int64_t n, double* out) {
    if (n <= 0) return;
    k_model_variogram<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(blk, sill, kind, pi, pj, lags, n, out);
}

// =========================================================================================
// Tabulated fast path (ck_math.h "Tabulated correlation")
// =========================================================================================
// C = amp * rho at given squared chords q (table construction; the polynomial fit is done on the host)
__global__ void k_table_nodes(const CkMatern* __restrict__ m, int metric, const double* __restrict__ q, long n,
                              double* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = m->amp * ck_matern_rho_scaled(*m, ck_s_of_q(*m, metric, q[i]));
}

void ck_launch_table_nodes(hipStream_t s, const CkMatern* m, int metric, const double* q, int64_t n, double* out) {
    if (n <= 0) return;
    k_table_nodes<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(m, metric, q, n, out);
}

// Table error against the exact evaluator, 8 probe points per interval:
//   |table - amp rho| / (|amp| max(rho, 1e-6))
// i.e. the relative error of the entry while rho >= 1e-6 and the absolute error in units of
// 1e-6 amp below that (ck_math.h explains why the tail is judged absolutely).
__global__ void k_table_check(const CkMatern* __restrict__ m, int metric, CkTable tab,
                              const double* __restrict__ coef, unsigned long long* __restrict__ max_err_bits) {
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= tab.n_int * 8) return;
    const int interval = it >> 3, j = it & 7;
    union { double d; unsigned long long u; } a, b;
    a.u = (unsigned long long)(tab.base + interval) << CK_TAB_SHIFT;
    b.u = (unsigned long long)(tab.base + interval + 1) << CK_TAB_SHIFT;
    const double q = a.d + (0.03125 + 0.125 * j) * (b.d - a.d);
    int iv;
    const double y = ck_table_y(q, &iv, tab.base);
    double e = 1.0;
    if (iv == interval) {
        const double got = ck_table_poly(coef, iv, y);
        const double rho = ck_matern_rho_scaled(*m, ck_s_of_q(*m, metric, q));
        e = m->amp == 0.0 ? fabs(got) : fabs(got - m->amp * rho) / (fabs(m->amp) * fmax(rho, 1e-6));
        if (!(e < 1.0)) e = 1.0;
    }
    union { double d; unsigned long long u; } ev;
    ev.d = e;
    atomicMax(max_err_bits, ev.u);   // non-negative doubles order like their bit patterns
}

void ck_launch_table_check(hipStream_t s, const CkMatern* m, int metric, CkTable tab, const double* coef,
                           unsigned long long* max_err_bits) {
    const int n = tab.n_int * 8;
    if (n <= 0) return;
    k_table_check<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(m, metric, tab, coef, max_err_bits);
}
