// ck_local.hip -- local-neighbourhood cokriging: one workgroup per prediction point.
//
// Replaces the per-row Python of point_prediction.Predictor (src/point_prediction.py:127-249):
//   _local_dist_ix   radius search d <= max_dist per process (cross-validation: process i also
//                    d > 0, :141-143)
//   _local_values    gather of the precomputed Sigma blocks  -> here the k x k local covariance is
//                    assembled on the fly from the k neighbours (no global Sigma is ever stored)
//   _pred_calc       cho_factor / cho_solve, pred = w.z, std = sqrt(c0 - w.c), nanmax([std, 0])
//                    -> one Cholesky of the local system carrying c and z as two extra rows
//                    (forward substitution only: pred = v.y, var = c0 - v.v)
// Empty neighbourhood -> (NaN, NaN) (:229-233); local Sigma not positive definite -> (NaN, NaN)
// (:218-222).  Neighbours keep the reference's order: process 0 sites ascending, then process 1.
// Systems with k <= 64 neighbours live in LDS (k_local_solve); larger ones go through the tiled path (batched
// 64-column steps on the matrix cores, see below) or, if option "local_tile_min" says so, a global scratch slab
// per point with a blocked factorisation in one workgroup (k_local_solve_big).
#include "ck_internal.h"

typedef double d2_t __attribute__((ext_vector_type(2)));

#define LP_TPB 256
#define LP_KL 64             // (64 + 2) * 64 doubles = 34 KB of LDS: four workgroups per CU (with 124 and one
                             // workgroup per CU the kernel was 2x slower than the tiled path from k ~ 40 on)

__device__ __forceinline__ double lp_dist(int metric, double a0, double a1, double a2, double b0, double b1,
                                          double b2) {
    return metric == CK_METRIC_HAVERSINE ? ck_haversine_km(a0, a1, a2, b0, b1, b2) : ck_euclid(a0, a1, b0, b1);
}

__device__ __forceinline__ bool lp_is_neighbour(int metric, int cv, int i_pred, double max_dist, const CkLayout& L,
                                                long g, double p0, double p1, double p2, const double* s0,
                                                const double* s1, const double* s2) {
    if (!(g < L.n0 || (g >= L.n0p && g < L.nend))) return false;
    const double d = lp_dist(metric, p0, p1, p2, s0[g], s1[g], s2[g]);
    const int proc = g >= L.n0p;
    if (cv && proc == i_pred) return d > 0.0 && d <= max_dist;
    return d <= max_dist;
}

// Covariance of a pair: through the block's table (ck_math.h "Tabulated covariance"; coefficients read from
// global memory / L2 here -- three 48 KB tables do not fit next to the local system in LDS) when the pair's
// squared chord is inside the table, else the exact evaluator (closer than the table's lower end incl.
// h == 0 and the nugget, beyond its upper end, or tables disabled).
struct LpTab {
    const CkTable* tabs;             // [3]
    const double* const* coefs;      // [3]
    int use;
};

// the exact formulas out of line: rare beside the table, and inlined they cost the callers their registers
__device__ __noinline__ double lp_exact_xyz(const CkMatern* m, int metric, int nug, double a0, double a1, double a2,
                                            double b0, double b1, double b2) {
    return ck_cov_entry(*m, lp_dist(metric, a0, a1, a2, b0, b1, b2), nug);
}

__device__ __forceinline__ double lp_cov(const CkMatern* blk, const LpTab& T, int bidx, int nug, int metric, double a0,
                                         double a1, double a2, double au0, double au1, double au2, double b0, double b1,
                                         double b2, double bu0, double bu1, double bu2) {
    if (T.use) {
        const double dx = au0 - bu0, dy = au1 - bu1, dz = au2 - bu2;
        const double q = dx * dx + dy * dy + dz * dz;
        int iv;
        const double y = ck_table_y(q, &iv, T.tabs[bidx].base);
        if ((unsigned)iv < (unsigned)T.tabs[bidx].n_int) return ck_table_poly(T.coefs[bidx], iv, y);
    }
    return lp_exact_xyz(&blk[bidx], metric, nug, a0, a1, a2, b0, b1, b2);
}

// Radius search with chunk culling: the sites are laid out along a Hilbert curve (ck_api.hip: site_order), so
// 256 consecutive sites are a compact patch.  Per chunk: the mean c of its chord vectors and rad = max |u - c|.
// A site can only be within max_dist of a point with chord vector q if its chord to q is <= cmax (haversine:
// 2 sin(max_dist / 2R), monotone; Euclid: max_dist), and |u - q| >= |c - q| - rad, so a chunk with
// |c - q| - rad > cmax holds no neighbour and is skipped by the whole workgroup.  cmax carries a relative
// margin of 1e-9 (host), far above the rounding of either formula; which sites pass is still decided by the
// reference's distance formula, so the neighbour lists do not change.
struct LpSearch {
    const double* cb;   // 4 x nchunk: centre x, y, z, rad (rad < 0: no valid site in the chunk)
    long nchunk;
    double cmax;
};

__device__ __forceinline__ bool lp_chunk_far(const LpSearch& R, long chunk, double q0, double q1, double q2) {
    if (chunk >= R.nchunk) return true;
    const double rad = R.cb[3 * R.nchunk + chunk];
    if (rad < 0.0) return true;
    const double dx = R.cb[chunk] - q0, dy = R.cb[R.nchunk + chunk] - q1, dz = R.cb[2 * R.nchunk + chunk] - q2;
    return sqrt(dx * dx + dy * dy + dz * dz) - rad > R.cmax;
}

__global__ __launch_bounds__(LP_TPB) void k_local_chunk_bounds(const double* __restrict__ su, CkLayout L, long nchunk,
                                                                double* __restrict__ cb) {
    __shared__ double red[4][LP_TPB];
    const int tid = threadIdx.x;
    const long g = (long)blockIdx.x * LP_TPB + tid;
    const bool valid = g < L.n0 || (g >= L.n0p && g < L.nend);
    const double x = valid ? su[g] : 0.0, y = valid ? su[L.npad + g] : 0.0, zc = valid ? su[2 * L.npad + g] : 0.0;
    red[0][tid] = x;
    red[1][tid] = y;
    red[2][tid] = zc;
    red[3][tid] = valid ? 1.0 : 0.0;
    __syncthreads();
    for (int s = LP_TPB / 2; s > 0; s >>= 1) {
        if (tid < s)
            for (int c = 0; c < 4; ++c) red[c][tid] += red[c][tid + s];
        __syncthreads();
    }
    const double n = red[3][0];
    const double cx = n > 0.0 ? red[0][0] / n : 0.0, cy = n > 0.0 ? red[1][0] / n : 0.0, cz = n > 0.0 ? red[2][0] / n : 0.0;
    __syncthreads();
    const double dx = x - cx, dy = y - cy, dz = zc - cz;
    red[0][tid] = valid ? sqrt(dx * dx + dy * dy + dz * dz) : 0.0;
    __syncthreads();
    for (int s = LP_TPB / 2; s > 0; s >>= 1) {
        if (tid < s) red[0][tid] = fmax(red[0][tid], red[0][tid + s]);
        __syncthreads();
    }
    if (tid == 0) {
        cb[blockIdx.x] = cx;
        cb[nchunk + blockIdx.x] = cy;
        cb[2 * nchunk + blockIdx.x] = cz;
        cb[3 * nchunk + blockIdx.x] = n > 0.0 ? red[0][0] * (1.0 + 1e-12) : -1.0;
    }
}

void ck_launch_local_chunk_bounds(hipStream_t s, const double* su, CkLayout L, double* cb) {
    const long nchunk = (L.nend + LP_TPB - 1) / LP_TPB;
    if (nchunk <= 0) return;
    k_local_chunk_bounds<<<dim3((unsigned)nchunk), dim3(LP_TPB), 0, s>>>(su, L, nchunk, cb);
}

// neighbour count per prediction point
__global__ __launch_bounds__(LP_TPB) void k_local_count(int metric, int i_pred, int cv, double max_dist,
                                                         const double* __restrict__ pc, long mpad,
                                                         const double* __restrict__ sc, CkLayout L,
                                                         int* __restrict__ counts, LpSearch R,
                                                         const double* __restrict__ pu) {
    __shared__ int red[LP_TPB];
    const long p = blockIdx.x;
    const double p0 = pc[p], p1 = pc[mpad + p], p2 = pc[2 * mpad + p];
    const double q0 = pu[p], q1 = pu[mpad + p], q2 = pu[2 * mpad + p];
    const double *s0 = sc, *s1 = sc + L.npad, *s2 = sc + 2 * L.npad;
    int c = 0;
    for (long g0 = 0; g0 < L.nend; g0 += LP_TPB) {
        if (lp_chunk_far(R, g0 / LP_TPB, q0, q1, q2)) continue;
        const long g = g0 + threadIdx.x;
        c += (g < L.nend && lp_is_neighbour(metric, cv, i_pred, max_dist, L, g, p0, p1, p2, s0, s1, s2)) ? 1 : 0;
    }
    red[threadIdx.x] = c;
    __syncthreads();
    for (int s = LP_TPB / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[p] = red[0];
}

__global__ __launch_bounds__(LP_TPB) void k_local_solve(const CkMatern* __restrict__ blk, int metric, int i_pred,
                                                         int cv, double max_dist, const double* __restrict__ pc,
                                                         long mpad, const double* __restrict__ sc,
                                                         const double* __restrict__ z, CkLayout L,
                                                         const int* __restrict__ counts,
                                                         const long long* __restrict__ slab_off,
                                                         double* __restrict__ slab, double c0var,
                                                         double* __restrict__ pred, double* __restrict__ err,
                                                         long p_base, LpTab T, const double* __restrict__ su,
                                                         const double* __restrict__ pu, LpSearch R, int k_hi) {
    __shared__ double lS[(LP_KL + 2) * LP_KL];
    __shared__ int lidx[LP_KL];
    __shared__ int wsum[LP_TPB / 64];
    __shared__ int fail;
    __shared__ double red[2][LP_TPB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long p = p_base + blockIdx.x;
    const int k = counts[p];
    if (k == 0) {   // src/point_prediction.py:229-233
        if (tid == 0) {
            pred[p] = NAN;
            err[p] = NAN;
        }
        return;
    }
    if (k > LP_KL || k > k_hi) return;   // larger systems: k_local_solve_big / the tiled path
    const double p0 = pc[p], p1 = pc[mpad + p], p2 = pc[2 * mpad + p];
    const double *s0 = sc, *s1 = sc + L.npad, *s2 = sc + 2 * L.npad;
    const double *u0 = su, *u1 = su + L.npad, *u2 = su + 2 * L.npad;        // chord vectors (table path)
    const double q0 = pu[p], q1 = pu[mpad + p], q2 = pu[2 * mpad + p];
    // storage: (k + 2) x k matrix (rows k, k + 1 carry c and z) and the neighbour index list
    double* S = lS;
    int* idx = lidx;
    const long ld = LP_KL;
    // ---- 1. neighbour list, in site order (block-wide ordered compaction) ----
    if (tid == 0) fail = 0;
    int base = 0;
    for (long g0 = 0; g0 < L.nend; g0 += LP_TPB) {
        if (lp_chunk_far(R, g0 / LP_TPB, q0, q1, q2)) continue;   // uniform
        const long g = g0 + tid;
        const bool f = g < L.nend && lp_is_neighbour(metric, cv, i_pred, max_dist, L, g, p0, p1, p2, s0, s1, s2);
        const unsigned long long bal = __ballot(f);
        const int below = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) wsum[wv] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w2 = 0; w2 < wv; ++w2) off += wsum[w2];
        if (f) idx[off + below] = (int)g;
        int tot = 0;
        for (int w2 = 0; w2 < LP_TPB / 64; ++w2) tot += wsum[w2];
        base += tot;
        __syncthreads();
    }
    // ---- 2. local covariance (lower triangle), c row, z row ----
    const long npair = (long)k * (k + 1) / 2;
    for (long e = tid; e < npair; e += LP_TPB) {
        // e -> (a, b), a >= b:  a = floor((sqrt(8e + 1) - 1) / 2)
        long a = (long)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
        while (a * (a + 1) / 2 > e) --a;
        while ((a + 1) * (a + 2) / 2 <= e) ++a;
        const long b = e - a * (a + 1) / 2;
        const long ga = idx[a], gb = idx[b];
        const int pa = ga >= L.n0p, pb = gb >= L.n0p;
        S[a * ld + b] = lp_cov(blk, T, pa + pb, pa == pb, metric, s0[ga], s1[ga], s2[ga], u0[ga], u1[ga], u2[ga], s0[gb],
                               s1[gb], s2[gb], u0[gb], u1[gb], u2[gb]);
    }
    for (int a = tid; a < k; a += LP_TPB) {
        const long ga = idx[a];
        const int pa = ga >= L.n0p;
        S[(long)k * ld + a] = lp_cov(blk, T, i_pred + pa, pa == i_pred, metric, p0, p1, p2, q0, q1, q2, s0[ga], s1[ga],
                                     s2[ga], u0[ga], u1[ga], u2[ga]);   // point_prediction.py:115-125
        S[(long)(k + 1) * ld + a] = z[ga];
    }
    __syncthreads();
    // ---- 3. Cholesky of the k x k block, the two extra rows ride along (forward substitution) ----
    for (int j = 0; j < k; ++j) {
        const double piv = S[(long)j * ld + j];
        if (!(piv > 0.0)) {
            if (tid == 0) fail = 1;
            break;   // uniform: every thread reads the same pivot
        }
        const double rd = 1.0 / sqrt(piv);
        __syncthreads();
        for (int a = j + 1 + tid; a < k + 2; a += LP_TPB) S[(long)a * ld + j] *= rd;
        if (tid == 0) S[(long)j * ld + j] = sqrt(piv);
        __syncthreads();
        // S[a][b] -= S[a][j] S[b][j] for j < b <= min(a, k - 1), a in (j, k + 2)
        const int nb = k - 1 - j;            // columns b = j + 1 .. k - 1
        const int na = k + 1 - j;            // rows    a = j + 1 .. k + 1
        const long tot = (long)na * nb;
        for (long e = tid; e < tot; e += LP_TPB) {
            const int a = j + 1 + (int)(e / nb), b = j + 1 + (int)(e % nb);
            if (b <= a) S[(long)a * ld + b] -= S[(long)a * ld + j] * S[(long)b * ld + j];
        }
        __syncthreads();
    }
    __syncthreads();
    if (fail) {   // src/point_prediction.py:218-222
        if (tid == 0) {
            pred[p] = NAN;
            err[p] = NAN;
        }
        return;
    }
    // ---- 4. pred = v . y, var = c0 - v . v ----
    double s1v = 0.0, s2v = 0.0;
    for (int a = tid; a < k; a += LP_TPB) {
        const double v = S[(long)k * ld + a], y = S[(long)(k + 1) * ld + a];
        s1v += v * y;
        s2v += v * v;
    }
    red[0][tid] = s1v;
    red[1][tid] = s2v;
    __syncthreads();
    for (int s = LP_TPB / 2; s > 0; s >>= 1) {
        if (tid < s) {
            red[0][tid] += red[0][tid + s];
            red[1][tid] += red[1][tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        pred[p] = red[0][0];
        const double sd = sqrt(c0var - red[1][0]);
        err[p] = (sd == sd) ? fmax(sd, 0.0) : 0.0;   // np.nanmax([std, 0.0]), point_prediction.py:217
    }
}

// ---------------------------------------------------------------------------------------
// Neighbourhoods beyond the LDS limit: blocked Cholesky of the local system in a global slab
// ---------------------------------------------------------------------------------------
// Same steps as k_local_solve (neighbour list in site order, local covariance with the c and z rows
// riding along, forward substitution only).  The factorisation is blocked: LB_IB columns are
// factored in place (unblocked, rank-1 updates confined to the column panel), then the trailing
// matrix is updated ONCE with all of them -- 64 x 64 tiles, the two LB_IB-wide operand strips of
// a tile staged (transposed) in LDS, a 4 x 4 register tile per thread.  The unblocked form makes
// one pass over the trailing matrix per COLUMN: k / LB_IB times the memory traffic (it measured
// 0.4 TFLOP/s at k = 1 359).
#define LB_IB 32
#define LB_PR 128   // panel rows per pass; LB_IB * (LB_PR + 1) <= 2 * LB_IB * 68 doubles of LDS
__global__ __launch_bounds__(LP_TPB, 3) void k_local_solve_big(const CkMatern* __restrict__ blk, int metric, int i_pred,
                                                             int cv, double max_dist,
                                                             const double* __restrict__ pc, long mpad,
                                                             const double* __restrict__ sc,
                                                             const double* __restrict__ z, CkLayout L,
                                                             const int* __restrict__ counts,
                                                             const long long* __restrict__ slab_off,
                                                             double* __restrict__ slab, double c0var,
                                                             double* __restrict__ pred, double* __restrict__ err,
                                                             long p_base, LpTab T, const double* __restrict__ su,
                                                             const double* __restrict__ pu, int k_hi, LpSearch R) {
    __shared__ __attribute__((aligned(16))) double lbuf[2 * LB_IB * (64 + 4)];
    double (*At)[64 + 4] = reinterpret_cast<double (*)[64 + 4]>(lbuf);                 // At[c][r] = strip of the tile's rows
    double (*Bt)[64 + 4] = reinterpret_cast<double (*)[64 + 4]>(lbuf + LB_IB * (64 + 4));   // Bt[c][r] = ... columns
    double (*Pt)[LB_PR + 1] = reinterpret_cast<double (*)[LB_PR + 1]>(lbuf);           // panel rows (3a), same storage
    __shared__ double Ld[LB_IB][LB_IB + 1];   // diagonal block of the column panel
    __shared__ double rdiag[LB_IB], Ldiag[LB_IB];
    __shared__ int wsum[LP_TPB / 64];
    __shared__ int fail;
    __shared__ double red[2][LP_TPB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long p = p_base + blockIdx.x;
    const int k = counts[p];
    if (k <= LP_KL || k > k_hi) return;   // k_local_solve (also the empty neighbourhoods) / the tiled path
    const double p0 = pc[p], p1 = pc[mpad + p], p2 = pc[2 * mpad + p];
    const double *s0 = sc, *s1 = sc + L.npad, *s2 = sc + 2 * L.npad;
    const double *u0 = su, *u1 = su + L.npad, *u2 = su + 2 * L.npad;        // chord vectors (table path)
    const double q0 = pu[p], q1 = pu[mpad + p], q2 = pu[2 * mpad + p];
    double* S = slab + slab_off[p];   // (k + 2) x k, rows k and k + 1 carry c and z
    int* idx = reinterpret_cast<int*>(S + (long)(k + 2) * k);
    const long ld = k;
    // ---- 1. neighbour list, in site order (block-wide ordered compaction) ----
    if (tid == 0) fail = 0;
    int base = 0;
    for (long g0 = 0; g0 < L.nend; g0 += LP_TPB) {
        if (lp_chunk_far(R, g0 / LP_TPB, q0, q1, q2)) continue;   // uniform
        const long g = g0 + tid;
        const bool f = g < L.nend && lp_is_neighbour(metric, cv, i_pred, max_dist, L, g, p0, p1, p2, s0, s1, s2);
        const unsigned long long bal = __ballot(f);
        const int below = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) wsum[wv] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w2 = 0; w2 < wv; ++w2) off += wsum[w2];
        if (f) idx[off + below] = (int)g;
        int tot = 0;
        for (int w2 = 0; w2 < LP_TPB / 64; ++w2) tot += wsum[w2];
        base += tot;
        __syncthreads();
    }
    // ---- 2. local covariance (lower triangle), c row, z row ----
    for (int a = 0; a < k; ++a) {   // row by row: coalesced writes, no sqrt to invert the triangular index
        const long ga = idx[a];
        const int pa = ga >= L.n0p;
        const double a0 = s0[ga], a1 = s1[ga], a2 = s2[ga], au0 = u0[ga], au1 = u1[ga], au2 = u2[ga];
        for (int b = tid; b <= a; b += LP_TPB) {
            const long gb = idx[b];
            const int pb = gb >= L.n0p;
            S[(long)a * ld + b] = lp_cov(blk, T, pa + pb, pa == pb, metric, a0, a1, a2, au0, au1, au2, s0[gb], s1[gb], s2[gb],
                                         u0[gb], u1[gb], u2[gb]);
        }
    }
    for (int a = tid; a < k; a += LP_TPB) {
        const long ga = idx[a];
        const int pa = ga >= L.n0p;
        S[(long)k * ld + a] = lp_cov(blk, T, i_pred + pa, pa == i_pred, metric, p0, p1, p2, q0, q1, q2, s0[ga], s1[ga],
                                     s2[ga], u0[ga], u1[ga], u2[ga]);   // point_prediction.py:115-125
        S[(long)(k + 1) * ld + a] = z[ga];
    }
    __syncthreads();
    // ---- 3. blocked Cholesky; rows k, k + 1 ride along (forward substitution) ----
    const int ty = tid >> 4, tx = tid & 15;
    bool bad = false;
    for (int jb = 0; jb < k && !bad; jb += LB_IB) {
        const int nbc = (k - jb < LB_IB) ? (k - jb) : LB_IB;
        const int jend = jb + nbc;
        // 3a. the column panel jb .. jend - 1: its diagonal block is factored in LDS (padded to LB_IB with the
        //     identity), then every row below is solved against it -- one thread per row, the row's LB_IB
        //     entries in registers, LB_PR rows per pass staged through LDS so that global memory sees whole
        //     256-byte row segments.  (Column-at-a-time on the slab walked it with a stride of k doubles:
        //     one cache line per entry, k times over -- half of this kernel's time at k = 1 359.)
        for (int e = tid; e < LB_IB * LB_IB; e += LP_TPB) {
            const int r = e / LB_IB, c = e % LB_IB;
            Ld[r][c] = (r < nbc && c <= r) ? S[(long)(jb + r) * ld + jb + c] : (r == c ? 1.0 : 0.0);
        }
        __syncthreads();
        for (int j = 0; j < LB_IB; ++j) {
            const double piv = Ld[j][j];
            if (!(piv > 0.0)) {   // uniform: every thread reads the same pivot
                if (tid == 0) fail = 1;
                bad = true;
                break;
            }
            const double rd = 1.0 / sqrt(piv);
            if (tid > j && tid < LB_IB) Ld[tid][j] *= rd;
            if (tid == 0) {
                rdiag[j] = rd;
                Ldiag[j] = sqrt(piv);
            }
            __syncthreads();
            for (int e = tid; e < LB_IB * LB_IB; e += LP_TPB) {
                const int a = e / LB_IB, b = e % LB_IB;
                if (b > j && a >= b) Ld[a][b] -= Ld[a][j] * Ld[b][j];
            }
            __syncthreads();
        }
        if (bad) break;
        for (int e = tid; e < LB_IB * LB_IB; e += LP_TPB) {
            const int r = e / LB_IB, c = e % LB_IB;
            if (r < nbc && c <= r) S[(long)(jb + r) * ld + jb + c] = (r == c) ? Ldiag[r] : Ld[r][c];
        }
        for (int r0 = jend; r0 < k + 2; r0 += LB_PR) {
            for (int e = tid; e < LB_PR * LB_IB; e += LP_TPB) {
                const int r = e / LB_IB, c = e % LB_IB, a = r0 + r;
                Pt[c][r] = (a < k + 2 && c < nbc) ? S[(long)a * ld + jb + c] : 0.0;
            }
            __syncthreads();
            if (tid < LB_PR) {
                double x[LB_IB];
#pragma unroll
                for (int c = 0; c < LB_IB; ++c) x[c] = Pt[c][tid];
#pragma unroll
                for (int j = 0; j < LB_IB; ++j) {
                    x[j] *= rdiag[j];
#pragma unroll
                    for (int c = j + 1; c < LB_IB; ++c) x[c] -= x[j] * Ld[c][j];
                }
#pragma unroll
                for (int c = 0; c < LB_IB; ++c) Pt[c][tid] = x[c];
            }
            __syncthreads();
            for (int e = tid; e < LB_PR * LB_IB; e += LP_TPB) {
                const int r = e / LB_IB, c = e % LB_IB, a = r0 + r;
                if (a < k + 2 && c < nbc) S[(long)a * ld + jb + c] = Pt[c][r];
            }
            __syncthreads();
        }
        // 3b. trailing update  S[a][b] -= sum_c S[a][jb + c] S[b][jb + c],  a in [jend, k + 2), b in [jend, min(a, k - 1)]
        const int nrow = k + 2 - jend, ncol = k - jend;
        const int tr = (nrow + 63) / 64, tc = (ncol + 63) / 64;
        for (int ta = 0; ta < tr; ++ta) {
            for (int tb = 0; tb < tc && tb <= ta; ++tb) {
                for (int e = tid; e < 64 * LB_IB; e += LP_TPB) {
                    const int r = e / LB_IB, c = e % LB_IB;            // consecutive threads: consecutive columns of one row
                    const int a = jend + ta * 64 + r, b = jend + tb * 64 + r;
                    At[c][r] = (a < k + 2 && c < nbc) ? S[(long)a * ld + jb + c] : 0.0;
                    Bt[c][r] = (b < k && c < nbc) ? S[(long)b * ld + jb + c] : 0.0;
                }
                __syncthreads();
                double acc[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j2 = 0; j2 < 4; ++j2) acc[i][j2] = 0.0;
#pragma unroll 8
                for (int c = 0; c < LB_IB; ++c) {
                    const d2_t a01 = *reinterpret_cast<const d2_t*>(&At[c][4 * ty]), a23 = *reinterpret_cast<const d2_t*>(&At[c][4 * ty + 2]);
                    const d2_t b01 = *reinterpret_cast<const d2_t*>(&Bt[c][4 * tx]), b23 = *reinterpret_cast<const d2_t*>(&Bt[c][4 * tx + 2]);
                    const double av[4] = {a01[0], a01[1], a23[0], a23[1]}, bv[4] = {b01[0], b01[1], b23[0], b23[1]};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j2 = 0; j2 < 4; ++j2) acc[i][j2] = fma(av[i], bv[j2], acc[i][j2]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int a = jend + ta * 64 + 4 * ty + i;
#pragma unroll
                    for (int j2 = 0; j2 < 4; ++j2) {
                        const int b = jend + tb * 64 + 4 * tx + j2;
                        if (a < k + 2 && b < k && b <= a) S[(long)a * ld + b] -= acc[i][j2];
                    }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
    if (fail) {   // src/point_prediction.py:218-222
        if (tid == 0) {
            pred[p] = NAN;
            err[p] = NAN;
        }
        return;
    }
    // ---- 4. pred = v . y, var = c0 - v . v ----
    double s1v = 0.0, s2v = 0.0;
    for (int a = tid; a < k; a += LP_TPB) {
        const double v = S[(long)k * ld + a], y = S[(long)(k + 1) * ld + a];
        s1v += v * y;
        s2v += v * v;
    }
    red[0][tid] = s1v;
    red[1][tid] = s2v;
    __syncthreads();
    for (int s = LP_TPB / 2; s > 0; s >>= 1) {
        if (tid < s) {
            red[0][tid] += red[0][tid + s];
            red[1][tid] += red[1][tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        pred[p] = red[0][0];
        const double sd = sqrt(c0var - red[1][0]);
        err[p] = (sd == sd) ? fmax(sd, 0.0) : 0.0;   // np.nanmax([std, 0.0]), point_prediction.py:217
    }
}

void ck_launch_local_count(hipStream_t s, int metric, int i_pred, int cv, double max_dist, const double* pc,
                           int64_t m, int64_t mpad, const double* sc, CkLayout L, int* counts, const double* cb,
                           double cmax, const double* pu) {
    if (m <= 0) return;
    const LpSearch R{cb, (long)((L.nend + LP_TPB - 1) / LP_TPB), cmax};
    k_local_count<<<dim3((unsigned)m), dim3(LP_TPB), 0, s>>>(metric, i_pred, cv, max_dist, pc, mpad, sc, L, counts, R, pu);
}

// points [p_base, p_base + m): slab_off is relative to `slab` within this batch (ck_predict_local)
void ck_launch_local_solve(hipStream_t s, const CkMatern* blk, int metric, int i_pred, int cv, double max_dist,
                           const double* pc, int64_t p_base, int64_t m, int64_t mpad, const double* sc, const double* z,
                           CkLayout L, const int* counts, const long long* slab_off, double* slab, double c0var,
                           double* pred, double* err, const CkTable* tabs, const double* const* coefs, int use_tab,
                           const double* su, const double* pu, int k_hi, const double* cb, double cmax) {
    if (m <= 0) return;
    const LpTab T{tabs, coefs, use_tab};
    const LpSearch R{cb, (long)((L.nend + LP_TPB - 1) / LP_TPB), cmax};
    k_local_solve<<<dim3((unsigned)m), dim3(LP_TPB), 0, s>>>(blk, metric, i_pred, cv, max_dist, pc, mpad, sc, z, L,
                                                             counts, slab_off, slab, c0var, pred, err, p_base, T, su, pu, R, k_hi);
    if (slab && k_hi > LP_KL)   // some neighbourhood is larger than the LDS limit
        k_local_solve_big<<<dim3((unsigned)m), dim3(LP_TPB), 0, s>>>(blk, metric, i_pred, cv, max_dist, pc, mpad, sc, z,
                                                                     L, counts, slab_off, slab, c0var, pred, err,
                                                                     p_base, T, su, pu, k_hi, R);
}

// ---------------------------------------------------------------------------------------
// tiled path (ck_internal.h: CkLocalSys; the factorisation steps are in ck_la.hip)
// ---------------------------------------------------------------------------------------
// the exact formulas as an out-of-line call: rare in the table kernel below, and out of line it does not inflate
// its registers
__device__ __noinline__ double lp_exact_pair(const CkMatern* m, int metric, int nug, const double* s0, const double* s1,
                                             const double* s2, long ga, long gb) {
    return ck_cov_entry(*m, lp_dist(metric, s0[ga], s1[ga], s2[ga], s0[gb], s1[gb], s2[gb]), nug);
}

// Neighbour lists and padded local systems of a batch.
// k_local_search_t (one workgroup per system): the neighbour list, the number k0 of process-0 neighbours, the
// identity padding, the zeros the 64 x 64 factorisation expects above the diagonal, the c and z rows.
// k_local_assemble_t (one launch per Matern block): the neighbours come out sorted by process (process 0 first),
// so the lower triangle splits into three regions with ONE block each -- (0,0), (1,0) and (1,1).  A launch keeps
// its block's table in LDS, like the joint assembly (ck_cov.hip), and its workgroups walk over the systems
// (loading 48 KB of table per system cost more than the entries of a 100-site neighbourhood): chord vectors of
// LT_BC columns and LT_AC rows staged in LDS, a thread takes one column of four rows at a time (four
// independent Horner chains), writes run along rows.  Entries outside the table (coincident sites, pairs beyond
// its range) and everything when the tables are off go through the exact evaluator in a second, rolled loop.
#define LT_BC 512
#define LT_WL 1024   // deferred exact pairs per system and block (LDS list; beyond that they are evaluated in place)
#define LT_AC 256
#define LT_TPB 512   // 8 waves: with two workgroups per CU (LDS) four waves per SIMD hide the table-read latency
__global__ __launch_bounds__(LP_TPB) void k_local_search_t(const CkMatern* __restrict__ blk, int metric, int i_pred, int cv,
                                                            double max_dist, const double* __restrict__ pc, long mpad,
                                                            const double* __restrict__ sc, const double* __restrict__ z,
                                                            CkLayout L, const CkLocalSys* __restrict__ sys,
                                                            double* __restrict__ slab, LpTab T,
                                                            const double* __restrict__ su,
                                                            const double* __restrict__ pu, LpSearch R,
                                                            int* __restrict__ k0out) {
    __shared__ int wsum[LP_TPB / 64], wsum0[LP_TPB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const CkLocalSys q = sys[blockIdx.x];
    const long p = q.p;
    const int k = q.k, kq = q.kq;
    const long ld = q.ld;
    const double p0 = pc[p], p1 = pc[mpad + p], p2 = pc[2 * mpad + p];
    const double *s0 = sc, *s1 = sc + L.npad, *s2 = sc + 2 * L.npad;
    const double *u0 = su, *u1 = su + L.npad, *u2 = su + 2 * L.npad;
    const double q0 = pu[p], q1 = pu[mpad + p], q2 = pu[2 * mpad + p];
    double* S = slab + q.off;
    int* idx = reinterpret_cast<int*>(S + (long)CK_LT_ROWS(kq) * ld + CK_LT_NINV * 64 * 64);
    int base = 0, k0 = 0;   // k0: neighbours of process 0
    for (long g0 = 0; g0 < L.nend; g0 += LP_TPB) {   // ordered compaction, as in k_local_solve
        if (lp_chunk_far(R, g0 / LP_TPB, q0, q1, q2)) continue;   // uniform
        const long g = g0 + tid;
        const bool f = g < L.nend && lp_is_neighbour(metric, cv, i_pred, max_dist, L, g, p0, p1, p2, s0, s1, s2);
        const unsigned long long bal = __ballot(f), bal0 = __ballot(f && g < L.n0p);
        const int below = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) {
            wsum[wv] = __popcll(bal);
            wsum0[wv] = __popcll(bal0);
        }
        __syncthreads();
        int off = base;
        for (int w2 = 0; w2 < wv; ++w2) off += wsum[w2];
        if (f) idx[off + below] = (int)g;
        for (int w2 = 0; w2 < LP_TPB / 64; ++w2) {
            base += wsum[w2];
            k0 += wsum0[w2];
        }
        __syncthreads();
    }
    if (tid == 0) k0out[blockIdx.x] = k0;
    // zeros up to the end of each row's 4-column diagonal block (the 64 x 64 factorisation loads whole 4 x 4
    // register blocks); rows [k, kq - 2): identity padding
    for (int a = tid; a < k; a += LP_TPB)
        for (int b = a + 1; b <= (a | 3); ++b) S[(long)a * ld + b] = 0.0;
    for (int a = k; a < kq - 2; ++a)
        for (int b = tid; b <= (a | 3); b += LP_TPB) S[(long)a * ld + b] = (b == a) ? 1.0 : 0.0;
    // rows kq - 2 (c) and kq - 1 (z): two more rows of the matrix, their own 2 x 2 corner diag(BIG, BIG)
    for (int a = tid; a < kq; a += LP_TPB) {
        double cv0 = 0.0, zv = 0.0;
        if (a < k) {
            const long ga = idx[a];
            const int pa = ga >= L.n0p;
            cv0 = lp_cov(blk, T, i_pred + pa, pa == i_pred, metric, p0, p1, p2, q0, q1, q2, s0[ga], s1[ga], s2[ga], u0[ga],
                         u1[ga], u2[ga]);   // point_prediction.py:115-125
            zv = z[ga];
        }
        S[(long)(kq - 2) * ld + a] = a == kq - 2 ? CK_LT_BIG : cv0;
        S[(long)(kq - 1) * ld + a] = a == kq - 1 ? CK_LT_BIG : zv;
    }
}

__global__ __launch_bounds__(LT_TPB, 4) void k_local_assemble_t(const CkMatern* __restrict__ blk, int metric, int reg,
                                                                 const double* __restrict__ sc, CkLayout L,
                                                                 const CkLocalSys* __restrict__ sys, int n_sys,
                                                                 double* __restrict__ slab, LpTab T,
                                                                 const double* __restrict__ su,
                                                                 const int* __restrict__ k0in) {
    __shared__ double tab[(CK_TAB_DEG + 1) * CK_TAB_STRIDE];
    __shared__ double bu[3][LT_BC];
    __shared__ double au[3][LT_AC];
    __shared__ int2 wl[LT_WL];   // (row, column) of entries the table does not cover: evaluated one pair per LANE after
    __shared__ int wl_n;         // the system's region is done, instead of one pair per WAVE where they are found
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double *s0 = sc, *s1 = sc + L.npad, *s2 = sc + 2 * L.npad;
    const double *u0 = su, *u1 = su + L.npad, *u2 = su + 2 * L.npad;
    const int nug = reg != 1;
    const double cdiag = blk[reg].amp + blk[reg].nugget;   // an entry's own site: h = 0 (ck_cov_entry)
    int tbase = 0, tn = 0;                                  // tn = 0: every entry takes the exact evaluator
    if (T.use) {
        tbase = T.tabs[reg].base;
        tn = T.tabs[reg].n_int;
        const double* cf = T.coefs[reg];
        for (int e = tid; e < (CK_TAB_DEG + 1) * tn; e += LT_TPB) {
            const int kk = e / tn, iv = e - kk * tn;
            tab[kk * CK_TAB_STRIDE + iv] = cf[kk * CK_TAB_STRIDE + iv];
        }
    }
    for (int sidx = blockIdx.x; sidx < n_sys; sidx += gridDim.x) {
        const CkLocalSys q = sys[sidx];
        const int k = q.k, k0 = k0in[sidx];
        const long ld = q.ld;
        double* S = slab + q.off;
        const int* idx = reinterpret_cast<const int*>(S + (long)CK_LT_ROWS(q.kq) * ld + CK_LT_NINV * 64 * 64);
        const int alo = reg == 0 ? 0 : k0, ahi = reg == 0 ? k0 : k;
        const int blo = reg == 2 ? k0 : 0, bhi = reg == 2 ? k : k0;
        if (alo >= ahi || blo >= bhi) continue;   // uniform
        if (tid == 0) wl_n = 0;   // (ordered against the appends by the barrier at the top of the chunk loop)
        for (int bc = blo; bc < bhi; bc += LT_BC) {
            const int nbc = min(LT_BC, bhi - bc);
            __syncthreads();
            for (int e = tid; e < nbc; e += LT_TPB) {
                const long g = idx[bc + e];
                bu[0][e] = u0[g];
                bu[1][e] = u1[g];
                bu[2][e] = u2[g];
            }
            for (int ac = max(alo, bc); ac < ahi; ac += LT_AC) {
                const int nac = min(LT_AC, ahi - ac);
                __syncthreads();
                if (tid < nac) {
                    const long g = idx[ac + tid];
                    au[0][tid] = u0[g];
                    au[1][tid] = u1[g];
                    au[2][tid] = u2[g];
                }
                __syncthreads();
                // a wave takes four rows, its lanes the columns: short rows (small neighbourhoods, the top of a
                // triangle) still fill most of a wave, and a wave's stores are 512 contiguous bytes per row
                for (int r0 = 4 * wv; r0 < nac; r0 += 4 * (LT_TPB / 64)) {
                    const int nr = min(4, nac - r0), amax = ac + r0 + nr - 1;
                    for (int bl = lane; bl < nbc && bc + bl <= amax; bl += 64) {
                        const int b = bc + bl;
                        const double b0 = bu[0][bl], b1 = bu[1][bl], b2 = bu[2][bl];
                        // four rows of this column at once: chords / interval indices, then all 32 coefficient
                        // reads, then four interleaved Horner chains (the barriers keep hipcc from sinking every
                        // read next to its FMA -- one exposed LDS latency per Horner step otherwise)
                        unsigned need = 0, fast = 0;
                        double y[4], cf[CK_TAB_DEG + 1][4];
                        const double* lp[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int rr = min(r0 + r, nac - 1);
                            const int a = ac + r0 + r;
                            const double dx = au[0][rr] - b0, dy = au[1][rr] - b1, dz = au[2][rr] - b2;
                            int iv;
                            y[r] = ck_table_y(dx * dx + dy * dy + dz * dz, &iv, tbase);
                            const bool in = r < nr && b <= a, hit = (unsigned)iv < (unsigned)tn;
                            fast |= (in && hit) ? 1u << r : 0u;
                            need |= (in && !hit) ? 1u << r : 0u;
                            lp[r] = tab + min(max(iv, 0), max(tn - 1, 0));   // keep the lookup inside the table
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int kk = CK_TAB_DEG; kk >= 0; --kk)
#pragma unroll
                            for (int r = 0; r < 4; ++r) cf[kk][r] = lp[r][kk * CK_TAB_STRIDE];
                        __builtin_amdgcn_sched_barrier(0);
                        double pv[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[r] = cf[CK_TAB_DEG][r];
#pragma unroll
                        for (int kk = CK_TAB_DEG - 1; kk >= 0; --kk)
#pragma unroll
                            for (int r = 0; r < 4; ++r) pv[r] = fma(pv[r], y[r], cf[kk][r]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int a = ac + r0 + r;
                            if (fast >> r & 1u)
                                S[(long)a * ld + b] = pv[r];
                            else if ((need >> r & 1u) && b == a) {   // one per row: kept out of the slow loop below
                                S[(long)a * ld + b] = cdiag;
                                need &= ~(1u << r);
                            }
                        }
                        if (need) {
#pragma unroll 1
                            for (int r = 0; r < 4; ++r)
                                if (need >> r & 1u) {
                                    const int a = ac + r0 + r;
                                    const int slot = atomicAdd(&wl_n, 1);
                                    if (slot < LT_WL) {
                                        wl[slot] = make_int2(a, b);
                                    } else {   // list full: in place
                                        S[(long)a * ld + b] = lp_exact_pair(&blk[reg], metric, nug, s0, s1, s2, idx[a], idx[b]);
                                    }
                                }
                        }
                    }
                }
            }
        }
        __syncthreads();
        const int nw = min(wl_n, LT_WL);
        for (int e = tid; e < nw; e += LT_TPB) {
            const int a = wl[e].x, b = wl[e].y;
            S[(long)a * ld + b] = lp_exact_pair(&blk[reg], metric, nug, s0, s1, s2, idx[a], idx[b]);
        }
    }
}

// pred = v . y, var = c0 - v . v from the two solved rows
__global__ __launch_bounds__(LP_TPB) void k_local_reduce_t(const CkLocalSys* __restrict__ sys, const double* __restrict__ slab,
                                                            const long long* __restrict__ info, double c0var,
                                                            double* __restrict__ pred, double* __restrict__ err) {
    __shared__ double red[2][LP_TPB];
    const int tid = threadIdx.x;
    const CkLocalSys q = sys[blockIdx.x];
    if (info[blockIdx.x] != 0) {   // src/point_prediction.py:218-222
        if (tid == 0) {
            pred[q.p] = NAN;
            err[q.p] = NAN;
        }
        return;
    }
    const double* v = slab + q.off + (long)(q.kq - 2) * q.ld;
    const double* y = v + q.ld;
    double s1v = 0.0, s2v = 0.0;
    for (int a = tid; a < q.k; a += LP_TPB) {
        s1v += v[a] * y[a];
        s2v += v[a] * v[a];
    }
    red[0][tid] = s1v;
    red[1][tid] = s2v;
    __syncthreads();
    for (int s = LP_TPB / 2; s > 0; s >>= 1) {
        if (tid < s) {
            red[0][tid] += red[0][tid + s];
            red[1][tid] += red[1][tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        pred[q.p] = red[0][0];
        const double sd = sqrt(c0var - red[1][0]);
        err[q.p] = (sd == sd) ? fmax(sd, 0.0) : 0.0;   // np.nanmax([std, 0.0]), point_prediction.py:217
    }
}

void ck_launch_local_assemble_t(hipStream_t s, const CkMatern* blk, int metric, int i_pred, int cv, double max_dist,
                                const double* pc, int64_t mpad, const double* sc, const double* z, CkLayout L,
                                const CkLocalSys* sys, int n_sys, double* slab, const CkTable* tabs,
                                const double* const* coefs, int use_tab, const double* su, const double* pu,
                                const double* cb, double cmax, int* k0buf) {
    if (n_sys <= 0) return;
    const LpTab T{tabs, coefs, use_tab};
    const LpSearch R{cb, (long)((L.nend + LP_TPB - 1) / LP_TPB), cmax};
    k_local_search_t<<<dim3((unsigned)n_sys), dim3(LP_TPB), 0, s>>>(blk, metric, i_pred, cv, max_dist, pc, mpad, sc, z, L, sys,
                                                                    slab, T, su, pu, R, k0buf);
    const int nreg = L.nend > L.n0p ? 3 : 1;               // a second process?
    const int grid = n_sys < 512 ? n_sys : 512;            // two workgroups per CU; they walk over the systems
    for (int reg = 0; reg < nreg; ++reg)
        k_local_assemble_t<<<dim3((unsigned)grid), dim3(LT_TPB), 0, s>>>(blk, metric, reg, sc, L, sys, n_sys, slab, T, su, k0buf);
}

void ck_launch_local_reduce_t(hipStream_t s, const CkLocalSys* sys, int n_sys, const double* slab,
                              const long long* info, double c0var, double* pred, double* err) {
    if (n_sys <= 0) return;
    k_local_reduce_t<<<dim3((unsigned)n_sys), dim3(LP_TPB), 0, s>>>(sys, slab, info, c0var, pred, err);
}

int ck_local_lds_limit() { return LP_KL; }
