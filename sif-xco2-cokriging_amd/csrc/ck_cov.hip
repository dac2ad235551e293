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
__global__ void k_prep_sites(const double* __restrict__ coords, long n, int metric, double* __restrict__ c0,
                             double* __restrict__ c1, double* __restrict__ c2, double* __restrict__ u) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = coords[2 * i], b = coords[2 * i + 1];
    if (metric == CK_METRIC_HAVERSINE) {
        const double lat = a * CK_DEG2RAD, lon = b * CK_DEG2RAD;   // numpy.radians (fields.py:334-335)
        const double cl = cos(lat);
        c0[i] = lat;
        c1[i] = lon;
        c2[i] = cl;
        if (u) {
            u[i] = cl * cos(lon);
            u[n + i] = cl * sin(lon);
            u[2 * n + i] = sin(lat);
        }
    } else {
        c0[i] = a;
        c1[i] = b;
        c2[i] = 0.0;
        if (u) {
            u[i] = a;
            u[n + i] = b;
            u[2 * n + i] = 0.0;
        }
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
// grid (CK_NB / 256, rows / 64), 256 threads; a workgroup walks four 64 x 64 sub-tiles.
// thread (ty = t >> 4, tx = t & 15) computes rows ty + 16 a (a = 0..3), cols 2 tx + {0, 1} + 32 b
// (b = 0, 1): every store instruction of a wave writes 4 rows x 256 contiguous bytes.
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

// number of entries the table path handed to the exact formulas since the last reset (diagnostic)
__device__ unsigned long long g_ck_fallback_entries = 0;

// One entry through the table.  The exact-formula call sits behind a WAVE-UNIFORM test (ballot):
// hipcc does not reliably skip a short divergent block that holds a call.
__device__ __forceinline__ double fast_entry(const CkMatern& m, const CkTable& tab, const double* lcoef, int metric,
                                             int nug, double ac0, double ac1, double ac2, double au0, double au1,
                                             double au2, double bc0, double bc1, double bc2, double bu0, double bu1,
                                             double bu2, bool valid) {
    const bool same = (ac0 == bc0 && ac1 == bc1);   // h == 0 exactly (model.py:195-196)
    const double dx = au0 - bu0, dy = au1 - bu1, dz = au2 - bu2;
    const double q = dx * dx + dy * dy + dz * dz;
    const bool in_tab = (q >= tab.q_lo && q < tab.q_hi);
    int iv;
    const double x = ck_table_x(in_tab ? q : tab.q_lo, &iv, tab.base);   // keep the lookup in range
    double val = m.amp * exp(ck_table_logrho(lcoef, tab.n_int, iv, x));
    if (same) val = nug ? m.amp + m.nugget : m.amp;
    const bool slow = valid && !same && !in_tab;   // padding lanes never ask for the exact formulas
    const unsigned long long sl = __builtin_amdgcn_ballot_w64(slow);
    if (sl != 0ULL) {
        if (slow) val = exact_entry_call(&m, metric, nug, ac0, ac1, ac2, bc0, bc1, bc2);
        if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(sl))
            atomicAdd(&g_ck_fallback_entries, (unsigned long long)__builtin_popcountll(sl));
    }
    return val;
}

// FAST: table path (every block's table enabled) | exact per-entry Bessel evaluation.
// AUX:  rows are prediction sites (row m = data values z, rows > m zero) | rows are data sites.
template <bool FAST, bool AUX>
__global__ __launch_bounds__(256) void k_assemble(const CkMatern* __restrict__ blk, const CkTable* __restrict__ tabs,
                                                   const double* const* __restrict__ coefs, int metric, int i_pred,
                                                   CkSiteRef R, long m, CkSiteRef S, const double* __restrict__ z,
                                                   CkLayout L, long row0, long col0, double* __restrict__ out) {
    __shared__ double lcoef[FAST ? (CK_TAB_DEG + 1) * CK_TAB_MAXINT : 1];
    __shared__ CkTable ltab;
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    const long rt = row0 + (long)blockIdx.y * 64;
    const int pr = AUX ? i_pred : (int)(rt >= L.n0p);
    // this thread's four rows
    double rc0[4], rc1[4], rc2[4], ru0[4], ru1[4], ru2[4];
    bool rv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long r = rt + ty + 16 * a;
        rv[a] = AUX ? (r < m) : site_valid(L, r);
        rc0[a] = R.c0[r];
        rc1[a] = R.c1[r];
        rc2[a] = R.c2[r];
        if (FAST) {
            ru0[a] = R.u0[r];
            ru1[a] = R.u1[r];
            ru2[a] = R.u2[r];
        }
    }
    int loaded = -1;
    for (int sub = 0; sub < 4; ++sub) {
        const long ct = col0 + (long)blockIdx.x * 256 + sub * 64;
        const int pc = (int)(ct >= L.n0p);
        const int bidx = pr + pc;
        const int nug = (pr == pc);
        const CkMatern& mb = blk[bidx];
        double* obase = out + (rt - row0) * CK_NB + (ct - col0);
        if (FAST && loaded != bidx) {
            __syncthreads();
            if (t == 0) ltab = tabs[bidx];
            const int cnt = (CK_TAB_DEG + 1) * tabs[bidx].n_int;
            const double* src = coefs[bidx];
            for (int k = t; k < cnt; k += 256) lcoef[k] = src[k];
            __syncthreads();
            loaded = bidx;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const long c = ct + 2 * tx + 32 * b;
            double cc0[2], cc1[2], cc2[2], cu0[2], cu1[2], cu2[2];
            bool cv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                cv[e] = site_valid(L, c + e);
                cc0[e] = S.c0[c + e];
                cc1[e] = S.c1[c + e];
                cc2[e] = S.c2[c + e];
                if (FAST) {
                    cu0[e] = S.u0[c + e];
                    cu1[e] = S.u1[c + e];
                    cu2[e] = S.u2[c + e];
                }
            }
            const double zc[2] = {AUX ? z[c] : 0.0, AUX ? z[c + 1] : 0.0};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const long r = rt + ty + 16 * a;
                d2_t v;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    double val = 0.0;
                    const bool valid = rv[a] && cv[e];
                    if (FAST) {
                        val = fast_entry(mb, ltab, lcoef, metric, nug, rc0[a], rc1[a], rc2[a], ru0[a], ru1[a], ru2[a],
                                         cc0[e], cc1[e], cc2[e], cu0[e], cu1[e], cu2[e], valid);
                    } else if (__builtin_amdgcn_ballot_w64(valid) != 0ULL) {
                        if (valid)
                            val = exact_entry_call(&mb, metric, nug, rc0[a], rc1[a], rc2[a], cc0[e], cc1[e], cc2[e]);
                    }
                    if (AUX) {
                        // rows: prediction sites | z | zero padding; padded columns are zero
                        if (!rv[a]) val = (r == m) ? zc[e] : 0.0;
                        if (!cv[e]) val = 0.0;
                    } else {
                        if (!(rv[a] && cv[e])) val = (r == c + e) ? 1.0 : 0.0;   // padding: identity
                    }
                    v[e] = val;
                }
                *reinterpret_cast<d2_t*>(obase + (ty + 16 * a) * CK_NB + 2 * tx + 32 * b) = v;
            }
        }
    }
}

void ck_launch_assemble_sigma_panel(hipStream_t s, bool fast, const CkMatern* blk, const CkTable* tabs,
                                    const double* const* coefs, int metric, const double* c, const double* u,
                                    CkLayout L, int64_t row0, int64_t nrows, int64_t col0, double* out) {
    if (nrows <= 0) return;
    const int64_t np = L.npad;
    CkSiteRef S{c, c + np, c + 2 * np, u, u + np, u + 2 * np};
    dim3 grid(CK_NB / 256, (unsigned)(nrows / 64));
    if (fast)
        k_assemble<true, false><<<grid, dim3(256), 0, s>>>(blk, tabs, coefs, metric, 0, S, 0, S, nullptr, L, row0,
                                                           col0, out);
    else
        k_assemble<false, false><<<grid, dim3(256), 0, s>>>(blk, tabs, coefs, metric, 0, S, 0, S, nullptr, L, row0,
                                                            col0, out);
}

void ck_launch_assemble_aux_panel(hipStream_t s, bool fast, const CkMatern* blk, const CkTable* tabs,
                                  const double* const* coefs, int metric, int i_pred, const double* pc,
                                  const double* pu, int64_t m, int64_t mpad, const double* c, const double* u,
                                  const double* z, CkLayout L, int64_t col0, double* out) {
    if (mpad <= 0) return;
    const int64_t np = L.npad;
    CkSiteRef P{pc, pc + mpad, pc + 2 * mpad, pu, pu + mpad, pu + 2 * mpad};
    CkSiteRef S{c, c + np, c + 2 * np, u, u + np, u + 2 * np};
    dim3 grid(CK_NB / 256, (unsigned)(mpad / 64));
    if (fast)
        k_assemble<true, true><<<grid, dim3(256), 0, s>>>(blk, tabs, coefs, metric, i_pred, P, m, S, z, L, 0, col0,
                                                          out);
    else
        k_assemble<false, true><<<grid, dim3(256), 0, s>>>(blk, tabs, coefs, metric, i_pred, P, m, S, z, L, 0, col0,
                                                           out);
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


// =========================================================================================
// Tabulated fast path (ck_math.h "Tabulated correlation")
// =========================================================================================
// log rho at given squared chords q (table construction; the polynomial fit is done on the host)
__global__ void k_table_nodes(const CkMatern* __restrict__ m, int metric, const double* __restrict__ q, long n,
                              double* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = log(ck_matern_rho_scaled(*m, ck_s_of_q(*m, metric, q[i])));
}

void ck_launch_table_nodes(hipStream_t s, const CkMatern* m, int metric, const double* q, int64_t n, double* out) {
    if (n <= 0) return;
    k_table_nodes<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(m, metric, q, n, out);
}

// max relative error of exp(P) against the exact evaluator, 8 probe points per interval
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
    const double x = ck_table_x(q, &iv, tab.base);
    const double got = exp(ck_table_logrho(coef, tab.n_int, iv, x));
    const double ref = ck_matern_rho_scaled(*m, ck_s_of_q(*m, metric, q));
    double e = 0.0;
    if (ref > 1e-290) e = fabs(got / ref - 1.0);
    else e = fabs(got - ref) > 1e-290 ? 1.0 : 0.0;
    if (iv != interval) e = 1.0;
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

int ck_fallback_counter(hipStream_t s, int reset, unsigned long long* out) {
    unsigned long long v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ck_fallback_entries), 8) != hipSuccess) return -1;
    if (out) *out = v;
    if (reset) {
        v = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_ck_fallback_entries), &v, 8) != hipSuccess) return -1;
    }
    return 0;
}
