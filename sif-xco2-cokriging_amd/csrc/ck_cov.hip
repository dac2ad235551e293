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

// Sigma block column ---------------------------------------------------------------------
// grid: (CK_NB / 64, nrows / 64); block 256 threads; tile 64 x 64.
// thread (ty = t >> 4, tx = t & 15) computes rows ty + 16 a (a = 0..3), cols 2 tx + {0, 1} + 32 b (b = 0, 1):
// every store instruction of a wave writes 4 rows x 256 contiguous bytes.
__global__ __launch_bounds__(256) void k_assemble_sigma(const CkMatern* __restrict__ blk, int metric,
                                                         const double* __restrict__ s0,
                                                         const double* __restrict__ s1,
                                                         const double* __restrict__ s2, long n0, long N, long row0,
                                                         long col0, double* __restrict__ out) {
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    const long rt = row0 + (long)blockIdx.y * 64, ct = col0 + (long)blockIdx.x * 64;
    // (a block column starts at its diagonal 512-block, which is written in full; everything
    //  below it is in the lower triangle)
    // uniform Matern block for the whole tile?
    const int prA = rt >= n0, prB = (rt + 63) >= n0, pcA = ct >= n0, pcB = (ct + 63) >= n0;
    const bool uni = (prA == prB) && (pcA == pcB) && (rt + 63 < N) && (ct + 63 < N);
    double* obase = out + (rt - row0) * CK_NB + (ct - col0);
    if (uni) {
        const CkMatern& m = blk[prA + pcA];
        const int nug = (prA == pcA);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const long c = ct + 2 * tx + 32 * b;
            const double b00 = s0[c], b01 = s1[c], b02 = s2[c];
            const double b10 = s0[c + 1], b11 = s1[c + 1], b12 = s2[c + 1];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const long r = rt + ty + 16 * a;
                const double a0 = s0[r], a1 = s1[r], a2 = s2[r];
                d2_t v;
                v[0] = ck_cov_entry(m, pair_dist(metric, a0, a1, a2, b00, b01, b02), nug);
                v[1] = ck_cov_entry(m, pair_dist(metric, a0, a1, a2, b10, b11, b12), nug);
                *reinterpret_cast<d2_t*>(obase + (ty + 16 * a) * CK_NB + 2 * tx + 32 * b) = v;
            }
        }
    } else {
        for (int b = 0; b < 2; ++b) {
            for (int a = 0; a < 4; ++a) {
                const long r = rt + ty + 16 * a;
                d2_t v;
                for (int e = 0; e < 2; ++e) {
                    const long c = ct + 2 * tx + 32 * b + e;
                    double val;
                    if (r >= N || c >= N) {
                        val = (r == c) ? 1.0 : 0.0;   // padding: identity
                    } else {
                        const int pr = r >= n0, pc = c >= n0;
                        val = ck_cov_entry(blk[pr + pc], pair_dist(metric, s0[r], s1[r], s2[r], s0[c], s1[c], s2[c]),
                                           pr == pc);
                    }
                    v[e] = val;
                }
                *reinterpret_cast<d2_t*>(obase + (ty + 16 * a) * CK_NB + 2 * tx + 32 * b) = v;
            }
        }
    }
}

void ck_launch_assemble_sigma_panel(hipStream_t s, const CkMatern* blk, int metric, const double* s0,
                                    const double* s1, const double* s2, int64_t n0, int64_t N, int64_t row0,
                                    int64_t nrows, int64_t col0, double* out) {
    if (nrows <= 0) return;
    k_assemble_sigma<<<dim3(CK_NB / 64, (unsigned)(nrows / 64)), dim3(256), 0, s>>>(blk, metric, s0, s1, s2, n0, N,
                                                                                    row0, col0, out);
}

// right-hand-side rows (c0^T and z^T) ----------------------------------------------------
// rows p < m: cov(prediction site p, data site c): C_ii with nugget-at-zero when the data site
// belongs to the predicted process, else C_12 (src/joint_prediction.py:114-121).
// row m: data values z (src/joint_prediction.py:67).  rows > m and padded columns: 0.
__global__ __launch_bounds__(256) void k_assemble_aux(const CkMatern* __restrict__ blk, int metric, int i_pred,
                                                       const double* __restrict__ p0, const double* __restrict__ p1,
                                                       const double* __restrict__ p2, long m,
                                                       const double* __restrict__ s0, const double* __restrict__ s1,
                                                       const double* __restrict__ s2, const double* __restrict__ z,
                                                       long n0, long N, long col0, double* __restrict__ out) {
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    const long rt = (long)blockIdx.y * 64, ct = col0 + (long)blockIdx.x * 64;
    double* obase = out + rt * CK_NB + (ct - col0);
    const int pcA = ct >= n0, pcB = (ct + 63) >= n0;
    const bool uni = (pcA == pcB) && (ct + 63 < N) && (rt + 63 < m);
    if (uni) {
        const int nug = (pcA == i_pred);
        const CkMatern& mb = blk[i_pred + pcA];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const long c = ct + 2 * tx + 32 * b;
            const double b00 = s0[c], b01 = s1[c], b02 = s2[c];
            const double b10 = s0[c + 1], b11 = s1[c + 1], b12 = s2[c + 1];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const long r = rt + ty + 16 * a;
                const double a0 = p0[r], a1 = p1[r], a2 = p2[r];
                d2_t v;
                v[0] = ck_cov_entry(mb, pair_dist(metric, a0, a1, a2, b00, b01, b02), nug);
                v[1] = ck_cov_entry(mb, pair_dist(metric, a0, a1, a2, b10, b11, b12), nug);
                *reinterpret_cast<d2_t*>(obase + (ty + 16 * a) * CK_NB + 2 * tx + 32 * b) = v;
            }
        }
    } else {
        for (int b = 0; b < 2; ++b) {
            for (int a = 0; a < 4; ++a) {
                const long r = rt + ty + 16 * a;
                d2_t v;
                for (int e = 0; e < 2; ++e) {
                    const long c = ct + 2 * tx + 32 * b + e;
                    double val = 0.0;
                    if (c < N) {
                        if (r < m) {
                            const int pc = c >= n0;
                            val = ck_cov_entry(blk[i_pred + pc],
                                               pair_dist(metric, p0[r], p1[r], p2[r], s0[c], s1[c], s2[c]),
                                               pc == i_pred);
                        } else if (r == m) {
                            val = z[c];
                        }
                    }
                    v[e] = val;
                }
                *reinterpret_cast<d2_t*>(obase + (ty + 16 * a) * CK_NB + 2 * tx + 32 * b) = v;
            }
        }
    }
}

void ck_launch_assemble_aux_panel(hipStream_t s, const CkMatern* blk, int metric, int i_pred, const double* p0,
                                  const double* p1, const double* p2, int64_t m, int64_t mpad, const double* s0,
                                  const double* s1, const double* s2, const double* z, int64_t n0, int64_t N,
                                  int64_t col0, double* out) {
    if (mpad <= 0) return;
    k_assemble_aux<<<dim3(CK_NB / 64, (unsigned)(mpad / 64)), dim3(256), 0, s>>>(blk, metric, i_pred, p0, p1, p2, m,
                                                                                 s0, s1, s2, z, n0, N, col0, out);
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

struct CkSiteRef {   // SoA views of one site set
    const double *c0, *c1, *c2;   // exact-formula coordinates
    const double *u0, *u1, *u2;   // chord vectors
};

// the exact formulas as an out-of-line call: rare (pairs closer than the table's lower end or
// beyond its upper end), and keeping it out of line keeps the table path's register count low
__device__ __noinline__ double exact_entry_call(const CkMatern* m, int metric, int nug, double ac0, double ac1,
                                                double ac2, double bc0, double bc1, double bc2) {
    return ck_cov_entry(*m, pair_dist(metric, ac0, ac1, ac2, bc0, bc1, bc2), nug);
}

// one entry through the table, falling back to the exact formulas outside its range
__device__ __forceinline__ double fast_entry(const CkMatern& m, const CkTable& tab, const double* lcoef, int metric,
                                             int nug, double ac0, double ac1, double ac2, double au0, double au1,
                                             double au2, double bc0, double bc1, double bc2, double bu0, double bu1,
                                             double bu2) {
    if (ac0 == bc0 && ac1 == bc1) return nug ? m.amp + m.nugget : m.amp;   // h == 0 exactly (model.py:195-196)
    const double dx = au0 - bu0, dy = au1 - bu1, dz = au2 - bu2;
    const double q = dx * dx + dy * dy + dz * dz;
    if (q >= tab.q_lo && q < tab.q_hi) {
        int iv;
        const double x = ck_table_x(q, &iv, tab.base);
        return m.amp * exp(ck_table_logrho(lcoef, tab.n_int, iv, x));
    }
    return exact_entry_call(&m, metric, nug, ac0, ac1, ac2, bc0, bc1, bc2);
}

// Sigma block column, table path.  grid (CK_NB / 256, nrows / 64); a workgroup walks four 64 x 64
// sub-tiles so that the table (<= 28 KB, staged in LDS) is loaded once per 128 KB of output.
__global__ __launch_bounds__(256) void k_assemble_sigma_fast(const CkMatern* __restrict__ blk,
                                                              const CkTable* __restrict__ tabs,
                                                              const double* const* __restrict__ coefs, int metric,
                                                              CkSiteRef S, long n0, long N, long row0, long col0,
                                                              double* __restrict__ out) {
    __shared__ double lcoef[(CK_TAB_DEG + 1) * CK_TAB_MAXINT];
    __shared__ CkTable ltab;
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    const long rt = row0 + (long)blockIdx.y * 64;
    int loaded = -1;
    for (int sub = 0; sub < 4; ++sub) {
        const long ct = col0 + (long)blockIdx.x * 256 + sub * 64;
        const int prA = rt >= n0, prB = (rt + 63) >= n0, pcA = ct >= n0, pcB = (ct + 63) >= n0;
        const bool uni = (prA == prB) && (pcA == pcB) && (rt + 63 < N) && (ct + 63 < N);
        const int bidx = prA + pcA;
        double* obase = out + (rt - row0) * CK_NB + (ct - col0);
        if (uni && tabs[bidx].enabled) {
            if (loaded != bidx) {
                __syncthreads();
                if (t == 0) ltab = tabs[bidx];
                const int cnt = (CK_TAB_DEG + 1) * tabs[bidx].n_int;
                const double* src = coefs[bidx];
                for (int k = t; k < cnt; k += 256) lcoef[k] = src[k];
                __syncthreads();
                loaded = bidx;
            }
            const CkMatern& m = blk[bidx];
            const int nug = (prA == pcA);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const long c = ct + 2 * tx + 32 * b;
                const double b0c0 = S.c0[c], b0c1 = S.c1[c], b0c2 = S.c2[c], b0u0 = S.u0[c], b0u1 = S.u1[c],
                             b0u2 = S.u2[c];
                const double b1c0 = S.c0[c + 1], b1c1 = S.c1[c + 1], b1c2 = S.c2[c + 1], b1u0 = S.u0[c + 1],
                             b1u1 = S.u1[c + 1], b1u2 = S.u2[c + 1];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const long r = rt + ty + 16 * a;
                    const double ac0 = S.c0[r], ac1 = S.c1[r], ac2 = S.c2[r], au0 = S.u0[r], au1 = S.u1[r],
                                 au2 = S.u2[r];
                    d2_t v;
                    v[0] = fast_entry(m, ltab, lcoef, metric, nug, ac0, ac1, ac2, au0, au1, au2, b0c0, b0c1, b0c2,
                                      b0u0, b0u1, b0u2);
                    v[1] = fast_entry(m, ltab, lcoef, metric, nug, ac0, ac1, ac2, au0, au1, au2, b1c0, b1c1, b1c2,
                                      b1u0, b1u1, b1u2);
                    *reinterpret_cast<d2_t*>(obase + (ty + 16 * a) * CK_NB + 2 * tx + 32 * b) = v;
                }
            }
        } else {
            // block boundary / padding / table disabled: exact formulas, per-entry block choice
            for (int b = 0; b < 2; ++b) {
                for (int a = 0; a < 4; ++a) {
                    const long r = rt + ty + 16 * a;
                    d2_t v;
                    for (int e = 0; e < 2; ++e) {
                        const long c = ct + 2 * tx + 32 * b + e;
                        double val;
                        if (r >= N || c >= N) {
                            val = (r == c) ? 1.0 : 0.0;
                        } else {
                            const int pr = r >= n0, pc = c >= n0;
                            val = exact_entry_call(&blk[pr + pc], metric, pr == pc, S.c0[r], S.c1[r], S.c2[r], S.c0[c],
                                                   S.c1[c], S.c2[c]);
                        }
                        v[e] = val;
                    }
                    *reinterpret_cast<d2_t*>(obase + (ty + 16 * a) * CK_NB + 2 * tx + 32 * b) = v;
                }
            }
        }
    }
}

void ck_launch_assemble_sigma_panel_fast(hipStream_t s, const CkMatern* blk, const CkTable* tabs,
                                         const double* const* coefs, int metric, const double* c, const double* u,
                                         int64_t npad, int64_t n0, int64_t N, int64_t row0, int64_t nrows,
                                         int64_t col0, double* out) {
    if (nrows <= 0) return;
    CkSiteRef S{c, c + npad, c + 2 * npad, u, u + npad, u + 2 * npad};
    k_assemble_sigma_fast<<<dim3(CK_NB / 256, (unsigned)(nrows / 64)), dim3(256), 0, s>>>(blk, tabs, coefs, metric, S,
                                                                                          n0, N, row0, col0, out);
}

// right-hand-side rows, table path (same structure; rows = prediction sites)
__global__ __launch_bounds__(256) void k_assemble_aux_fast(const CkMatern* __restrict__ blk,
                                                            const CkTable* __restrict__ tabs,
                                                            const double* const* __restrict__ coefs, int metric,
                                                            int i_pred, CkSiteRef P, long m, CkSiteRef S,
                                                            const double* __restrict__ z, long n0, long N, long col0,
                                                            double* __restrict__ out) {
    __shared__ double lcoef[(CK_TAB_DEG + 1) * CK_TAB_MAXINT];
    __shared__ CkTable ltab;
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    const long rt = (long)blockIdx.y * 64;
    int loaded = -1;
    for (int sub = 0; sub < 4; ++sub) {
        const long ct = col0 + (long)blockIdx.x * 256 + sub * 64;
        const int pcA = ct >= n0, pcB = (ct + 63) >= n0;
        const bool uni = (pcA == pcB) && (ct + 63 < N) && (rt + 63 < m);
        const int bidx = i_pred + pcA;
        double* obase = out + rt * CK_NB + (ct - col0);
        if (uni && tabs[bidx].enabled) {
            if (loaded != bidx) {
                __syncthreads();
                if (t == 0) ltab = tabs[bidx];
                const int cnt = (CK_TAB_DEG + 1) * tabs[bidx].n_int;
                const double* src = coefs[bidx];
                for (int k = t; k < cnt; k += 256) lcoef[k] = src[k];
                __syncthreads();
                loaded = bidx;
            }
            const CkMatern& mb = blk[bidx];
            const int nug = (pcA == i_pred);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const long c = ct + 2 * tx + 32 * b;
                const double b0c0 = S.c0[c], b0c1 = S.c1[c], b0c2 = S.c2[c], b0u0 = S.u0[c], b0u1 = S.u1[c],
                             b0u2 = S.u2[c];
                const double b1c0 = S.c0[c + 1], b1c1 = S.c1[c + 1], b1c2 = S.c2[c + 1], b1u0 = S.u0[c + 1],
                             b1u1 = S.u1[c + 1], b1u2 = S.u2[c + 1];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const long r = rt + ty + 16 * a;
                    const double ac0 = P.c0[r], ac1 = P.c1[r], ac2 = P.c2[r], au0 = P.u0[r], au1 = P.u1[r],
                                 au2 = P.u2[r];
                    d2_t v;
                    v[0] = fast_entry(mb, ltab, lcoef, metric, nug, ac0, ac1, ac2, au0, au1, au2, b0c0, b0c1, b0c2,
                                      b0u0, b0u1, b0u2);
                    v[1] = fast_entry(mb, ltab, lcoef, metric, nug, ac0, ac1, ac2, au0, au1, au2, b1c0, b1c1, b1c2,
                                      b1u0, b1u1, b1u2);
                    *reinterpret_cast<d2_t*>(obase + (ty + 16 * a) * CK_NB + 2 * tx + 32 * b) = v;
                }
            }
        } else {
            for (int b = 0; b < 2; ++b) {
                for (int a = 0; a < 4; ++a) {
                    const long r = rt + ty + 16 * a;
                    d2_t v;
                    for (int e = 0; e < 2; ++e) {
                        const long c = ct + 2 * tx + 32 * b + e;
                        double val = 0.0;
                        if (c < N) {
                            if (r < m) {
                                const int pc = c >= n0;
                                val = exact_entry_call(&blk[i_pred + pc], metric, pc == i_pred, P.c0[r], P.c1[r], P.c2[r],
                                                       S.c0[c], S.c1[c], S.c2[c]);
                            } else if (r == m) {
                                val = z[c];
                            }
                        }
                        v[e] = val;
                    }
                    *reinterpret_cast<d2_t*>(obase + (ty + 16 * a) * CK_NB + 2 * tx + 32 * b) = v;
                }
            }
        }
    }
}

void ck_launch_assemble_aux_panel_fast(hipStream_t s, const CkMatern* blk, const CkTable* tabs,
                                       const double* const* coefs, int metric, int i_pred, const double* pc,
                                       const double* pu, int64_t m, int64_t mpad, const double* c, const double* u,
                                       int64_t npad, const double* z, int64_t n0, int64_t N, int64_t col0,
                                       double* out) {
    if (mpad <= 0) return;
    CkSiteRef P{pc, pc + mpad, pc + 2 * mpad, pu, pu + mpad, pu + 2 * mpad};
    CkSiteRef S{c, c + npad, c + 2 * npad, u, u + npad, u + 2 * npad};
    k_assemble_aux_fast<<<dim3(CK_NB / 256, (unsigned)(mpad / 64)), dim3(256), 0, s>>>(blk, tabs, coefs, metric,
                                                                                       i_pred, P, m, S, z, n0, N,
                                                                                       col0, out);
}
