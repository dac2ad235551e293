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
__global__ void k_prep_sites(const double* __restrict__ coords, long n, int metric, double* __restrict__ c0,
                             double* __restrict__ c1, double* __restrict__ c2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = coords[2 * i], b = coords[2 * i + 1];
    if (metric == CK_METRIC_HAVERSINE) {
        const double lat = a * CK_DEG2RAD, lon = b * CK_DEG2RAD;   // numpy.radians (fields.py:334-335)
        c0[i] = lat;
        c1[i] = lon;
        c2[i] = cos(lat);
    } else {
        c0[i] = a;
        c1[i] = b;
        c2[i] = 0.0;
    }
}

void ck_launch_prep_sites(hipStream_t s, const double* coords, int64_t n, int metric, double* c0, double* c1,
                          double* c2) {
    if (n <= 0) return;
    k_prep_sites<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(coords, n, metric, c0, c1, c2);
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
