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
// Systems with k <= 124 neighbours live in LDS; larger ones in a global scratch slab per point.
#include "ck_internal.h"

#define LP_TPB 256
#define LP_KL 124            // (124 + 2) * 124 doubles = 125 KB of LDS

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

// neighbour count per prediction point
__global__ __launch_bounds__(LP_TPB) void k_local_count(int metric, int i_pred, int cv, double max_dist,
                                                         const double* __restrict__ pc, long mpad,
                                                         const double* __restrict__ sc, CkLayout L,
                                                         int* __restrict__ counts) {
    __shared__ int red[LP_TPB];
    const long p = blockIdx.x;
    const double p0 = pc[p], p1 = pc[mpad + p], p2 = pc[2 * mpad + p];
    const double *s0 = sc, *s1 = sc + L.npad, *s2 = sc + 2 * L.npad;
    int c = 0;
    for (long g = threadIdx.x; g < L.nend; g += LP_TPB)
        c += lp_is_neighbour(metric, cv, i_pred, max_dist, L, g, p0, p1, p2, s0, s1, s2) ? 1 : 0;
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
                                                         double* __restrict__ pred, double* __restrict__ err) {
    __shared__ double lS[(LP_KL + 2) * LP_KL];
    __shared__ int lidx[LP_KL];
    __shared__ int wsum[LP_TPB / 64];
    __shared__ int fail;
    __shared__ double red[2][LP_TPB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long p = blockIdx.x;
    const int k = counts[p];
    if (k == 0) {   // src/point_prediction.py:229-233
        if (tid == 0) {
            pred[p] = NAN;
            err[p] = NAN;
        }
        return;
    }
    const double p0 = pc[p], p1 = pc[mpad + p], p2 = pc[2 * mpad + p];
    const double *s0 = sc, *s1 = sc + L.npad, *s2 = sc + 2 * L.npad;
    // storage: (k + 2) x k matrix (rows k, k + 1 carry c and z) and the neighbour index list
    double* S;
    int* idx;
    long ld;
    if (k <= LP_KL) {
        S = lS;
        idx = lidx;
        ld = LP_KL;
    } else {
        S = slab + slab_off[p];
        idx = reinterpret_cast<int*>(S + (long)(k + 2) * k);
        ld = k;
    }
    // ---- 1. neighbour list, in site order (block-wide ordered compaction) ----
    if (tid == 0) fail = 0;
    int base = 0;
    for (long g0 = 0; g0 < L.nend; g0 += LP_TPB) {
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
        const double d = lp_dist(metric, s0[ga], s1[ga], s2[ga], s0[gb], s1[gb], s2[gb]);
        S[a * ld + b] = ck_cov_entry(blk[pa + pb], d, pa == pb);
    }
    for (int a = tid; a < k; a += LP_TPB) {
        const long ga = idx[a];
        const int pa = ga >= L.n0p;
        const double d = lp_dist(metric, p0, p1, p2, s0[ga], s1[ga], s2[ga]);
        S[(long)k * ld + a] = ck_cov_entry(blk[i_pred + pa], d, pa == i_pred);   // point_prediction.py:115-125
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

void ck_launch_local_count(hipStream_t s, int metric, int i_pred, int cv, double max_dist, const double* pc,
                           int64_t m, int64_t mpad, const double* sc, CkLayout L, int* counts) {
    if (m <= 0) return;
    k_local_count<<<dim3((unsigned)m), dim3(LP_TPB), 0, s>>>(metric, i_pred, cv, max_dist, pc, mpad, sc, L, counts);
}

void ck_launch_local_solve(hipStream_t s, const CkMatern* blk, int metric, int i_pred, int cv, double max_dist,
                           const double* pc, int64_t m, int64_t mpad, const double* sc, const double* z, CkLayout L,
                           const int* counts, const long long* slab_off, double* slab, double c0var, double* pred,
                           double* err) {
    if (m <= 0) return;
    k_local_solve<<<dim3((unsigned)m), dim3(LP_TPB), 0, s>>>(blk, metric, i_pred, cv, max_dist, pc, mpad, sc, z, L,
                                                             counts, slab_off, slab, c0var, pred, err);
}

int ck_local_lds_limit() { return LP_KL; }
