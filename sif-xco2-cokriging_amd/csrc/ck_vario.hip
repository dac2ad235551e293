// ck_vario.hip -- empirical (cross-)semivariogram / covariogram: pairwise lag binning.
//
// Replaces MultiField._variogram_cloud + get_variogram's pd.cut/groupby
// (src/fields.py:192-232, 378-386) without the dense n_i x n_j distance and cloud matrices:
// every pair is visited in registers, binned, and accumulated into a per-lane histogram in LDS.
//
// Distance: bins are decided on a monotone function r of the distance, so the inner loop needs
// no sqrt / asin / sin:
//   haversine  r = sin^2(theta / 2) = |u_i - u_j|^2 / 4   with u the unit vectors of the sites
//              (the same r as sklearn's rdist, src/fields.py:336; d = 2 R asin(sqrt(r)));
//   Euclidean  r = dx^2 + dy^2.
// Bin edges (data dependent, src/fields.py:389-403) are transformed to r-space on the host.
// Two passes, as the reference needs lo = min positive and hi = max retained distance before it
// can place the edges: pass 1 finds the two extreme pairs (their distances are then recomputed
// with the full-accuracy formula), pass 2 bins.
//
// Pass 2 layout: one workgroup = 256 "i" points in registers x chunks of 1024 "j" points that are
// the same for every lane and therefore come through scalar loads; each lane owns a private histogram (sum f64 + count u32 per bin) in LDS,
// laid out [bin][lane] so that the 64 lanes of a wave always hit distinct banks whatever bins
// they choose: no atomics, no conflicts, deterministic sums.  Workgroups walk the tile list with
// a fixed stride and write one partial histogram each; a second kernel adds the partials in a
// fixed order.
#include "ck_internal.h"

#define VG_TPB 256
#define VG_JCHUNK 1024
#define VG_MAXBINS 36
#define VG_LUT 8192
#define VG_G 8           // pairs per lane processed together (k_vario_bin); 16 measured slower
#define VG_W 8           // bins a tile's pairs may span for the compare-only binning (k_vario_bin)

struct VarioPartialExt {
    double rmin, rmax;
    long long imin, jmin, imax, jmax;
};

// per-site: haversine -> unit vector (x, y, z); Euclid -> (x, y, 0)
__global__ void k_vario_prep(const double* __restrict__ coords, long n, int metric, double* __restrict__ u0,
                             double* __restrict__ u1, double* __restrict__ u2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = coords[2 * i], b = coords[2 * i + 1];
    if (metric == CK_METRIC_HAVERSINE) {
        const double lat = a * CK_DEG2RAD, lon = b * CK_DEG2RAD;
        const double cl = cos(lat);
        u0[i] = cl * cos(lon);
        u1[i] = cl * sin(lon);
        u2[i] = sin(lat);
    } else {
        u0[i] = a;
        u1[i] = b;
        u2[i] = 0.0;
    }
}

__device__ __forceinline__ double pair_r(int metric, double ax, double ay, double az, double bx, double by,
                                         double bz) {
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    const double q = dx * dx + dy * dy + dz * dz;
    return metric == CK_METRIC_HAVERSINE ? 0.25 * q : q;
}

// tile t -> (I block, J chunk).  same != 0: only chunks that reach the strict upper triangle.
struct TileMap {
    long nI, nJ;
};

// ---- tile culling -------------------------------------------------------------------------------------
// The host lays the points out along a Hilbert curve (ck_api.hip: vario_upload), so the 256 "i" points of a
// tile and the 1024 "j" points of its chunk are two compact patches.  Per block of points: the mean c of its
// vectors u and rad = max |u - c|.  Every pair of a tile has |u_i - u_j| >= |c_I - c_J| - rad_I - rad_J; if that
// already exceeds the largest retained chord the whole tile is skipped -- with max_dist = 1 500 km on CONUS
// three tiles in four.  (A pair that is not retained costs the binning pass exactly what a retained one does.)
// bounds: 4 x nblk doubles (c.x, c.y, c.z, rad).
__global__ __launch_bounds__(VG_TPB) void k_vario_bounds(const double* __restrict__ u0, const double* __restrict__ u1,
                                                          const double* __restrict__ u2, long n, int blk, long nblk,
                                                          double* __restrict__ out) {
    __shared__ double red[3][VG_TPB];
    const int tid = threadIdx.x;
    const long lo = (long)blockIdx.x * blk, hi = (lo + blk < n) ? lo + blk : n;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (long g = lo + tid; g < hi; g += VG_TPB) {
        sx += u0[g];
        sy += u1[g];
        sz += u2[g];
    }
    red[0][tid] = sx;
    red[1][tid] = sy;
    red[2][tid] = sz;
    __syncthreads();
    for (int s = VG_TPB / 2; s > 0; s >>= 1) {
        if (tid < s)
            for (int c = 0; c < 3; ++c) red[c][tid] += red[c][tid + s];
        __syncthreads();
    }
    const double inv = 1.0 / (double)(hi - lo);
    const double cx = red[0][0] * inv, cy = red[1][0] * inv, cz = red[2][0] * inv;
    __syncthreads();
    double rm = 0.0;
    for (long g = lo + tid; g < hi; g += VG_TPB) {
        const double dx = u0[g] - cx, dy = u1[g] - cy, dz = u2[g] - cz;
        rm = fmax(rm, sqrt(dx * dx + dy * dy + dz * dz));
    }
    red[0][tid] = rm;
    __syncthreads();
    for (int s = VG_TPB / 2; s > 0; s >>= 1) {
        if (tid < s) red[0][tid] = fmax(red[0][tid], red[0][tid + s]);
        __syncthreads();
    }
    if (tid == 0) {
        out[blockIdx.x] = cx;
        out[nblk + blockIdx.x] = cy;
        out[2 * nblk + blockIdx.x] = cz;
        out[3 * nblk + blockIdx.x] = red[0][0] * (1.0 + 1e-12);
    }
}

// cmax: largest chord |u_i - u_j| a retained pair can have, with its safety margin (host); bounds may be null
__device__ __forceinline__ bool vario_tile_far(const double* __restrict__ ib, long nI, long bi,
                                               const double* __restrict__ jb, long nJ, long bj, double cmax) {
    if (!ib) return false;
    const double dx = ib[bi] - jb[bj], dy = ib[nI + bi] - jb[nJ + bj], dz = ib[2 * nI + bi] - jb[2 * nJ + bj];
    return sqrt(dx * dx + dy * dy + dz * dz) - ib[3 * nI + bi] - jb[3 * nJ + bj] > cmax;
}

// ---- pass 1: extreme pairs ------------------------------------------------------------------------
__global__ __launch_bounds__(VG_TPB) void k_vario_extent(int metric, int same, const double* __restrict__ iu0,
                                                          const double* __restrict__ iu1,
                                                          const double* __restrict__ iu2, long ni,
                                                          const double* __restrict__ ju0,
                                                          const double* __restrict__ ju1,
                                                          const double* __restrict__ ju2, long nj, double rcap,
                                                          VarioPartialExt* __restrict__ part, int rank, int world,
                                                          const double* __restrict__ ib, const double* __restrict__ jb,
                                                          double cmax, unsigned long long* best) {
    // best[0]: bit pattern of the largest retained r any workgroup has seen so far, best[1]: of the smallest
    // positive one (non-negative doubles order like their bit patterns).  A tile whose bounding balls say that all
    // its pairs lie strictly inside (rlo, rhi) with rhi < best[0] and rlo > best[1] cannot change either extreme
    // and is skipped; a stale hint only makes the test more conservative.  Nine tiles in ten go this way once the
    // first wave of workgroups has reported: the largest retained lag sits in the tiles that straddle max_dist,
    // the smallest positive one in tiles whose balls touch.
    __shared__ double red_r[VG_TPB];
    __shared__ long long red_i[VG_TPB], red_j[VG_TPB];
    const int tid = threadIdx.x;
    const long nI = (ni + VG_TPB - 1) / VG_TPB, nJ = (nj + VG_JCHUNK - 1) / VG_JCHUNK;
    double rmin = 1e300, rmax = -1.0;
    long long imin = -1, jmin = -1, imax = -1, jmax = -1;
    // tile list sharded over processes (ck_set_partition): this one takes the tiles t = rank (mod world)
    for (long t = (long)blockIdx.x * world + rank; t < nI * nJ; t += (long)gridDim.x * world) {
        const long bi = t / nJ, bj = t - bi * nJ;
        const long i0 = bi * VG_TPB, j0 = bj * VG_JCHUNK;
        if (same && j0 + VG_JCHUNK - 1 <= i0) continue;   // chunk entirely at or below the diagonal
        if (ib) {
            const double dx = ib[bi] - jb[bj], dy = ib[nI + bi] - jb[nJ + bj], dz = ib[2 * nI + bi] - jb[2 * nJ + bj];
            const double dc = sqrt(dx * dx + dy * dy + dz * dz), rr = ib[3 * nI + bi] + jb[3 * nJ + bj];
            if (dc - rr > cmax) continue;   // no pair of this tile within max_dist
            const double sc = metric == CK_METRIC_HAVERSINE ? 0.25 : 1.0;
            const double lo1 = fmax(dc - rr, 0.0) * (1.0 - 1e-9), hi1 = (dc + rr) * (1.0 + 1e-9) + 1e-12;
            const double rlo = sc * lo1 * lo1 * (1.0 - 1e-12), rhi = fmin(sc * hi1 * hi1 * (1.0 + 1e-12), rcap);
            const double bmax = __longlong_as_double((long long)__atomic_load_n(&best[0], __ATOMIC_RELAXED));
            const double bmin = __longlong_as_double((long long)__atomic_load_n(&best[1], __ATOMIC_RELAXED));
            if (rhi < bmax && rlo > bmin) continue;   // cannot hold a new extreme (uniform: same loads for all lanes)
        }
        const long i = i0 + tid;
        const bool live = i < ni;
        const long ic = live ? i : ni - 1;
        const double ax = iu0[ic], ay = iu1[ic], az = iu2[ic];
        const long jend = (nj - j0 < VG_JCHUNK) ? (nj - j0) : VG_JCHUNK;
        const long kbeg = same ? (i + 1 - j0) : 0;        // per lane
        long k0 = same ? (i0 + 1 - j0) : 0;               // uniform: the "j" point comes through scalar loads
        if (k0 < 0) k0 = 0;
#pragma unroll 8
        for (long k = k0; k < jend; ++k) {
            const double r = pair_r(metric, ax, ay, az, ju0[j0 + k], ju1[j0 + k], ju2[j0 + k]);
            if (live && k >= kbeg && r <= rcap) {
                if (r > rmax) {
                    rmax = r;
                    imax = i;
                    jmax = j0 + k;
                }
                if (r > 0.0 && r < rmin) {
                    rmin = r;
                    imin = i;
                    jmin = j0 + k;
                }
            }
        }
        if (ib) {   // publish this wave's extremes so far: hints for every workgroup's tile test above
            double wmax = rmax, wmin = rmin;
            for (int off = 32; off > 0; off >>= 1) {
                wmax = fmax(wmax, __shfl_xor(wmax, off));
                wmin = fmin(wmin, __shfl_xor(wmin, off));
            }
            if ((tid & 63) == 0) {
                if (wmax > 0.0) atomicMax(&best[0], (unsigned long long)__double_as_longlong(wmax));
                if (wmin < 1e300) atomicMin(&best[1], (unsigned long long)__double_as_longlong(wmin));
            }
        }
    }
    // workgroup reduction (min)
    __syncthreads();
    red_r[tid] = rmin;
    red_i[tid] = imin;
    red_j[tid] = jmin;
    __syncthreads();
    for (int s = VG_TPB / 2; s > 0; s >>= 1) {
        if (tid < s && red_r[tid + s] < red_r[tid]) {
            red_r[tid] = red_r[tid + s];
            red_i[tid] = red_i[tid + s];
            red_j[tid] = red_j[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        part[blockIdx.x].rmin = red_r[0];
        part[blockIdx.x].imin = red_i[0];
        part[blockIdx.x].jmin = red_j[0];
    }
    __syncthreads();
    red_r[tid] = rmax;
    red_i[tid] = imax;
    red_j[tid] = jmax;
    __syncthreads();
    for (int s = VG_TPB / 2; s > 0; s >>= 1) {
        if (tid < s && red_r[tid + s] > red_r[tid]) {
            red_r[tid] = red_r[tid + s];
            red_i[tid] = red_i[tid + s];
            red_j[tid] = red_j[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        part[blockIdx.x].rmax = red_r[0];
        part[blockIdx.x].imax = red_i[0];
        part[blockIdx.x].jmax = red_j[0];
    }
}

// ---- pass 2: binning ---------------------------------------------------------------------------------
// thr[0..nb]: r-space edges (thr[0] = 0).  Pair belongs to bin b if thr[b] < r <= thr[b+1]; r == 0
// goes to bin 0 (pd.cut include_lowest, src/fields.py:214-216).  lut[c] = bin of the lower end of
// r-cell c (cells uniform in r up to thr[nb]); at most one edge per cell is assumed by the +1 fix-up,
// which the host guarantees by choosing VG_LUT cells >= 4 / (narrowest bin in r-space).
__global__ __launch_bounds__(VG_TPB) void k_vario_bin(int metric, int same, int covariogram,
                                                       const double* __restrict__ iu0,
                                                       const double* __restrict__ iu1,
                                                       const double* __restrict__ iu2,
                                                       const double* __restrict__ iv, long ni,
                                                       const double* __restrict__ ju0,
                                                       const double* __restrict__ ju1,
                                                       const double* __restrict__ ju2,
                                                       const double* __restrict__ jv, long nj, double rcap,
                                                       int nb, const double* __restrict__ thr,
                                                       const unsigned char* __restrict__ lut, double inv_cell,
                                                       double* __restrict__ part_sum,
                                                       unsigned long long* __restrict__ part_cnt, int rank, int world,
                                                       const double* __restrict__ ib, const double* __restrict__ jb,
                                                       double cmax) {
    __shared__ double hsum[VG_MAXBINS + 1][VG_TPB];        // + 1: the trash row of pairs that are not retained
    __shared__ unsigned int hcnt[VG_MAXBINS + 1][VG_TPB];
    __shared__ double sthr[VG_MAXBINS + 2];
    __shared__ unsigned char slut[VG_LUT];
    const int tid = threadIdx.x;
    for (int b = 0; b <= nb; ++b) {
        hsum[b][tid] = 0.0;
        hcnt[b][tid] = 0u;
    }
    for (int k = tid; k <= nb; k += VG_TPB) sthr[k] = thr[k];
    for (int k = tid; k < VG_LUT; k += VG_TPB) slut[k] = lut[k];
    __syncthreads();
    const double rtop = thr[nb];
    // one group of VG_G consecutive "j" points for this lane's "i" point (see the comment in the loop)
#define VG_GROUP8(FULL)                                                                                         \
    {                                                                                                           \
        double r[VG_G], bv[VG_G];                                                                                     \
        int b[VG_G];                                                                                               \
        bool keep[VG_G];                                                                                           \
        _Pragma("unroll") for (int u = 0; u < VG_G; ++u) {                                                         \
            const long kk = (FULL) ? kg + u : (kg + u < jend ? kg + u : jend - 1);                              \
            r[u] = pair_r(metric, ax, ay, az, ju0[j0 + kk], ju1[j0 + kk], ju2[j0 + kk]);                        \
            bv[u] = jv[j0 + kk];                                                                                \
            keep[u] = live && ((FULL) || kg + u < jend) && kg + u >= kbeg && r[u] <= rcap && r[u] <= rtop;      \
            int c = (int)(fmin(r[u], rtop) * inv_cell);                                                         \
            c = c < VG_LUT - 1 ? c : VG_LUT - 1;                                                                \
            b[u] = slut[c];                                                                                     \
        }                                                                                                       \
        _Pragma("unroll") for (int pass = 0; pass < 2; ++pass) { /* up to two edges inside one r-cell */        \
            double e[VG_G];                                                                                        \
            _Pragma("unroll") for (int u = 0; u < VG_G; ++u) e[u] = sthr[b[u] + 1];                                \
            _Pragma("unroll") for (int u = 0; u < VG_G; ++u) {                                                     \
                b[u] += (r[u] > e[u]) ? 1 : 0;                                                                  \
                b[u] = b[u] < nb - 1 ? b[u] : nb - 1;                                                           \
            }                                                                                                   \
        }                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < VG_G; ++u) {                                                         \
            double cl;                                                                                          \
            if (covariogram) {                                                                                  \
                cl = av * bv[u]; /* fields.py:382-383 */                                                        \
            } else {                                                                                            \
                const double df = av - bv[u]; /* fields.py:384-385 */                                           \
                cl = 0.5 * (df * df);                                                                           \
            }                                                                                                   \
            const int bb = keep[u] ? b[u] : nb;                                                                 \
            __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)&hsum[bb][tid],      \
                                                keep[u] ? cl : 0.0);                                            \
            atomicAdd(&hcnt[bb][tid], keep[u] ? 1u : 0u);                                                       \
        }                                                                                                       \
    }
    // The same group for a tile whose pairs can only fall into the VG_W bins wb .. wb + VG_W - 1 (decided per tile
    // from the bounding balls): the bin is wb + the number of the window's inner edges below r -- compares against
    // seven wave-uniform thresholds instead of three dependent LDS lookups.
#define VG_GROUP8W(FULL)                                                                                        \
    {                                                                                                           \
        _Pragma("unroll") for (int u = 0; u < VG_G; ++u) {                                                         \
            const long kk = (FULL) ? kg + u : (kg + u < jend ? kg + u : jend - 1);                              \
            const double r = pair_r(metric, ax, ay, az, ju0[j0 + kk], ju1[j0 + kk], ju2[j0 + kk]);              \
            const double bvu = jv[j0 + kk];                                                                     \
            const bool keep = live && ((FULL) || kg + u < jend) && kg + u >= kbeg && r <= rcap && r <= rtop;    \
            int b = wb;                                                                                         \
            _Pragma("unroll") for (int w = 0; w < VG_W - 1; ++w) b += (r > wt[w]) ? 1 : 0;                         \
            double cl;                                                                                          \
            if (covariogram) {                                                                                  \
                cl = av * bvu;                                                                                  \
            } else {                                                                                            \
                const double df = av - bvu;                                                                     \
                cl = 0.5 * (df * df);                                                                           \
            }                                                                                                   \
            const int bb = keep ? b : nb;                                                                       \
            __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)&hsum[bb][tid],      \
                                                keep ? cl : 0.0);                                               \
            atomicAdd(&hcnt[bb][tid], keep ? 1u : 0u);                                                          \
        }                                                                                                       \
    }
    // ... and for an INTERIOR tile: every pair is retained (the balls say r <= min(rcap, rtop) for all of them, the
    // chunk lies strictly above the diagonal, all 256 "i" lanes are live) -- no per-pair conditions at all.
#define VG_GROUP8I                                                                                              \
    {                                                                                                           \
        _Pragma("unroll") for (int u = 0; u < VG_G; ++u) {                                                         \
            const long kk = kg + u;                                                                             \
            const double r = pair_r(metric, ax, ay, az, ju0[j0 + kk], ju1[j0 + kk], ju2[j0 + kk]);              \
            const double bvu = jv[j0 + kk];                                                                     \
            int b = wb;                                                                                         \
            _Pragma("unroll") for (int w = 0; w < VG_W - 1; ++w) b += (r > wt[w]) ? 1 : 0;                         \
            double cl;                                                                                          \
            if (covariogram) {                                                                                  \
                cl = av * bvu;                                                                                  \
            } else {                                                                                            \
                const double df = av - bvu;                                                                     \
                cl = 0.5 * (df * df);                                                                           \
            }                                                                                                   \
            __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)&hsum[b][tid], cl);  \
            atomicAdd(&hcnt[b][tid], 1u);                                                                       \
        }                                                                                                       \
    }
    const long nI = (ni + VG_TPB - 1) / VG_TPB, nJ = (nj + VG_JCHUNK - 1) / VG_JCHUNK;
    // per-lane 64-bit count spill: hcnt is 32-bit, a lane sees at most nJ * VG_JCHUNK pairs per I block
    // tile list sharded over processes (ck_set_partition): this one takes the tiles t = rank (mod world)
    for (long t = (long)blockIdx.x * world + rank; t < nI * nJ; t += (long)gridDim.x * world) {
        const long bi = t / nJ, bj = t - bi * nJ;
        const long i0 = bi * VG_TPB, j0 = bj * VG_JCHUNK;
        if (same && j0 + VG_JCHUNK - 1 <= i0) continue;
        // bounding balls of the two point blocks: chords of this tile's pairs lie in [dlo, dhi]
        double dlo = 0.0, dhi = 1e300;
        if (ib) {
            const double dx = ib[bi] - jb[bj], dy = ib[nI + bi] - jb[nJ + bj], dz = ib[2 * nI + bi] - jb[2 * nJ + bj];
            const double dc = sqrt(dx * dx + dy * dy + dz * dz), rr = ib[3 * nI + bi] + jb[3 * nJ + bj];
            dlo = dc - rr;
            dhi = dc + rr;
        }
        if (dlo > cmax) continue;   // no retained pair in this tile
        // bin window of the tile (margins far above the rounding of either side): every pair has
        // wb <= bin <= wbhi; the fast group needs the window to fit VG_W bins
        int wb = 0, wbhi = nb - 1;
        bool all_in = false;   // every pair of the tile within max_dist and the last edge
        if (ib) {
            const double sc = metric == CK_METRIC_HAVERSINE ? 0.25 : 1.0;
            const double lo1 = fmax(dlo, 0.0) * (1.0 - 1e-9), hi1 = dhi * (1.0 + 1e-9) + 1e-12;
            const double rlo = sc * lo1 * lo1 * (1.0 - 1e-12), rhi = sc * hi1 * hi1 * (1.0 + 1e-12);
            wbhi = 0;
            for (int e = 1; e < nb; ++e) {
                wb += (rlo > sthr[e]) ? 1 : 0;
                wbhi += (rhi > sthr[e]) ? 1 : 0;
            }
            all_in = rhi <= rcap && rhi <= rtop;
        }
        const bool narrow = ib && wbhi - wb < VG_W;
        bool interior = false;   // set below, once the tile's extent in i and j is known
        double wt[VG_W - 1];   // the window's inner edges thr[wb + 1 ..], +inf beyond the last bin
#pragma unroll
        for (int w = 0; w < VG_W - 1; ++w) wt[w] = (wb + 1 + w < nb) ? sthr[wb + 1 + w] : 1e300;
        const long i = i0 + tid;
        const bool live = i < ni;
        const long ic = live ? i : ni - 1;
        const double ax = iu0[ic], ay = iu1[ic], az = iu2[ic], av = iv[ic];
        const long jend = (nj - j0 < VG_JCHUNK) ? (nj - j0) : VG_JCHUNK;
        // The "j" point is the same for every lane of the workgroup: its coordinates and value come
        // through SCALAR loads (uniform address, 8 points per s_load_dwordx16 once unrolled) and sit
        // in SGPRs -- no LDS staging, no LDS reads in the pair loop.  In the triangular (same) case
        // the loop starts at the first column any lane of the block needs; each lane masks k < kbeg.
        const long kbeg = same ? (i + 1 - j0) : 0;                     // per lane
        long k0 = same ? (i0 + 1 - j0) : 0;                            // uniform
        if (k0 < 0) k0 = 0;
        // VG_G pairs at a time, in stages, so that the three dependent LDS lookups of a pair (r-cell
        // -> bin, then up to two edge fix-ups) are each issued for all eight before the first result
        // is needed; at one wave per SIMD (the private histograms fill the LDS) a pair-by-pair loop
        // pays those round trips one after the other -- hipcc does not software-pipeline them.  The
        // histogram update is a fire-and-forget LDS add into this lane's own slot (ds_add_f64 /
        // ds_add_u32): sequential per lane in program order, hence deterministic; pairs that are not
        // retained add 0 to a trash row (index nb).
        long kg = k0;
        interior = narrow && all_in && i0 + VG_TPB <= ni && (!same || j0 >= i0 + VG_TPB);
        if (interior) {
            for (; kg + VG_G <= jend; kg += VG_G) VG_GROUP8I;
            for (; kg < jend; kg += VG_G) VG_GROUP8W(false);
        } else if (narrow) {
            for (; kg + VG_G <= jend; kg += VG_G) VG_GROUP8W(true);
            for (; kg < jend; kg += VG_G) VG_GROUP8W(false);
        } else {
            for (; kg + VG_G <= jend; kg += VG_G)      // full groups: consecutive scalar loads merge into s_load_dwordx16
                VG_GROUP8(true);
            for (; kg < jend; kg += VG_G) VG_GROUP8(false);   // tail: indices clamped, pairs beyond jend masked
        }
    }
#undef VG_GROUP8
#undef VG_GROUP8W
#undef VG_GROUP8I
    __syncthreads();
    // reduce the 256 private histograms: thread b sums bin b in lane order (deterministic)
    if (tid < nb) {
        double s = 0.0;
        unsigned long long c = 0;
        for (int l = 0; l < VG_TPB; ++l) {
            s += hsum[tid][(l + tid) & (VG_TPB - 1)];   // skewed start: bank-conflict free, fixed order per bin
            c += hcnt[tid][(l + tid) & (VG_TPB - 1)];
        }
        part_sum[(long)blockIdx.x * VG_MAXBINS + tid] = s;
        part_cnt[(long)blockIdx.x * VG_MAXBINS + tid] = c;
    }
}

__global__ void k_vario_final(const double* __restrict__ part_sum, const unsigned long long* __restrict__ part_cnt,
                              int nparts, int nb, double* __restrict__ sums, long long* __restrict__ counts) {
    const int b = threadIdx.x;
    if (b >= nb) return;
    double s = 0.0;
    unsigned long long c = 0;
    for (int p = 0; p < nparts; ++p) {
        s += part_sum[(long)p * VG_MAXBINS + b];
        c += part_cnt[(long)p * VG_MAXBINS + b];
    }
    sums[b] = s;
    counts[b] = (long long)c;
}

// ---- launch wrappers ------------------------------------------------------------------------------------
void ck_launch_vario_prep(hipStream_t s, const double* coords, int64_t n, int metric, double* u0, double* u1,
                          double* u2) {
    if (n <= 0) return;
    k_vario_prep<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(coords, n, metric, u0, u1, u2);
}

int ck_vario_grid(int64_t ni, int64_t nj) {
    const int64_t nI = (ni + VG_TPB - 1) / VG_TPB, nJ = (nj + VG_JCHUNK - 1) / VG_JCHUNK;
    int64_t tiles = nI * nJ;
    if (tiles < 1) tiles = 1;
    return (int)(tiles < 2048 ? tiles : 2048);
}

void ck_launch_vario_extent(hipStream_t s, int grid, int metric, int same, const double* iu, int64_t ni,
                            const double* ju, int64_t nj, double rcap, void* part, int rank, int world,
                            const double* ib, const double* jb, double cmax, unsigned long long* best) {
    // best: two words of device memory, initialised here to "nothing seen yet" (largest retained r = 0.0, smallest
    // positive r = the largest finite double)
    static const unsigned long long init[2] = {0ULL, 0x7fefffffffffffffULL};
    (void)hipMemcpyAsync(best, init, sizeof(init), hipMemcpyHostToDevice, s);
    k_vario_extent<<<dim3(grid), dim3(VG_TPB), 0, s>>>(metric, same, iu, iu + ni, iu + 2 * ni, ni, ju, ju + nj,
                                                       ju + 2 * nj, nj, rcap, (VarioPartialExt*)part, rank, world, ib, jb,
                                                       cmax, best);
}

// bounding balls of the "i" blocks (VG_TPB points) or the "j" chunks (VG_JCHUNK points): 4 x nblk doubles
int64_t ck_vario_nblocks(int64_t n, int j_side) { return j_side ? (n + VG_JCHUNK - 1) / VG_JCHUNK : (n + VG_TPB - 1) / VG_TPB; }
void ck_launch_vario_bounds(hipStream_t s, const double* u, int64_t n, int j_side, double* out) {
    const int64_t nblk = ck_vario_nblocks(n, j_side);
    if (nblk <= 0) return;
    k_vario_bounds<<<dim3((unsigned)nblk), dim3(VG_TPB), 0, s>>>(u, u + n, u + 2 * n, n, j_side ? VG_JCHUNK : VG_TPB, nblk, out);
}

void ck_launch_vario_bin(hipStream_t s, int grid, int metric, int same, int covariogram, const double* iu,
                         const double* iv, int64_t ni, const double* ju, const double* jv, int64_t nj, double rcap,
                         int nb, const double* thr, const unsigned char* lut, double inv_cell, double* part_sum,
                         unsigned long long* part_cnt, double* sums, long long* counts, int rank, int world,
                         const double* ib, const double* jb, double cmax) {
    k_vario_bin<<<dim3(grid), dim3(VG_TPB), 0, s>>>(metric, same, covariogram, iu, iu + ni, iu + 2 * ni, iv, ni, ju,
                                                    ju + nj, ju + 2 * nj, jv, nj, rcap, nb, thr, lut, inv_cell,
                                                    part_sum, part_cnt, rank, world, ib, jb, cmax);
    k_vario_final<<<dim3(1), dim3(64), 0, s>>>(part_sum, part_cnt, grid, nb, sums, counts);
}
