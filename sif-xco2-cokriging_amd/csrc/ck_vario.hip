// ck_vario.hip -- empirical (cross-)semivariogram / covariogram: pairwise lag binning.
//
// Replaces MultiField._variogram_cloud + get_variogram's pd.cut/groupby
// (src/fields.py:192-232, 378-386) without the dense n_i x n_j distance and cloud matrices:
// every pair is visited in registers and accumulated into per-lane REGISTER accumulators.
//
// Distance.  The reference decides everything on the rounded distance d: `distance <= max_dist`, then
// pd.cut on the edges (src/fields.py:212-216).  The kernels work on a monotone function q of d that needs
// no sqrt / asin / sin per pair:
//   haversine  q = |u_i - u_j|^2 = 4 sin^2(theta / 2), u the unit vectors of the sites (d = 2 R asin(sqrt(q) / 2));
//   Euclidean  q = dx^2 + dy^2.
// A pair whose q lies within the rounding band of a threshold (a bin edge or max_dist; the band is a few
// 1e-15 sqrt(q), ck_api.hip: vario_band) is NOT decided in q-space: it is re-decided with the reference's
// own formula and its own comparison (`d > edge`) -- Euclidean: on the device, sqrt(fl(dx dx) + fl(dy dy))
// without contraction is bit-identical to scipy's cdist; haversine: the pair goes to a list and the host
// decides it with libm's sin / cos / asin, which is what sklearn's haversine_distances calls (device and
// numpy SIMD trigonometry differ from libm in the last bit of ~8 % of the distances).  On lattice data, where
// many pairs sit at exactly the same distance and max_dist / the edges can coincide with lattice distances,
// that is what makes the integer counts the reference's.
//
// Two passes, as the reference needs lo = min positive and hi = max retained distance before it can place
// the edges (src/fields.py:389-403): pass 1 (k_vario_extent, then k_vario_collect for every pair within the
// band of the two extremes, decided on the host) and pass 2 (k_vario_bin).
//
// Pass 2.  "Levels" 1 .. E are the thresholds in ascending order: the inner edges below the cap, then the
// cap min(max_dist, last edge); the bin of a pair is the number of levels it passes (d > threshold), a pair
// that passes level E is not retained.  One WAVE owns a pair tile of 64 "i" points (one per lane, in
// registers) x 1024 "j" points (the same for every lane: scalar loads); per 256-point sub-chunk the bounding
// balls of the two point blocks say which levels every pair passes (nlow) and which none can reach (> nhigh),
// so a pair is compared against nhigh - nlow thresholds only -- with the points in Hilbert order 1 to 3.  A lane
// keeps one sum and one count per bin of the window in registers (slot k = bin nlow + k): nested compares on the
// way up, one FMA and one integer add where the pair stops.  They are reduced over the wave and added to the wave's
// histogram in LDS only when nlow changes.  No LDS traffic and no atomics in the pair loop (the first version
// kept per-lane histograms in LDS -- 113 KB, one wave per SIMD, two LDS atomics per pair).
// Sums are deterministic: fixed tile -> wave assignment, fixed reduction orders.
#include "ck_internal.h"

#include <stdlib.h>

#include <utility>

#define VG_TPB 256
#define VG_JCHUNK 1024   // "j" points of a pair tile
#define VG_JSUB 256      // sub-chunk: unit of the level-window decision of the binning pass
#define VG_IW 64         // "i" points of a wave tile (binning pass)
#define VG_MAXBINS CK_VG_MAXBINS
#define VG_SLOTS 9       // slot 0: base level (passed by every pair of the sub-chunk); slots 1..8 compared

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

__device__ __forceinline__ double pair_q(double ax, double ay, double az, double bx, double by, double bz) {
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    return dx * dx + dy * dy + dz * dz;
}

// scipy's cdist (src/fields.py:342) for two columns: sqrt(fl(fl(dx dx) + fl(dy dy))), no contraction
__device__ __forceinline__ double euclid_exact(double ax, double ay, double bx, double by) {
#pragma clang fp contract(off)
    const double dx = ax - bx, dy = ay - by;
    const double sx = dx * dx;
    const double sy = dy * dy;
    return __builtin_sqrt(sx + sy);
}

// ---- tile culling -------------------------------------------------------------------------------------
// The host lays the points out along a Hilbert curve (ck_api.hip: vario_upload), so a block of consecutive
// points is a compact patch.  Per block of `blk` points: the mean c of its vectors u and rad = max |u - c|.
// Every pair of two blocks has |c_I - c_J| - rad_I - rad_J <= |u_i - u_j| <= |c_I - c_J| + rad_I + rad_J.
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

// conservative [qlo, qhi] of the squared chords between block bi of `ib` and block bj of `jb`; *dlo = lower chord bound
template <class P>
__device__ __forceinline__ void tile_q_range(P ib, long nI, long bi, P jb, long nJ, long bj, double* dlo, double* qlo,
                                             double* qhi) {
    const double dx = ib[bi] - jb[bj], dy = ib[nI + bi] - jb[nJ + bj], dz = ib[2 * nI + bi] - jb[2 * nJ + bj];
    const double dc = sqrt(dx * dx + dy * dy + dz * dz), rr = ib[3 * nI + bi] + jb[3 * nJ + bj];
    *dlo = dc - rr;
    const double lo1 = fmax(dc - rr, 0.0) * (1.0 - 1e-9), hi1 = (dc + rr) * (1.0 + 1e-9) + 1e-12;
    *qlo = lo1 * lo1 * (1.0 - 1e-12);
    *qhi = hi1 * hi1 * (1.0 + 1e-12);
}

// Read-only data at wave-uniform addresses ("j" points, thresholds, bounding balls) is read through the CONSTANT
// address space: hipcc then fetches it with scalar loads into SGPRs whatever stores and atomics the kernel also
// contains (with plain global pointers -- even const __restrict__ kernel parameters -- the list append's atomic in the
// same loop made it fall back to per-lane vector loads of one and the same address).  Nothing writes these arrays
// while the kernels run.
typedef const double __attribute__((address_space(4))) * vg_cptr;
__device__ __forceinline__ vg_cptr vg_const(const double* p) { return (vg_cptr)(uintptr_t)p; }

// ---- pass 1a: extreme pairs in q-space --------------------------------------------------------------
// Largest q <= qcap and smallest positive q over this process's pair tiles (qcap already carries the upper
// band of max_dist: the host decides the pairs near it).  Same tiling as the binning pass: a wave owns 64 "i"
// points x 1024 "j" points and decides per 256-point sub-chunk whether it can hold a new extreme.
struct VarioExtArgs {
    int same, rank, world;
    const double *iu0, *iu1, *iu2;
    long ni;
    const double *ju0, *ju1, *ju2;
    long nj;
    const double *ib, *jb, *jsb;   // bounding balls: 64-point "i" blocks, 1024-point "j" chunks, 256-point sub-chunks
    double qcap, cmax;
};

// Pairs with qwin_lo <= q <= qcap are appended to `list` on the way: with dense data the largest retained q lies within
// that thin window under the cap, and the host then has every candidate for the largest distance without a second pass.
__global__ __launch_bounds__(VG_TPB) void k_vario_extent(const VarioExtArgs a, VarioPartialExt* __restrict__ part,
                                                          unsigned long long* best, double qwin_lo,
                                                          CkVarioPair* __restrict__ list, unsigned* __restrict__ count,
                                                          unsigned cap) {
    // best[0]: bit pattern of the largest retained q any wave has seen so far, best[1]: of the smallest positive
    // one (non-negative doubles order like their bit patterns).  A (sub-)tile whose bounding balls say that all its
    // pairs lie strictly inside (qlo, qhi) with qhi < best[0] and qlo > best[1] cannot change either extreme and is
    // skipped; a stale hint only makes the test more conservative.  Once the first waves have reported, what is left
    // are the sub-tiles that straddle max_dist (largest retained lag) and those whose balls touch (smallest).
    __shared__ double red_r[VG_TPB];
    __shared__ long long red_i[VG_TPB], red_j[VG_TPB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const vg_cptr ju0 = vg_const(a.ju0), ju1 = vg_const(a.ju1), ju2 = vg_const(a.ju2);
    const long ni = a.ni, nj = a.nj;
    const long nIw = (ni + VG_IW - 1) / VG_IW, nJ = (nj + VG_JCHUNK - 1) / VG_JCHUNK, nJs = (nj + VG_JSUB - 1) / VG_JSUB;
    const long nwaves = (long)gridDim.x * (VG_TPB / 64), wid = (long)blockIdx.x * (VG_TPB / 64) + wv;
    const double qcap = a.qcap;
    double rmin = 1e300, rmax = -1.0;
    long long imin = -1, jmin = -1, imax = -1, jmax = -1;
    for (long t = wid * a.world + a.rank; t < nIw * nJ; t += nwaves * a.world) {
        const long bi = t / nJ, bj = t - bi * nJ;
        const long i0 = bi * VG_IW, j0 = bj * VG_JCHUNK;
        if (a.same && j0 + VG_JCHUNK - 1 <= i0) continue;   // chunk entirely at or below the diagonal
        {
            double dlo, qlo, qhi;
            tile_q_range(vg_const(a.ib), nIw, bi, vg_const(a.jb), nJ, bj, &dlo, &qlo, &qhi);
            if (dlo > a.cmax) continue;   // no pair of this tile within max_dist
            qhi = fmin(qhi, qcap);
            const double bmax = __longlong_as_double((long long)__atomic_load_n(&best[0], __ATOMIC_RELAXED));
            const double bmin = __longlong_as_double((long long)__atomic_load_n(&best[1], __ATOMIC_RELAXED));
            if (qhi < bmax && qlo > bmin) continue;
        }
        const long i = i0 + lane;
        const bool live = i < ni;
        const long ic = live ? i : ni - 1;
        const double ax = a.iu0[ic], ay = a.iu1[ic], az = a.iu2[ic];
        bool touched = false;
        for (int sc = 0; sc < VG_JCHUNK / VG_JSUB; ++sc) {
            const long js = j0 + (long)sc * VG_JSUB;
            if (js >= nj) break;
            if (a.same && js + VG_JSUB - 1 <= i0) continue;
            {
                double dlo, qlo, qhi;
                tile_q_range(vg_const(a.ib), nIw, bi, vg_const(a.jsb), nJs, js / VG_JSUB, &dlo, &qlo, &qhi);
                if (dlo > a.cmax) continue;
                qhi = fmin(qhi, qcap);
                const double bmax = __longlong_as_double((long long)__atomic_load_n(&best[0], __ATOMIC_RELAXED));
                const double bmin = __longlong_as_double((long long)__atomic_load_n(&best[1], __ATOMIC_RELAXED));
                if (qhi < bmax && qlo > bmin) continue;
            }
            touched = true;
            const long jlen = (nj - js < VG_JSUB) ? (nj - js) : VG_JSUB;
            auto one = [&](long j, double bx, double by, double bz) __attribute__((always_inline)) {
                const double r = pair_q(ax, ay, az, bx, by, bz);
                if (live && (!a.same || j > i) && r <= qcap) {
                    if (r >= qwin_lo) {
                        const unsigned at = atomicAdd(count, 1u);
                        if (at < cap) list[at] = CkVarioPair{(int)i, (int)j, 0, 0};
                    }
                    if (r > rmax) {
                        rmax = r;
                        imax = i;
                        jmax = j;
                    }
                    if (r > 0.0 && r < rmin) {
                        rmin = r;
                        imin = i;
                        jmin = j;
                    }
                }
            };
            long k = 0;
            for (; k + 4 <= jlen; k += 4) {   // four "j" points per round: one s_load_dwordx8 per coordinate array
                const long j = js + k;        // uniform: scalar loads
                double bx[4], by[4], bz[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    bx[u] = ju0[j + u];
                    by[u] = ju1[j + u];
                    bz[u] = ju2[j + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) one(j + u, bx[u], by[u], bz[u]);
            }
            for (; k < jlen; ++k) one(js + k, ju0[js + k], ju1[js + k], ju2[js + k]);
        }
        if (touched) {   // publish this wave's extremes so far: hints for every wave's tests above
            double wmax = rmax, wmin = rmin;
            for (int off = 32; off > 0; off >>= 1) {
                wmax = fmax(wmax, __shfl_xor(wmax, off));
                wmin = fmin(wmin, __shfl_xor(wmin, off));
            }
            if (lane == 0) {
                // only when it improves the published value: 5 120 waves hammering two addresses with an atomic per
                // tile serialise at the L2 (the pass took 85 ms for a quarter of the binning pass's pairs)
                const double bmax = __longlong_as_double((long long)__atomic_load_n(&best[0], __ATOMIC_RELAXED));
                const double bmin = __longlong_as_double((long long)__atomic_load_n(&best[1], __ATOMIC_RELAXED));
                if (wmax > bmax) atomicMax(&best[0], (unsigned long long)__double_as_longlong(wmax));
                if (wmin < bmin) atomicMin(&best[1], (unsigned long long)__double_as_longlong(wmin));
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

// ---- pass 1b: every pair within the band of the two extremes -> list (the host decides them) ---------
// top candidates: qtop_lo <= q <= qcap; bottom candidates: 0 < q <= qbot_hi
__global__ __launch_bounds__(VG_TPB) void k_vario_collect(const VarioExtArgs a, double qtop_lo, double qbot_hi,
                                                           CkVarioPair* __restrict__ list, unsigned* __restrict__ count,
                                                           unsigned cap) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const vg_cptr ju0 = vg_const(a.ju0), ju1 = vg_const(a.ju1), ju2 = vg_const(a.ju2);
    const long ni = a.ni, nj = a.nj;
    const long nIw = (ni + VG_IW - 1) / VG_IW, nJ = (nj + VG_JCHUNK - 1) / VG_JCHUNK, nJs = (nj + VG_JSUB - 1) / VG_JSUB;
    const long nwaves = (long)gridDim.x * (VG_TPB / 64), wid = (long)blockIdx.x * (VG_TPB / 64) + wv;
    const double qcap = a.qcap;
    for (long t = wid * a.world + a.rank; t < nIw * nJ; t += nwaves * a.world) {
        const long bi = t / nJ, bj = t - bi * nJ;
        const long i0 = bi * VG_IW, j0 = bj * VG_JCHUNK;
        if (a.same && j0 + VG_JCHUNK - 1 <= i0) continue;
        {
            double dlo, qlo, qhi;
            tile_q_range(vg_const(a.ib), nIw, bi, vg_const(a.jb), nJ, bj, &dlo, &qlo, &qhi);
            if (!((qhi >= qtop_lo && qlo <= qcap) || qlo <= qbot_hi)) continue;
        }
        const long i = i0 + lane;
        const bool live = i < ni;
        const long ic = live ? i : ni - 1;
        const double ax = a.iu0[ic], ay = a.iu1[ic], az = a.iu2[ic];
        for (int sc = 0; sc < VG_JCHUNK / VG_JSUB; ++sc) {
            const long js = j0 + (long)sc * VG_JSUB;
            if (js >= nj) break;
            if (a.same && js + VG_JSUB - 1 <= i0) continue;
            {
                double dlo, qlo, qhi;
                tile_q_range(vg_const(a.ib), nIw, bi, vg_const(a.jsb), nJs, js / VG_JSUB, &dlo, &qlo, &qhi);
                if (!((qhi >= qtop_lo && qlo <= qcap) || qlo <= qbot_hi)) continue;
            }
            const long jlen = (nj - js < VG_JSUB) ? (nj - js) : VG_JSUB;
            auto one = [&](long j, double bx, double by, double bz) __attribute__((always_inline)) {
                const double q = pair_q(ax, ay, az, bx, by, bz);
                const bool hit = (q >= qtop_lo && q <= qcap) || (q > 0.0 && q <= qbot_hi);
                if (live && (!a.same || j > i) && hit) {
                    const unsigned at = atomicAdd(count, 1u);
                    if (at < cap) list[at] = CkVarioPair{(int)i, (int)j, 0, 0};
                }
            };
            long k = 0;
            for (; k + 4 <= jlen; k += 4) {
                const long j = js + k;
                double bx[4], by[4], bz[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    bx[u] = ju0[j + u];
                    by[u] = ju1[j + u];
                    bz[u] = ju2[j + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) one(j + u, bx[u], by[u], bz[u]);
            }
            for (; k < jlen; ++k) one(js + k, ju0[js + k], ju1[js + k], ju2[js + k]);
        }
    }
}

// ---- pass 2: binning ---------------------------------------------------------------------------------
// The kernel compares a monotone function x of q whose thresholds the host provides as two arrays per level:
// A[e] (x > A[e]: level passed for certain) and B[e] < A[e] (x <= B[e]: certainly not passed; in between: the pair is
// inside the level's rounding band and is decided exactly).
//   Euclidean  x = q = dx^2 + dy^2                                   (4 FP64 operations per pair);
//   haversine  x = -(u_i . u_j) = q / 2 - 1   with -u_i kept in the lane (3 operations per pair instead of the 6 of
//              |u_i - u_j|^2; the cancellation near x = -1 costs an ABSOLUTE 1e-15 of accuracy in q, which the band
//              of this pass allows for -- ck_api.hip: vario_band_bin).
struct VarioBinArgs {
    int same, nlev, rank, world;
    const double *iu0, *iu1, *iu2, *iv;
    long ni;
    const double *ju0, *ju1, *ju2, *jv;
    long nj;
    const double* xa;     // [1 .. nlev]: x > xa[e]: level e passed for certain
    const double* xb;     // [1 .. nlev]: x <= xb[e]: certainly not
    const double* dthr;   // [1 .. nlev]: the threshold as a distance (edge or cap), for the exact decision
    double cmax;          // largest chord that can reach the band of the cap
    const double *ib, *jb, *jsb;   // bounding balls: 64-point "i" blocks, 1024-point "j" chunks, 256-point sub-chunks
    double* part_sum;
    unsigned long long* part_cnt;   // per workgroup: VG_MAXBINS counts + [VG_MAXBINS] visited pairs
    CkVarioPair* list;
    unsigned* count;
    unsigned cap;
};
typedef const VarioBinArgs __attribute__((address_space(4))) * vg_args_ptr;

struct VarioPairCtx {
    double ax, ay, bx, by;   // Euclidean: the raw coordinates
    long i, j;
};

// Is the pair (within the band of level `lev`) above the threshold?  Euclidean: decided here, exactly as the
// reference would; haversine: deferred to the host, here "not above".
template <int METRIC>
__device__ __forceinline__ bool vario_near(vg_args_ptr a, const VarioPairCtx& c, int lev) {
    if (METRIC == CK_METRIC_EUCLID) return euclid_exact(c.ax, c.ay, c.bx, c.by) > vg_const(a->dthr)[lev];
    const unsigned at = atomicAdd(a->count, 1u);
    if (at < a->cap) a->list[at] = CkVarioPair{(int)c.i, (int)c.j, lev, 0};
    return false;
}

// Levels e0 + K .. e0 + NW for one pair.  Slot k accumulates the pairs of bin e0 + k, i.e. those that pass level
// e0 + k and not level e0 + k + 1: a lane only compares on its way up and accumulates once, where it stops (the
// first version accumulated CUMULATIVE sums at every level passed and took differences afterwards -- one FMA and one
// add more per level, and for smooth fields, whose near bins hold much smaller cloud values than the far ones, the
// differences lost up to six digits).
// slot K += the pair.  The empty asm with the slot number as an immediate keeps the nine update sites DISTINCT for the
// optimiser: identical, it merges them into one block that indexes the accumulators with a run-time slot number,
// and the accumulators then live in scratch memory (measured: 165 -> 399 ms).
template <int K>
__device__ __forceinline__ void vario_acc(double (&S)[VG_SLOTS], unsigned (&C)[VG_SLOTS], double m1, double m2) {
    S[K] = fma(m1, m2, S[K]);
    C[K] += 1u;
    asm volatile("" : : "n"(K));   // last in its block: code is merged from the end of the blocks backwards
}

template <int METRIC, int K, int NW>
__device__ __forceinline__ void vario_chain(vg_args_ptr a, const double (&A)[VG_SLOTS], const double (&B)[VG_SLOTS],
                                            double (&S)[VG_SLOTS], unsigned (&C)[VG_SLOTS], double x, double m1, double m2,
                                            int e0, const VarioPairCtx& c) {
    if constexpr (K <= NW) {
        if (x > A[K]) {
            vario_chain<METRIC, K + 1, NW>(a, A, B, S, C, x, m1, m2, e0, c);
        } else {
            // inside the band of level e0 + K: decided exactly (a pair that is above it cannot reach the next level).
            // Slot 0 of a follow-up window is the previous window's last level: haversine pairs were listed there.
            bool up = false;
            if (x > B[K]) up = (K == 0 && METRIC == CK_METRIC_HAVERSINE) ? false : vario_near<METRIC>(a, c, e0 + K);
            if (up)
                vario_acc<K>(S, C, m1, m2);
            else if (K > 0)   // K == 0: the pair belongs to the window below
                vario_acc<(K > 0 ? K - 1 : 0)>(S, C, m1, m2);
        }
    } else {
        vario_acc<NW>(S, C, m1, m2);   // passed every level of the window
    }
}

// BASE: level e0 is passed by every pair of the sub-chunk (no compare).
// NW: number of compared slots (window width).  CHECK: per-pair validity (ragged last block, diagonal).
template <int METRIC, int COV, int NW, bool BASE, bool CHECK>
__device__ __forceinline__ void vario_pair(vg_args_ptr a, const double (&A)[VG_SLOTS], const double (&B)[VG_SLOTS],
                                           double (&S)[VG_SLOTS], unsigned (&C)[VG_SLOTS], int e0, int same, double ax,
                                           double ay, double az, double rx, double ry, double av, long i, bool live, long j,
                                           double bx, double by, double bz, double bv) {
    double x;
    if (METRIC == CK_METRIC_HAVERSINE) {
        x = fma(az, bz, fma(ay, by, ax * bx));   // (ax, ay, az) = -u_i
    } else {
        const double dx = ax - bx, dy = ay - by;
        x = dx * dx + dy * dy;
    }
    double m1, m2;
    if (COV) {
        m1 = av;   // fields.py:382-383
        m2 = bv;
    } else {
        m1 = m2 = av - bv;   // fields.py:384-385; the factor 0.5 is applied to the bin sums
    }
    const VarioPairCtx c{rx, ry, bx, by, i, j};
    if (!CHECK || (live && (!same || j > i))) vario_chain<METRIC, BASE ? 1 : 0, NW>(a, A, B, S, C, x, m1, m2, e0, c);
}

// one 256-point sub-chunk against the wave's 64 "i" points
template <int METRIC, int COV, int NW, bool BASE, bool CHECK>
__device__ __forceinline__ void vario_subchunk(vg_args_ptr a, vg_cptr ju0, vg_cptr ju1, vg_cptr ju2, vg_cptr jv,
                                               const double (&A)[VG_SLOTS], const double (&B)[VG_SLOTS],
                                               double (&S)[VG_SLOTS], unsigned (&C)[VG_SLOTS], int e0, int same, double ax,
                                               double ay, double az, double rx, double ry, double av, long i, bool live,
                                               long js, long jlen) {
    long k = 0;
    // Four "j" points per round: their scalar loads merge (s_load_dwordx8 per array).  (Fetching the next round's points
    // a round ahead -- scalar loads return out of order, so a wave can only wait for all of them -- was measured and
    // dropped: 32 more live SGPRs doubled the SGPR spills, 163 -> 199 ms.)
    for (; k + 4 <= jlen; k += 4) {
        const long j = js + k;
        double bx[4], by[4], bz[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bx[u] = ju0[j + u];
            by[u] = ju1[j + u];
            bz[u] = METRIC == CK_METRIC_HAVERSINE ? ju2[j + u] : 0.0;
            bv[u] = jv[j + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            vario_pair<METRIC, COV, NW, BASE, CHECK>(a, A, B, S, C, e0, same, ax, ay, az, rx, ry, av, i, live, j + u, bx[u],
                                                     by[u], bz[u], bv[u]);
    }
    for (; k < jlen; ++k) {
        const long j = js + k;
        vario_pair<METRIC, COV, NW, BASE, CHECK>(a, A, B, S, C, e0, same, ax, ay, az, rx, ry, av, i, live, j, ju0[j], ju1[j],
                                                 METRIC == CK_METRIC_HAVERSINE ? ju2[j] : 0.0, jv[j]);
    }
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): the slot arrays are only ever indexed with
// compile-time constants, so that they are split into registers before any loop is unrolled (a `for` loop over
// them, even under #pragma unroll, kept them in scratch memory: by the time the loop was unrolled the optimiser had
// merged the slots' identical update code into one block with a run-time index)
template <class F, int... Is>
__device__ __forceinline__ void vg_static_for(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void vg_for(F&& f) {
    vg_static_for(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ double lane_value(double v, int src_lane) {   // wave-uniform src_lane: two v_readlane
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane),
                            __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}

// The arguments live in device memory and are read through the constant address space where they are needed
// (scalar loads): as by-value kernel arguments their thirty pointers and sizes stayed in SGPRs for the whole kernel
// and pushed the hot loop's thresholds and "j" points out into VGPR lanes (209 spilled SGPRs, 98 VGPRs).
template <int METRIC, int COV>
__global__ __launch_bounds__(VG_TPB) void k_vario_bin(const VarioBinArgs* __restrict__ args) {
    const vg_args_ptr a = (vg_args_ptr)(uintptr_t)args;
    __shared__ double hsum[VG_TPB / 64][VG_MAXBINS];
    __shared__ unsigned long long hcnt[VG_TPB / 64][VG_MAXBINS + 1];   // [VG_MAXBINS]: visited pairs
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (lane < VG_MAXBINS) hsum[wv][lane] = 0.0;
    if (lane <= VG_MAXBINS) hcnt[wv][lane] = 0ull;
    const int E = a->nlev, same = a->same;
    // lane e - 1 holds the two thresholds of level e
    const double xa_lane = lane < E ? vg_const(a->xa)[lane + 1] : INFINITY;
    const double xb_lane = lane < E ? vg_const(a->xb)[lane + 1] : INFINITY;
    const long ni = a->ni, nj = a->nj;
    const long nIw = (ni + VG_IW - 1) / VG_IW, nJ = (nj + VG_JCHUNK - 1) / VG_JCHUNK, nJs = (nj + VG_JSUB - 1) / VG_JSUB;
    const long nwaves = (long)gridDim.x * (VG_TPB / 64), wid = (long)blockIdx.x * (VG_TPB / 64) + wv;
    const int world = a->world, rank = a->rank;
    const double cmax = a->cmax;
    double S[VG_SLOTS];
    unsigned C[VG_SLOTS];
    vg_for<VG_SLOTS>([&](auto k) {
        S[k.value] = 0.0;
        C[k.value] = 0u;
    });
    int e0cur = -1;
    unsigned long long visited = 0;
    // bins e0cur .. e0cur + 7 <- the slot accumulators, summed over the wave
    // (always_inline: called from two places, and left as a call it would force the accumulators into scratch memory)
    auto flush = [&]() __attribute__((always_inline)) {
        if (e0cur >= 0) {
            vg_for<VG_SLOTS - 1>([&](auto k) {   // slot 8 belongs to the follow-up window
                const double sk = wave_sum(S[k.value]);
                const unsigned ck = wave_sum_u(C[k.value]);
                if (lane == 0 && e0cur + k.value < E) {
                    hsum[wv][e0cur + k.value] += sk;
                    hcnt[wv][e0cur + k.value] += (unsigned long long)ck;
                }
            });
            vg_for<VG_SLOTS>([&](auto k) {
                S[k.value] = 0.0;
                C[k.value] = 0u;
            });
        }
    };
    // wave tiles sharded over processes (ck_set_partition): this process takes the tiles t = rank (mod world)
    for (long t = wid * world + rank; t < nIw * nJ; t += nwaves * world) {
        const long bi = t / nJ, bj = t - bi * nJ;
        const long i0 = bi * VG_IW, j0 = bj * VG_JCHUNK;
        if (same && j0 + VG_JCHUNK - 1 <= i0) continue;   // chunk entirely at or below the diagonal
        {
            double dlo, qlo, qhi;
            tile_q_range(vg_const(a->ib), nIw, bi, vg_const(a->jb), nJ, bj, &dlo, &qlo, &qhi);
            if (dlo > cmax) continue;   // every pair beyond the cap
        }
        const long i = i0 + lane;
        const bool live = i < ni;
        const long ic = live ? i : ni - 1;
        const double rx = a->iu0[ic], ry = a->iu1[ic], rz = a->iu2[ic], av = a->iv[ic];
        // haversine: the lane keeps -u_i, so that x = (-u_i) . u_j grows with the distance
        const double ax = METRIC == CK_METRIC_HAVERSINE ? -rx : rx, ay = METRIC == CK_METRIC_HAVERSINE ? -ry : ry, az = -rz;
        const vg_cptr ju0 = vg_const(a->ju0), ju1 = vg_const(a->ju1), ju2 = vg_const(a->ju2), jv = vg_const(a->jv);
        for (int sc = 0; sc < VG_JCHUNK / VG_JSUB; ++sc) {
            const long js = j0 + (long)sc * VG_JSUB;
            if (js >= nj) break;
            if (same && js + VG_JSUB - 1 <= i0) continue;
            const long jlen = (nj - js < VG_JSUB) ? (nj - js) : VG_JSUB;
            double dlo, qlo, qhi;
            tile_q_range(vg_const(a->ib), nIw, bi, vg_const(a->jsb), nJs, js / VG_JSUB, &dlo, &qlo, &qhi);
            if (dlo > cmax) continue;
            // the sub-chunk's x range (with the absolute slack of the dot-product form)
            const double xlo = METRIC == CK_METRIC_HAVERSINE ? 0.5 * qlo - 1.0 - 2e-15 : qlo;
            const double xhi = METRIC == CK_METRIC_HAVERSINE ? 0.5 * qhi - 1.0 + 2e-15 : qhi;
            // levels 1 .. nlow: passed by every pair; levels > nhigh: out of reach even with the band
            const int nlow = __popcll(__ballot(xa_lane < xlo));
            const int nhigh = __popcll(__ballot(xb_lane < xhi));
            if (nlow >= E) continue;   // every pair beyond the cap
            const bool check = (i0 + VG_IW > ni) || (same && js < i0 + VG_IW);
            visited += (unsigned long long)jlen * VG_IW;
            for (int e0 = nlow;; e0 += VG_SLOTS - 1) {
                if (e0 != e0cur) {
                    flush();
                    e0cur = e0;
                }
                const int nw = (nhigh - e0 < VG_SLOTS - 1) ? (nhigh - e0) : (VG_SLOTS - 1);   // compared slots
                double A[VG_SLOTS], B[VG_SLOTS];
                vg_for<VG_SLOTS>([&](auto k) {
                    const int lev = e0 + k.value;   // its thresholds sit in lane lev - 1
                    if (lev >= 1 && lev <= nhigh) {
                        A[k.value] = lane_value(xa_lane, lev - 1);
                        B[k.value] = lane_value(xb_lane, lev - 1);
                    } else {
                        A[k.value] = B[k.value] = k.value > 0 ? INFINITY : -INFINITY;
                    }
                });
                const bool base = e0 == nlow;   // slot 0 = a level every pair passes (or the virtual level 0)
#define VG_RUN(NWV)                                                                                                        \
    if (check)                                                                                                             \
        vario_subchunk<METRIC, COV, NWV, true, true>(a, ju0, ju1, ju2, jv, A, B, S, C, e0, same, ax, ay, az, rx, ry, av, i, \
                                                     live, js, jlen);                                                      \
    else                                                                                                                   \
        vario_subchunk<METRIC, COV, NWV, true, false>(a, ju0, ju1, ju2, jv, A, B, S, C, e0, same, ax, ay, az, rx, ry, av,   \
                                                      i, live, js, jlen);
                if (!base) {
                    vario_subchunk<METRIC, COV, VG_SLOTS - 1, false, true>(a, ju0, ju1, ju2, jv, A, B, S, C, e0, same, ax, ay,
                                                                           az, rx, ry, av, i, live, js, jlen);
                } else {
                    switch (nw) {
                    case 0: VG_RUN(0) break;
                    case 1: VG_RUN(1) break;
                    case 2: VG_RUN(2) break;
                    case 3: VG_RUN(3) break;
                    case 4: VG_RUN(4) break;
                    default: VG_RUN(VG_SLOTS - 1) break;
                    }
                }
#undef VG_RUN
                if (nhigh < e0 + VG_SLOTS - 1 || e0 + VG_SLOTS - 1 >= E) break;   // no pair passes this window's last slot
            }
        }
    }
    flush();
    if (lane == 0) hcnt[wv][VG_MAXBINS] = visited;
    __syncthreads();
    // the workgroup's four wave histograms in a fixed order
    const int tid = threadIdx.x;
    if (tid <= VG_MAXBINS) {
        double s = 0.0;
        unsigned long long c = 0;
        for (int w = 0; w < VG_TPB / 64; ++w) {
            if (tid < VG_MAXBINS) s += hsum[w][tid];
            c += hcnt[w][tid];
        }
        if (tid < VG_MAXBINS) a->part_sum[(long)blockIdx.x * VG_MAXBINS + tid] = s;
        a->part_cnt[(long)blockIdx.x * (VG_MAXBINS + 1) + tid] = c;
    }
}

__global__ __launch_bounds__(64) void k_vario_final(const double* __restrict__ part_sum, const unsigned long long* __restrict__ part_cnt,
                                                     int nparts, int nb, double scale, double* __restrict__ sums,
                                                     long long* __restrict__ counts) {
    // one workgroup (= one wave) per bin; the partials of a bin are summed in a fixed order: lane l takes p = l, l + 64, ...
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b > VG_MAXBINS || (b >= nb && b != VG_MAXBINS)) return;
    double s = 0.0;
    unsigned long long c = 0;
    for (int p = lane; p < nparts; p += 64) {
        if (b < VG_MAXBINS) s += part_sum[(long)p * VG_MAXBINS + b];
        c += part_cnt[(long)p * (VG_MAXBINS + 1) + b];
    }
    s = wave_sum(s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0) {
        if (b < VG_MAXBINS) sums[b] = s * scale;
        counts[b] = (long long)c;   // counts[VG_MAXBINS]: pairs visited
    }
}

// ---- launch wrappers ------------------------------------------------------------------------------------
void ck_launch_vario_prep(hipStream_t s, const double* coords, int64_t n, int metric, double* u0, double* u1,
                          double* u2) {
    if (n <= 0) return;
    k_vario_prep<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(coords, n, metric, u0, u1, u2);
}

static VarioExtArgs vario_ext_args(int same, const double* iu, int64_t ni, const double* ju, int64_t nj, double qcap,
                                   double cmax, int rank, int world, const double* ib64, const double* jb1024,
                                   const double* jb256) {
    VarioExtArgs a;
    a.same = same;
    a.rank = rank;
    a.world = world;
    a.iu0 = iu;
    a.iu1 = iu + ni;
    a.iu2 = iu + 2 * ni;
    a.ni = ni;
    a.ju0 = ju;
    a.ju1 = ju + nj;
    a.ju2 = ju + 2 * nj;
    a.nj = nj;
    a.ib = ib64;
    a.jb = jb1024;
    a.jsb = jb256;
    a.qcap = qcap;
    a.cmax = cmax;
    return a;
}

void ck_launch_vario_extent(hipStream_t s, int grid, int same, const double* iu, int64_t ni, const double* ju,
                            int64_t nj, double qcap, void* part, int rank, int world, const double* ib64,
                            const double* jb1024, const double* jb256, double cmax, unsigned long long* best, double qwin_lo,
                            CkVarioPair* list, unsigned* count, unsigned cap) {
    // best: two words of device memory, initialised here to "nothing seen yet" (largest retained q = 0.0, smallest
    // positive q = the largest finite double)
    static const unsigned long long init[2] = {0ULL, 0x7fefffffffffffffULL};
    (void)hipMemcpyAsync(best, init, sizeof(init), hipMemcpyHostToDevice, s);
    k_vario_extent<<<dim3(grid), dim3(VG_TPB), 0, s>>>(
        vario_ext_args(same, iu, ni, ju, nj, qcap, cmax, rank, world, ib64, jb1024, jb256), (VarioPartialExt*)part, best, qwin_lo,
        list, count, cap);
}

void ck_launch_vario_collect(hipStream_t s, int grid, int same, const double* iu, int64_t ni, const double* ju,
                             int64_t nj, double qtop_lo, double qcap, double qbot_hi, CkVarioPair* list, unsigned* count,
                             unsigned cap, int rank, int world, const double* ib64, const double* jb1024,
                             const double* jb256) {
    k_vario_collect<<<dim3(grid), dim3(VG_TPB), 0, s>>>(
        vario_ext_args(same, iu, ni, ju, nj, qcap, 0.0, rank, world, ib64, jb1024, jb256), qtop_lo, qbot_hi, list, count, cap);
}

// bounding balls of blocks of `blk` consecutive points: 4 x ceil(n / blk) doubles
int64_t ck_vario_nblocks(int64_t n, int blk) { return (n + blk - 1) / blk; }
void ck_launch_vario_bounds(hipStream_t s, const double* u, int64_t n, int blk, double* out) {
    const int64_t nblk = ck_vario_nblocks(n, blk);
    if (nblk <= 0) return;
    k_vario_bounds<<<dim3((unsigned)nblk), dim3(VG_TPB), 0, s>>>(u, u + n, u + 2 * n, n, blk, nblk, out);
}

void ck_launch_vario_bin(hipStream_t s, int metric, int same, int covariogram, const double* iu, const double* iv,
                         int64_t ni, const double* ju, const double* jv, int64_t nj, int nlev, const double* xa,
                         const double* xb, const double* dthr, double cmax, const double* ib64, const double* jb1024,
                         const double* jb256, int grid, double* part_sum, unsigned long long* part_cnt, CkVarioPair* list,
                         unsigned* count, unsigned cap, int rank, int world, int nb, double* sums, long long* counts,
                         void* args_dev) {
    VarioBinArgs a;
    a.same = same;
    a.nlev = nlev;
    a.rank = rank;
    a.world = world;
    a.iu0 = iu;
    a.iu1 = iu + ni;
    a.iu2 = iu + 2 * ni;
    a.iv = iv;
    a.ni = ni;
    a.ju0 = ju;
    a.ju1 = ju + nj;
    a.ju2 = ju + 2 * nj;
    a.jv = jv;
    a.nj = nj;
    a.xa = xa;
    a.xb = xb;
    a.dthr = dthr;
    a.cmax = cmax;
    a.ib = ib64;
    a.jb = jb1024;
    a.jsb = jb256;
    a.part_sum = part_sum;
    a.part_cnt = part_cnt;
    a.list = list;
    a.count = count;
    a.cap = cap;
    static_assert(sizeof(VarioBinArgs) <= CK_VG_ARGS_BYTES, "argument block");
    (void)hipMemcpyAsync(args_dev, &a, sizeof(a), hipMemcpyHostToDevice, s);   // pageable source: staged before return
    const VarioBinArgs* ad = (const VarioBinArgs*)args_dev;
    const dim3 g(grid), b(VG_TPB);
    if (metric == CK_METRIC_HAVERSINE) {
        if (covariogram)
            k_vario_bin<CK_METRIC_HAVERSINE, 1><<<g, b, 0, s>>>(ad);
        else
            k_vario_bin<CK_METRIC_HAVERSINE, 0><<<g, b, 0, s>>>(ad);
    } else {
        if (covariogram)
            k_vario_bin<CK_METRIC_EUCLID, 1><<<g, b, 0, s>>>(ad);
        else
            k_vario_bin<CK_METRIC_EUCLID, 0><<<g, b, 0, s>>>(ad);
    }
    k_vario_final<<<dim3(VG_MAXBINS + 1), dim3(64), 0, s>>>(part_sum, part_cnt, grid, nb, covariogram ? 1.0 : 0.5, sums, counts);
}

// Workgroups of the three pair passes.  Wave tiles are dealt out with a fixed stride (deterministic sums), so every
// workgroup should be resident from the start: with more workgroups than fit, the late ones begin when the first
// finish and the tail of the launch runs half empty.  The binning kernel holds 5 waves per SIMD (83 VGPRs), i.e.
// 5 four-wave workgroups per CU.
int ck_vario_bin_grid(int64_t ni, int64_t nj) {
    const int64_t nIw = (ni + VG_IW - 1) / VG_IW, nJ = (nj + VG_JCHUNK - 1) / VG_JCHUNK;
    int64_t wgs = (nIw * nJ + (VG_TPB / 64) - 1) / (VG_TPB / 64);
    if (wgs < 1) wgs = 1;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    int per_cu = 5;   // measured at 1 M soundings, bin pass: 4 -> 166 ms, 5 -> 148, 6 -> 186, 8 -> 163
    if (const char* e = getenv("CK_VG_WGS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;   // for that measurement
    const int64_t cap = (int64_t)cus * per_cu;
    return (int)(wgs < cap ? wgs : cap);
}
