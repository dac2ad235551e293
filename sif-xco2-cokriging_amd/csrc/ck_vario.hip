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
// keeps CUMULATIVE sums S[k] = sum of the cloud values of its pairs that pass level nlow + k (and counts
// C[k]) in registers: nested compares, one FMA and one integer add per level passed; bin nlow + k is
// S[k] - S[k+1].  They are reduced over the wave and added to the wave's histogram in LDS only when nlow
// changes.  No LDS traffic and no atomics in the pair loop, ~60 VGPRs: eight waves per SIMD (the first version
// kept per-lane histograms in LDS -- 113 KB, one wave per SIMD, two LDS atomics per pair).
// Sums are deterministic: fixed tile -> wave assignment, fixed reduction orders.
#include "ck_internal.h"

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

// ---- pass 1a: extreme pairs in q-space --------------------------------------------------------------
// Largest q <= qcap and smallest positive q over this process's pair tiles (qcap already carries the upper
// band of max_dist: the host decides the pairs near it).
__global__ __launch_bounds__(VG_TPB) void k_vario_extent(int same, const double* __restrict__ iu0,
                                                          const double* __restrict__ iu1,
                                                          const double* __restrict__ iu2, long ni,
                                                          const double* __restrict__ ju0,
                                                          const double* __restrict__ ju1,
                                                          const double* __restrict__ ju2, long nj, double qcap,
                                                          VarioPartialExt* __restrict__ part, int rank, int world,
                                                          const double* __restrict__ ib, const double* __restrict__ jb,
                                                          double cmax, unsigned long long* best) {
    // best[0]: bit pattern of the largest retained q any workgroup has seen so far, best[1]: of the smallest
    // positive one (non-negative doubles order like their bit patterns).  A tile whose bounding balls say that all
    // its pairs lie strictly inside (qlo, qhi) with qhi < best[0] and qlo > best[1] cannot change either extreme
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
            double dlo, qlo, qhi;
            tile_q_range(ib, nI, bi, jb, nJ, bj, &dlo, &qlo, &qhi);
            if (dlo > cmax) continue;   // no pair of this tile within max_dist
            qhi = fmin(qhi, qcap);
            const double bmax = __longlong_as_double((long long)__atomic_load_n(&best[0], __ATOMIC_RELAXED));
            const double bmin = __longlong_as_double((long long)__atomic_load_n(&best[1], __ATOMIC_RELAXED));
            if (qhi < bmax && qlo > bmin) continue;   // cannot hold a new extreme (uniform: same loads for all lanes)
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
            const double r = pair_q(ax, ay, az, ju0[j0 + k], ju1[j0 + k], ju2[j0 + k]);
            if (live && k >= kbeg && r <= qcap) {
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

// ---- pass 1b: every pair within the band of the two extremes -> list (the host decides them) ---------
// top candidates: qtop_lo <= q <= qcap; bottom candidates: 0 < q <= qbot_hi
__global__ __launch_bounds__(VG_TPB) void k_vario_collect(int same, const double* __restrict__ iu0,
                                                           const double* __restrict__ iu1,
                                                           const double* __restrict__ iu2, long ni,
                                                           const double* __restrict__ ju0,
                                                           const double* __restrict__ ju1,
                                                           const double* __restrict__ ju2, long nj, double qtop_lo,
                                                           double qcap, double qbot_hi, CkVarioPair* __restrict__ list,
                                                           unsigned* __restrict__ count, unsigned cap, int rank, int world,
                                                           const double* __restrict__ ib, const double* __restrict__ jb) {
    const int tid = threadIdx.x;
    const long nI = (ni + VG_TPB - 1) / VG_TPB, nJ = (nj + VG_JCHUNK - 1) / VG_JCHUNK;
    for (long t = (long)blockIdx.x * world + rank; t < nI * nJ; t += (long)gridDim.x * world) {
        const long bi = t / nJ, bj = t - bi * nJ;
        const long i0 = bi * VG_TPB, j0 = bj * VG_JCHUNK;
        if (same && j0 + VG_JCHUNK - 1 <= i0) continue;
        if (ib) {
            double dlo, qlo, qhi;
            tile_q_range(ib, nI, bi, jb, nJ, bj, &dlo, &qlo, &qhi);
            if (!((qhi >= qtop_lo && qlo <= qcap) || qlo <= qbot_hi)) continue;
        }
        const long i = i0 + tid;
        const bool live = i < ni;
        const long ic = live ? i : ni - 1;
        const double ax = iu0[ic], ay = iu1[ic], az = iu2[ic];
        const long jend = (nj - j0 < VG_JCHUNK) ? (nj - j0) : VG_JCHUNK;
        const long kbeg = same ? (i + 1 - j0) : 0;
        long k0 = same ? (i0 + 1 - j0) : 0;
        if (k0 < 0) k0 = 0;
#pragma unroll 4
        for (long k = k0; k < jend; ++k) {
            const double q = pair_q(ax, ay, az, ju0[j0 + k], ju1[j0 + k], ju2[j0 + k]);
            const bool hit = (q >= qtop_lo && q <= qcap) || (q > 0.0 && q <= qbot_hi);
            if (live && k >= kbeg && hit) {
                const unsigned at = atomicAdd(count, 1u);
                if (at < cap) list[at] = CkVarioPair{(int)i, (int)(j0 + k), 0, 0};
            }
        }
    }
}

// ---- pass 2: binning ---------------------------------------------------------------------------------
struct VarioBinArgs {
    int same, nlev, rank, world;
    const double *iu0, *iu1, *iu2, *iv;
    long ni;
    const double *ju0, *ju1, *ju2, *jv;
    long nj;
    const double* thi;    // [1 .. nlev]: q-space threshold + band (level passed for certain if q > thi)
    const double* dthr;   // [1 .. nlev]: the threshold as a distance (edge or cap), for the exact decision
    double gam;           // q * gam > thi  <=>  q within the band or above
    double cmax;          // largest chord that can reach the band of the cap
    const double *ib, *jb, *jsb;   // bounding balls: 64-point "i" blocks, 1024-point "j" chunks, 256-point sub-chunks
    double* part_sum;
    unsigned long long* part_cnt;   // per workgroup: VG_MAXBINS counts + [VG_MAXBINS] visited pairs
    CkVarioPair* list;
    unsigned* count;
    unsigned cap;
};

struct VarioPairCtx {
    double ax, ay, bx, by;
    long i, j;
};

// Is the pair (within the band of level `lev`) above the threshold?  Euclidean: decided here, exactly as the
// reference would; haversine: deferred to the host, here "not above".
template <int METRIC>
__device__ __forceinline__ bool vario_near(const VarioBinArgs& a, const VarioPairCtx& c, int lev) {
    if (METRIC == CK_METRIC_EUCLID) return euclid_exact(c.ax, c.ay, c.bx, c.by) > a.dthr[lev];
    const unsigned at = atomicAdd(a.count, 1u);
    if (at < a.cap) a.list[at] = CkVarioPair{(int)c.i, (int)c.j, lev, 0};
    return false;
}

// levels e0 + K .. e0 + NW for one pair: cumulative accumulators of the levels it passes
template <int METRIC, int K, int NW>
__device__ __forceinline__ void vario_chain(const VarioBinArgs& a, const double (&T)[VG_SLOTS], double (&S)[VG_SLOTS],
                                            unsigned (&C)[VG_SLOTS], double q, double m1, double m2, int e0,
                                            const VarioPairCtx& c) {
    if constexpr (K <= NW) {
        if (q > T[K]) {
            S[K] = fma(m1, m2, S[K]);
            C[K] += 1u;
            vario_chain<METRIC, K + 1, NW>(a, T, S, C, q, m1, m2, e0, c);
        } else if (q * a.gam > T[K]) {
            // slot 0 of a follow-up window is the previous window's last level: the pair was listed there already
            if (!(K == 0 && METRIC == CK_METRIC_HAVERSINE) && vario_near<METRIC>(a, c, e0 + K)) {
                S[K] = fma(m1, m2, S[K]);
                C[K] += 1u;
            }
        }
    }
}

// one 256-point sub-chunk against the wave's 64 "i" points.  BASE: slot 0 is passed by every pair (no compare).
// NW: number of compared slots (window width).  CHECK: per-pair validity (ragged last block, diagonal).
// Read-only data at wave-uniform addresses ("j" points, thresholds, bounding balls) is read through the CONSTANT
// address space: hipcc then fetches it with scalar loads into SGPRs whatever stores and atomics the kernel also
// contains (with plain global pointers -- even const __restrict__ kernel parameters -- the list append's atomic in the
// same loop made it fall back to per-lane vector loads of one and the same address).  Nothing writes these arrays
// while the kernel runs.
typedef const double __attribute__((address_space(4))) * vg_cptr;
__device__ __forceinline__ vg_cptr vg_const(const double* p) { return (vg_cptr)(uintptr_t)p; }

template <int METRIC, int COV, int NW, bool BASE, bool CHECK>
__device__ __forceinline__ void vario_pair(const VarioBinArgs& a, const double (&T)[VG_SLOTS], double (&S)[VG_SLOTS],
                                           unsigned (&C)[VG_SLOTS], int e0, double ax, double ay, double az, double av,
                                           long i, bool live, long j, double bx, double by, double bz, double bv) {
    double q;
    if (METRIC == CK_METRIC_HAVERSINE)
        q = pair_q(ax, ay, az, bx, by, bz);
    else {
        const double dx = ax - bx, dy = ay - by;
        q = dx * dx + dy * dy;
    }
    double m1, m2;
    if (COV) {
        m1 = av;   // fields.py:382-383
        m2 = bv;
    } else {
        m1 = m2 = av - bv;   // fields.py:384-385; the factor 0.5 is applied to the bin sums
    }
    const VarioPairCtx c{ax, ay, bx, by, i, j};
    if (!CHECK || (live && (!a.same || j > i))) {
        if (BASE) {
            S[0] = fma(m1, m2, S[0]);
            C[0] += 1u;
            vario_chain<METRIC, 1, NW>(a, T, S, C, q, m1, m2, e0, c);
        } else {
            vario_chain<METRIC, 0, NW>(a, T, S, C, q, m1, m2, e0, c);
        }
    }
}

// one 256-point sub-chunk against the wave's 64 "i" points.  BASE: slot 0 is passed by every pair (no compare).
// NW: number of compared slots (window width).  CHECK: per-pair validity (ragged last block, diagonal).
template <int METRIC, int COV, int NW, bool BASE, bool CHECK>
__device__ __forceinline__ void vario_subchunk(const VarioBinArgs& a, vg_cptr ju0, vg_cptr ju1, vg_cptr ju2, vg_cptr jv,
                                               const double (&T)[VG_SLOTS], double (&S)[VG_SLOTS], unsigned (&C)[VG_SLOTS],
                                               int e0, double ax, double ay, double az, double av, long i, bool live,
                                               long js, long jlen) {
    long k = 0;
    for (; k + 4 <= jlen; k += 4) {   // four "j" points per round: their scalar loads merge (s_load_dwordx8 per array)
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
            vario_pair<METRIC, COV, NW, BASE, CHECK>(a, T, S, C, e0, ax, ay, az, av, i, live, j + u, bx[u], by[u], bz[u], bv[u]);
    }
    for (; k < jlen; ++k) {
        const long j = js + k;
        vario_pair<METRIC, COV, NW, BASE, CHECK>(a, T, S, C, e0, ax, ay, az, av, i, live, j, ju0[j], ju1[j],
                                                 METRIC == CK_METRIC_HAVERSINE ? ju2[j] : 0.0, jv[j]);
    }
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

template <int METRIC, int COV>
__global__ __launch_bounds__(VG_TPB) void k_vario_bin(const VarioBinArgs a) {
    const vg_cptr ju0 = vg_const(a.ju0), ju1 = vg_const(a.ju1), ju2 = vg_const(a.ju2), jv = vg_const(a.jv);
    const vg_cptr thi = vg_const(a.thi);
    __shared__ double hsum[VG_TPB / 64][VG_MAXBINS];
    __shared__ unsigned long long hcnt[VG_TPB / 64][VG_MAXBINS + 1];   // [VG_MAXBINS]: visited pairs
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (lane < VG_MAXBINS) hsum[wv][lane] = 0.0;
    if (lane <= VG_MAXBINS) hcnt[wv][lane] = 0ull;
    const int E = a.nlev;
    const double thi_lane = lane < E ? thi[lane + 1] : INFINITY;   // lane e - 1 holds the threshold of level e
    const int thi_lo = __double2loint(thi_lane), thi_hi = __double2hiint(thi_lane);
    const long ni = a.ni, nj = a.nj;
    const long nIw = (ni + VG_IW - 1) / VG_IW, nJ = (nj + VG_JCHUNK - 1) / VG_JCHUNK, nJs = (nj + VG_JSUB - 1) / VG_JSUB;
    const long nwaves = (long)gridDim.x * (VG_TPB / 64), wid = (long)blockIdx.x * (VG_TPB / 64) + wv;
    double S[VG_SLOTS];
    unsigned C[VG_SLOTS];
#pragma unroll
    for (int k = 0; k < VG_SLOTS; ++k) {
        S[k] = 0.0;
        C[k] = 0u;
    }
    int e0cur = -1;
    unsigned long long visited = 0;
    // bins e0cur .. e0cur + 7 <- differences of the cumulative accumulators, summed over the wave
    auto flush = [&]() {
        if (e0cur >= 0) {
            double s[VG_SLOTS];
            unsigned c[VG_SLOTS];
#pragma unroll
            for (int k = 0; k < VG_SLOTS; ++k) {
                s[k] = wave_sum(S[k]);
                c[k] = wave_sum_u(C[k]);
                S[k] = 0.0;
                C[k] = 0u;
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < VG_SLOTS - 1; ++k)
                    if (e0cur + k < E) {
                        hsum[wv][e0cur + k] += s[k] - s[k + 1];
                        hcnt[wv][e0cur + k] += (unsigned long long)(c[k] - c[k + 1]);
                    }
            }
        }
    };
    // wave tiles sharded over processes (ck_set_partition): this process takes the tiles t = rank (mod world)
    for (long t = wid * a.world + a.rank; t < nIw * nJ; t += nwaves * a.world) {
        const long bi = t / nJ, bj = t - bi * nJ;
        const long i0 = bi * VG_IW, j0 = bj * VG_JCHUNK;
        if (a.same && j0 + VG_JCHUNK - 1 <= i0) continue;   // chunk entirely at or below the diagonal
        {
            double dlo, qlo, qhi;
            tile_q_range(vg_const(a.ib), nIw, bi, vg_const(a.jb), nJ, bj, &dlo, &qlo, &qhi);
            if (dlo > a.cmax) continue;   // every pair beyond the cap
        }
        const long i = i0 + lane;
        const bool live = i < ni;
        const long ic = live ? i : ni - 1;
        const double ax = a.iu0[ic], ay = a.iu1[ic], az = a.iu2[ic], av = a.iv[ic];
        for (int sc = 0; sc < VG_JCHUNK / VG_JSUB; ++sc) {
            const long js = j0 + (long)sc * VG_JSUB;
            if (js >= nj) break;
            if (a.same && js + VG_JSUB - 1 <= i0) continue;
            const long jlen = (nj - js < VG_JSUB) ? (nj - js) : VG_JSUB;
            double dlo, qlo, qhi;
            tile_q_range(vg_const(a.ib), nIw, bi, vg_const(a.jsb), nJs, js / VG_JSUB, &dlo, &qlo, &qhi);
            if (dlo > a.cmax) continue;
            // levels 1 .. nlow: passed by every pair; levels > nhigh: out of reach even with the band
            const int nlow = __popcll(__ballot(thi_lane < qlo));
            const int nhigh = __popcll(__ballot(thi_lane < qhi * a.gam));
            if (nlow >= E) continue;   // every pair beyond the cap
            const bool check = (i0 + VG_IW > ni) || (a.same && js < i0 + VG_IW);
            visited += (unsigned long long)jlen * VG_IW;
            for (int e0 = nlow;; e0 += VG_SLOTS - 1) {
                if (e0 != e0cur) {
                    flush();
                    e0cur = e0;
                }
                const int nw = (nhigh - e0 < VG_SLOTS - 1) ? (nhigh - e0) : (VG_SLOTS - 1);   // compared slots
                double T[VG_SLOTS];
                T[0] = -INFINITY;
#pragma unroll
                for (int k = 0; k < VG_SLOTS; ++k) {
                    const int lev = e0 + k;   // its threshold sits in lane lev - 1
                    if (lev >= 1 && lev <= nhigh) {
                        const int sl = lev - 1;
                        T[k] = __hiloint2double(__builtin_amdgcn_readlane(thi_hi, sl), __builtin_amdgcn_readlane(thi_lo, sl));
                    } else if (k > 0) {
                        T[k] = INFINITY;
                    }
                }
                const bool base = e0 == nlow;   // slot 0 = a level every pair passes (or the virtual level 0)
#define VG_RUN(NWV)                                                                                                  \
    if (check)                                                                                                       \
        vario_subchunk<METRIC, COV, NWV, true, true>(a, ju0, ju1, ju2, jv, T, S, C, e0, ax, ay, az, av, i, live, js, jlen);             \
    else                                                                                                             \
        vario_subchunk<METRIC, COV, NWV, true, false>(a, ju0, ju1, ju2, jv, T, S, C, e0, ax, ay, az, av, i, live, js, jlen);
                if (!base) {
                    vario_subchunk<METRIC, COV, VG_SLOTS - 1, false, true>(a, ju0, ju1, ju2, jv, T, S, C, e0, ax, ay, az, av, i, live, js, jlen);
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
        if (tid < VG_MAXBINS) a.part_sum[(long)blockIdx.x * VG_MAXBINS + tid] = s;
        a.part_cnt[(long)blockIdx.x * (VG_MAXBINS + 1) + tid] = c;
    }
}

__global__ void k_vario_final(const double* __restrict__ part_sum, const unsigned long long* __restrict__ part_cnt,
                              int nparts, int nb, double scale, double* __restrict__ sums, long long* __restrict__ counts) {
    const int b = threadIdx.x;
    if (b > VG_MAXBINS || (b >= nb && b != VG_MAXBINS)) return;
    double s = 0.0;
    unsigned long long c = 0;
    for (int p = 0; p < nparts; ++p) {
        if (b < VG_MAXBINS) s += part_sum[(long)p * VG_MAXBINS + b];
        c += part_cnt[(long)p * (VG_MAXBINS + 1) + b];
    }
    if (b < VG_MAXBINS) sums[b] = s * scale;
    counts[b] = (long long)c;   // counts[VG_MAXBINS]: pairs visited
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

void ck_launch_vario_extent(hipStream_t s, int grid, int same, const double* iu, int64_t ni, const double* ju,
                            int64_t nj, double qcap, void* part, int rank, int world, const double* ib, const double* jb,
                            double cmax, unsigned long long* best) {
    // best: two words of device memory, initialised here to "nothing seen yet" (largest retained q = 0.0, smallest
    // positive q = the largest finite double)
    static const unsigned long long init[2] = {0ULL, 0x7fefffffffffffffULL};
    (void)hipMemcpyAsync(best, init, sizeof(init), hipMemcpyHostToDevice, s);
    k_vario_extent<<<dim3(grid), dim3(VG_TPB), 0, s>>>(same, iu, iu + ni, iu + 2 * ni, ni, ju, ju + nj, ju + 2 * nj, nj,
                                                       qcap, (VarioPartialExt*)part, rank, world, ib, jb, cmax, best);
}

void ck_launch_vario_collect(hipStream_t s, int grid, int same, const double* iu, int64_t ni, const double* ju,
                             int64_t nj, double qtop_lo, double qcap, double qbot_hi, CkVarioPair* list, unsigned* count,
                             unsigned cap, int rank, int world, const double* ib, const double* jb) {
    k_vario_collect<<<dim3(grid), dim3(VG_TPB), 0, s>>>(same, iu, iu + ni, iu + 2 * ni, ni, ju, ju + nj, ju + 2 * nj, nj,
                                                        qtop_lo, qcap, qbot_hi, list, count, cap, rank, world, ib, jb);
}

// bounding balls of blocks of `blk` consecutive points: 4 x ceil(n / blk) doubles
int64_t ck_vario_nblocks(int64_t n, int blk) { return (n + blk - 1) / blk; }
void ck_launch_vario_bounds(hipStream_t s, const double* u, int64_t n, int blk, double* out) {
    const int64_t nblk = ck_vario_nblocks(n, blk);
    if (nblk <= 0) return;
    k_vario_bounds<<<dim3((unsigned)nblk), dim3(VG_TPB), 0, s>>>(u, u + n, u + 2 * n, n, blk, nblk, out);
}

void ck_launch_vario_bin(hipStream_t s, int metric, int same, int covariogram, const double* iu, const double* iv,
                         int64_t ni, const double* ju, const double* jv, int64_t nj, int nlev, const double* thi,
                         const double* dthr, double gam, double cmax, const double* ib64, const double* jb1024,
                         const double* jb256, int grid, double* part_sum, unsigned long long* part_cnt, CkVarioPair* list,
                         unsigned* count, unsigned cap, int rank, int world, int nb, double* sums, long long* counts) {
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
    a.thi = thi;
    a.dthr = dthr;
    a.gam = gam;
    a.cmax = cmax;
    a.ib = ib64;
    a.jb = jb1024;
    a.jsb = jb256;
    a.part_sum = part_sum;
    a.part_cnt = part_cnt;
    a.list = list;
    a.count = count;
    a.cap = cap;
    const dim3 g(grid), b(VG_TPB);
    if (metric == CK_METRIC_HAVERSINE) {
        if (covariogram)
            k_vario_bin<CK_METRIC_HAVERSINE, 1><<<g, b, 0, s>>>(a);
        else
            k_vario_bin<CK_METRIC_HAVERSINE, 0><<<g, b, 0, s>>>(a);
    } else {
        if (covariogram)
            k_vario_bin<CK_METRIC_EUCLID, 1><<<g, b, 0, s>>>(a);
        else
            k_vario_bin<CK_METRIC_EUCLID, 0><<<g, b, 0, s>>>(a);
    }
    k_vario_final<<<dim3(1), dim3(64), 0, s>>>(part_sum, part_cnt, grid, nb, covariogram ? 1.0 : 0.5, sums, counts);
}

// workgroups of the binning pass: eight 256-thread workgroups per CU fill the chip once
int ck_vario_bin_grid(int64_t ni, int64_t nj) {
    const int64_t nIw = (ni + VG_IW - 1) / VG_IW, nJ = (nj + VG_JCHUNK - 1) / VG_JCHUNK;
    int64_t wgs = (nIw * nJ + (VG_TPB / 64) - 1) / (VG_TPB / 64);
    if (wgs < 1) wgs = 1;
    return (int)(wgs < 2048 ? wgs : 2048);
}
