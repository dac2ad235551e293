// ck_vario.hip -- empirical (cross-)semivariogram / covariogram: pairwise lag binning.
//
// Replaces MultiField._variogram_cloud + get_variogram's pd.cut/groupby
// (src/fields.py:192-232, 378-386) without the dense n_i x n_j distance and cloud matrices:
// every pair is visited in registers and accumulated into a small per-lane table in LDS.
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
// registers) x 1024 "j" points (the same for every lane: scalar loads); per 128-point sub-chunk the bounding
// balls of the two point blocks say which levels every pair passes (nlow) and which none can reach (> nhigh),
// so a pair is compared against nhigh - nlow thresholds only -- with the points in Hilbert order mostly 3 or 4 at
// 30 bins over 1500 km.  A lane keeps one sum and one count per bin of the window (slot k = bin nlow + k) in its own
// column of the wave's table in LDS: per level one subtraction whose sign bit is shifted into a word, then two LDS
// atomics without return at slot = popcount of that word (see vario_round).  The table is reduced over the wave and
// added to the wave's histogram only when nlow changes.  (Round 1 kept per-lane histograms of ALL bins in LDS -- 113 KB,
// one wave per SIMD; the window keeps the table at 6 slots, 6.7 KB per wave, five waves per SIMD.)
// Sums are deterministic: fixed tile -> wave assignment, fixed reduction orders.
#include "ck_internal.h"

#include <stdlib.h>

#include <utility>

#define VG_TPB 256
#define VG_JCHUNK CK_VG_JCHUNK   // "j" points of a pair tile
#define VG_JSUB CK_VG_JSUB   // sub-chunk: unit of the level-window decision of the binning pass
#define VG_IW 64         // "i" points of a wave tile (binning pass)
#define VG_MAXBINS CK_VG_MAXBINS
#ifndef VG_SLOTS
#define VG_SLOTS 6       // slot 0: base level (passed by every pair of the sub-chunk); slots 1..5 compared
#endif
#ifndef VG_WAVES
#define VG_WAVES 5       // waves per SIMD asked of the register allocator for k_vario_bin
#endif

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
// the double at byte offset `off` of a constant-address-space array (scalar load: base + 32-bit offset register)
__device__ __forceinline__ double vg_at(vg_cptr p, unsigned off) {
    return *(vg_cptr)((const char __attribute__((address_space(4)))*)p + off);
}

// ---- pass 1a: extreme pairs in q-space --------------------------------------------------------------
// Largest q <= qcap and smallest positive q over this process's pair tiles (qcap already carries the upper
// band of max_dist: the host decides the pairs near it).  Same tiling as the binning pass: a wave owns 64 "i"
// points x 1024 "j" points and decides per 128-point sub-chunk whether it can hold a new extreme.
struct VarioExtArgs {
    int same, rank, world;
    const double *iu0, *iu1, *iu2;
    long ni;
    const double *ju0, *ju1, *ju2;
    long nj;
    const double *ib, *jb, *jsb;   // bounding balls: 64-point "i" blocks, 1024-point "j" chunks, 128-point sub-chunks
    double qcap, cmax;
};

// Pairs with qwin_lo <= q <= qcap are appended to `list` on the way: with dense data the largest retained q lies within
// that thin window under the cap, and the host then has every candidate for the largest distance without a second pass.
//
// What the host needs of the two extremes is a bound each -- a value not above the largest retained q, one not below
// the smallest positive q -- tight enough that the band it derives from them (ck_api.hip: ck_vario_extent) holds few
// pairs; every pair inside those bands is then decided with the reference's arithmetic.  So the lanes track UPPER WORDS
// only, as unsigned integers (non-negative doubles order like their bit patterns):
//   top:    t = qcap - q; the smallest upper word of t.  A pair beyond the cap has t < 0, i.e. an upper word >= 2^31, and
//           drops out of the minimum by itself;
//   bottom: the smallest upper word of q (a zero distance, upper word 0, is dealt with on the rare path below).
// Per round of four "j" points: 6 operations per pair for q, 1 for t, two three-way minima and two plain ones for the
// eight upper words, one fold per extreme and one test -- a pair in the window, a zero distance? -- whose rare path
// lists the window's pairs and folds the bottom exactly.  No branch per pair, no index bookkeeping (the first form:
// three compares, a branch and up to six register moves per pair; ck_vario_extent as a whole 24.8 -> 15.6 ms at 1 M
// soundings).
__device__ __forceinline__ double vario_upper(unsigned hi) {   // a double >= every non-negative double with this upper word
    return hi >= 0x7fefffffu ? INFINITY : __hiloint2double((int)(hi + 1u), 0);
}
__device__ __forceinline__ unsigned wave_min_u(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, off);
        v = o < v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(VG_TPB) void k_vario_extent(const VarioExtArgs a, VarioPartialExt* __restrict__ part,
                                                          unsigned* best, double qwin_lo, CkVarioPair* __restrict__ list,
                                                          unsigned* __restrict__ count, unsigned cap) {
    // best[0] / best[1]: the smallest upper word of t / of a positive q any wave has published so far.  A (sub-)tile whose
    // bounding balls say that none of its pairs can have a smaller t or a smaller q is skipped; a stale hint only makes
    // the test more conservative.  Once the first waves have reported, what is left are the sub-tiles that straddle
    // max_dist (largest retained lag) and those whose balls touch (smallest).
    __shared__ unsigned red[2][VG_TPB / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const vg_cptr ju0 = vg_const(a.ju0), ju1 = vg_const(a.ju1), ju2 = vg_const(a.ju2);
    const long ni = a.ni, nj = a.nj;
    const long nIw = (ni + VG_IW - 1) / VG_IW, nJ = (nj + VG_JCHUNK - 1) / VG_JCHUNK, nJs = (nj + VG_JSUB - 1) / VG_JSUB;
    const long nwaves = (long)gridDim.x * (VG_TPB / 64), wid = (long)blockIdx.x * (VG_TPB / 64) + wv;
    const double qcap = a.qcap;
    const unsigned win_hi = (unsigned)__double2hiint(qcap - qwin_lo) + 1u;   // upper words of the t inside the window
    unsigned utop = 0x7fffffffu, ubot = 0xffffffffu;                        // "none yet"
    auto cannot_improve = [&](double qlo, double qhi) {
        const double ttop = vario_upper(__atomic_load_n(&best[0], __ATOMIC_RELAXED));
        const double rbot = vario_upper(__atomic_load_n(&best[1], __ATOMIC_RELAXED));
        return (qcap - fmin(qhi, qcap) > ttop) && (qlo > rbot);
    };
    for (long t = wid * a.world + a.rank; t < nIw * nJ; t += nwaves * a.world) {
        const long bi = t / nJ, bj = t - bi * nJ;
        const long i0 = bi * VG_IW, j0 = bj * VG_JCHUNK;
        if (a.same && j0 + VG_JCHUNK - 1 <= i0) continue;   // chunk entirely at or below the diagonal
        {
            double dlo, qlo, qhi;
            tile_q_range(vg_const(a.ib), nIw, bi, vg_const(a.jb), nJ, bj, &dlo, &qlo, &qhi);
            if (dlo > a.cmax) continue;   // no pair of this tile within max_dist
            if (cannot_improve(qlo, qhi)) continue;
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
                if (cannot_improve(qlo, qhi)) continue;
            }
            touched = true;
            const bool check = (i0 + VG_IW > ni) || (a.same && js < i0 + VG_IW);   // ragged last block, diagonal
            const unsigned n = (unsigned)((nj - js < VG_JSUB) ? (nj - js) : VG_JSUB);
            // one round of U "j" points
            auto round = [&](auto uc, auto cc, unsigned k) __attribute__((always_inline)) {
                constexpr int U = decltype(uc)::value;
                constexpr bool CHECK = decltype(cc)::value;
                double r[U], tt[U], bx[U], by[U], bz[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    bx[u] = ju0[js + k + u];
                    by[u] = ju1[js + k + u];
                    bz[u] = ju2[js + k + u];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    r[u] = pair_q(ax, ay, az, bx[u], by[u], bz[u]);
                    if (CHECK) r[u] = (live && (!a.same || js + k + u > i)) ? r[u] : NAN;   // upper word 0x7ff8....: in no minimum
                    tt[u] = qcap - r[u];
                }
                unsigned ut = (unsigned)__double2hiint(tt[0]), ub = (unsigned)__double2hiint(r[0]);
#pragma unroll
                for (int u = 1; u < U; ++u) {
                    const unsigned ht = (unsigned)__double2hiint(tt[u]), hr = (unsigned)__double2hiint(r[u]);
                    ut = ht < ut ? ht : ut;
                    ub = hr < ub ? hr : ub;
                }
                utop = ut < utop ? ut : utop;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64((ut <= win_hi) | (ub == 0u)) != 0ull, 0)) {
                    // rare: a pair inside the window under the cap (listed for the host), or a zero distance (which
                    // must not enter the bottom minimum)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (r[u] <= qcap && r[u] >= qwin_lo) {
                            const unsigned at = atomicAdd(count, 1u);
                            if (at < cap) list[at] = CkVarioPair{(int)i, (int)(js + k + u), 0, 0};
                        }
                        const unsigned hr = (unsigned)__double2hiint(r[u]);
                        if (r[u] > 0.0 && hr < ubot) ubot = hr;
                    }
                } else {
                    ubot = ub < ubot ? ub : ubot;
                }
            };
            unsigned k = 0;
            if (check) {
                for (; k + 4 <= n; k += 4) round(std::integral_constant<int, 4>{}, std::true_type{}, k);
                for (; k < n; ++k) round(std::integral_constant<int, 1>{}, std::true_type{}, k);
            } else {
                for (; k + 4 <= n; k += 4) round(std::integral_constant<int, 4>{}, std::false_type{}, k);
                for (; k < n; ++k) round(std::integral_constant<int, 1>{}, std::false_type{}, k);
            }
        }
        if (touched) {   // publish this wave's extremes so far: hints for every wave's tests above
            const unsigned wtop = wave_min_u(utop), wbot = wave_min_u(ubot);
            if (lane == 0) {
                // only when it improves the published value: thousands of waves hammering two addresses with an atomic
                // per tile serialise at the L2 (the pass took 85 ms for a quarter of the binning pass's pairs)
                if (wtop < __atomic_load_n(&best[0], __ATOMIC_RELAXED)) atomicMin(&best[0], wtop);
                if (wbot < __atomic_load_n(&best[1], __ATOMIC_RELAXED)) atomicMin(&best[1], wbot);
            }
        }
    }
    // workgroup reduction; the partial result carries the two bounds as doubles and "a pair exists" in the index fields
    const unsigned wtop = wave_min_u(utop), wbot = wave_min_u(ubot);
    if (lane == 0) {
        red[0][wv] = wtop;
        red[1][wv] = wbot;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned gt = red[0][0], gb = red[1][0];
        for (int w = 1; w < VG_TPB / 64; ++w) {
            gt = red[0][w] < gt ? red[0][w] : gt;
            gb = red[1][w] < gb ? red[1][w] : gb;
        }
        const bool has_top = gt < 0x7ff00000u, has_bot = gb < 0x7ff00000u;
        VarioPartialExt o;
        o.rmax = has_top ? qcap - vario_upper(gt) : -1.0;   // <= the largest retained q of this workgroup's pairs
        o.rmin = has_bot ? vario_upper(gb) : 1e300;         // >= the smallest positive one
        o.imax = o.jmax = has_top ? 0 : -1;
        o.imin = o.jmin = has_bot ? 0 : -1;
        part[blockIdx.x] = o;
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
    const double *ib, *jb, *jsb;   // bounding balls: 64-point "i" blocks, 1024-point "j" chunks, 128-point sub-chunks
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
// reference would; haversine, and levels that stand for several edges closer together than the rounding of the
// distances (dthr < 0, ck_api.hip: ck_vario_bin): deferred to the host -- listed, if `list_it` (level e0 of a
// follow-up window was listed as the previous window's last level) -- and here "not above".
template <int METRIC>
__device__ __noinline__ bool vario_near(vg_args_ptr a, const VarioPairCtx& c, int lev, bool list_it) {
    if (METRIC == CK_METRIC_EUCLID) {
        const double thr = vg_const(a->dthr)[lev];
        if (thr >= 0.0) return euclid_exact(c.ax, c.ay, c.bx, c.by) > thr;
    }
    if (list_it) {
        const unsigned at = atomicAdd(a->count, 1u);
        if (at < a->cap) a->list[at] = CkVarioPair{(int)c.i, (int)c.j, lev, 0};
    }
    return false;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): the per-level arrays (band centres, in scalar
// registers) are only ever indexed with compile-time constants
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

// ---- one round of "j" points against the wave's 64 "i" points -----------------------------------------------------
// Slot k of a window accumulates the pairs of bin e0 + k: those that pass level e0 + k and not level e0 + k + 1.  Every
// lane owns one column of a small table in LDS -- VG_SLOTS sums and counts -- and a pair costs it, per compared level:
//   t = M - x          (M: the centre of the level's rounding band; sign bit set <=> "passed")           v_add_f64
//   acc = acc << 1 | sign(t)                                                                             v_alignbit_b32
//   dm = min(dm, |t|)  on the upper words of two levels at a time                                        1/2 v_min3_f32
// then slot = popcount(acc) and two LDS atomics without return (ds_add_f64, ds_add_u32) at slot * stride + lane.  No
// compare instruction, no branch, no lane mask, no update that most lanes sit out.  A pair within the band of a level
// is rare: one test of dm per round sends the wave to the exact path, where Euclidean pairs are decided as the reference
// would, haversine pairs are "not above" and go on the list for the host, and a pair whose exact side differs from its
// sign rule is moved from one slot to the other.
// How the kernel got here (1 M soundings, 1500 km, 30 bins; vector / scalar / LDS / branch instructions per 64-pair
// step from the SQ counters; 83 % of the pairs sit in windows of three or four compared levels):
//   round 1: per-lane tables of all 36 bins in LDS, 27 KB per wave, one wave per SIMD                         428 ms
//   a branch per level and pair on the way up, slot accumulators in registers        19 / 22 / 0.5 / 8.5     147 ms
//   the same without branches (lane masks, every slot's update under its mask)       26 / 23 / 0.5 / 1       147 ms
//       -- the two tie: the SIMDs issue about one instruction per 3 cycles whatever the mix, and 2 (NW + 1) masked
//          updates per pair, each idle in most lanes, cost what the branches did
//   per-lane slot tables in LDS, slot number from v_cmp + v_addc per level             25 / 3 / 2.4 / 1      102 ms
//   sign bits instead of compares, band test on upper words (this form)                19 / 3 / 2.4 / 1       87 ms
// and now 82 % of the cycles are vector-ALU cycles (4 per instruction).  (The version before all of them accumulated
// CUMULATIVE sums per level and took differences, which lost up to six digits on smooth fields.)
#define VG_PLANE (VG_SLOTS * 64)   // doubles between a lane's sum and the cell of its count of the same slot
// slot += (p, n): the count is a 32-bit integer in an 8-byte cell VG_PLANE doubles behind the sum, so that one
// address register serves both atomics.  Measured (1 M soundings, bin pass): counts kept as doubles 91 ms (the second
// atomic as slow as the first: the kernel ran at the LDS unit's rate); 32-bit counts in a plane of their own, 4-byte
// cells, one more address computation per pair 89 ms; this form 87 ms (and again with 128-point sub-chunks: 83.1 against
// 80.9 ms).
__device__ __forceinline__ void vario_put(double* ls, unsigned slot, double p, unsigned n) {
    __hip_atomic_fetch_add(ls + slot * 64u, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add((unsigned*)(ls + slot * 64u + VG_PLANE), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// dm = min(dm, |a|, |b|) on the HIGH WORDS of doubles read as floats: the bit patterns of non-negative doubles are
// ordered like the numbers, and so are their upper halves (up to ties in the lower half); patterns that are NaNs as
// floats belong to doubles beyond 2^1017 and drop out of a minimum.  Written as asm for the |.| input modifiers
// without the canonicalisation the compiler's own float minimum would add.
__device__ __forceinline__ void vario_min3abs(unsigned& dm, unsigned a, unsigned b) {
    asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(dm) : "v"(a), "v"(b));
}
__device__ __forceinline__ void vario_min2abs(unsigned& dm, unsigned a) {
    asm("v_min_f32_e64 %0, %0, |%1|" : "+v"(dm) : "v"(a));
}

// BASE: level e0 is passed by every pair of the sub-chunk (no compare).
// NW: number of compared slots (window width).  CHECK: per-pair validity (ragged last block, diagonal).
// ls: the lane's column of the wave's table (slot stride 64 doubles; counts VG_PLANE doubles behind the sums).
// hmax_hi: upper half of a bound on every level's band half-width, rounded up.
template <int METRIC, int COV, int NW, bool BASE, bool CHECK, int U>
__device__ __forceinline__ void vario_round(vg_args_ptr a, const double (&M)[VG_SLOTS], unsigned hmax_hi, double xa_lane,
                                            double xb_lane, double* ls, int e0, int same, double ax, double ay, double az,
                                            double rx, double ry, double av, long i, bool live, long j,
                                            const double (&bx)[U], const double (&by)[U], const double (&bz)[U],
                                            const double (&bv)[U]) {
    constexpr int K0 = BASE ? 1 : 0;        // first compared level of the window
    constexpr int NL = NW - K0 + 1;         // compared levels
    double x[U], p[U];
    unsigned pop[U];
    bool ok[U];
    unsigned prev = 0u;          // a level's word waiting for its partner in the three-way minimum
    unsigned dm = 0x7f800000u;   // +inf
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (METRIC == CK_METRIC_HAVERSINE) {
            x[u] = fma(az, bz[u], fma(ay, by[u], ax * bx[u]));   // (ax, ay, az) = -u_i
        } else {
            const double dx = ax - bx[u], dy = ay - by[u];
            x[u] = fma(dx, dx, dy * dy);
        }
        if (COV) {
            p[u] = av * bv[u];   // fields.py:382-383
        } else {
            const double d = av - bv[u];   // fields.py:384-385; the factor 0.5 is applied to the bin sums
            p[u] = d * d;
        }
        ok[u] = true;
        if (CHECK) {
            ok[u] = live & (!same | (j + u > i));
            x[u] = ok[u] ? x[u] : -INFINITY;   // passes no level, sits in no band
        }
        // per level: t = M - x (sign bit set <=> the pair is above the centre of the level's band: "passed", for now),
        // the sign bits shifted into a word, |t| into the running minimum
        unsigned acc = 0u;
        vg_for<VG_SLOTS>([&](auto kc) {
            constexpr int kk = kc.value;
            if (kk >= K0 && kk <= NW) {
                const double t = M[kk] - x[u];
                const unsigned h = (unsigned)__double2hiint(t);
                acc = __builtin_amdgcn_alignbit(acc, h, 31);   // (acc << 1) | sign
                if ((u * NL + (kk - K0)) & 1)
                    vario_min3abs(dm, prev, h);
                else if (u * NL + (kk - K0) == U * NL - 1)
                    vario_min2abs(dm, h);
                else
                    prev = h;
            }
        });
        pop[u] = __builtin_popcount(acc);   // levels passed; BASE: = slot, else slot + 1 (level e0 itself is compared:
        if (!BASE) ok[u] = ok[u] & (pop[u] > 0u);   // a pair that fails it belongs to the window below)
    }
    double* const lb = BASE ? ls : ls - 64;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (BASE && !CHECK)
            vario_put(lb, pop[u], p[u], 1u);
        else if (ok[u])
            vario_put(lb, pop[u], p[u], 1u);
    }
    if (NL > 0 && __builtin_expect(__builtin_amdgcn_ballot_w64(dm <= hmax_hi) != 0ull, 0)) {
        // some lane has a pair close to a level: rare.  Inside the band (B, A] the pair is decided exactly and, where
        // that differs from the sign rule above, moved: "above" belongs in slot kk, "not above" in slot kk - 1 (in none
        // for kk = 0: the window below has it).  Haversine pairs are "not above" here and go on the list for the host
        // -- once: level e0 of a follow-up window was listed as the previous window's last level.
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const VarioPairCtx c{rx, ry, bx[u], by[u], i, j + u};
            vg_for<VG_SLOTS>([&](auto kc) {
                constexpr int kk = kc.value;
                if (kk >= K0 && kk <= NW) {
                    const double Ak = lane_value(xa_lane, e0 + kk - 1), Bk = lane_value(xb_lane, e0 + kk - 1);
                    if ((x[u] > Bk) & !(x[u] > Ak)) {
                        const bool was = x[u] > M[kk];
                        const bool up = vario_near<METRIC>(a, c, e0 + kk, kk > 0);
                        if (up != was) {
                            const double sg = up ? 1.0 : -1.0;
                            const unsigned one = up ? 1u : ~0u;
                            vario_put(ls, (unsigned)kk, sg * p[u], one);
                            if (kk > 0) vario_put(ls, (unsigned)(kk > 0 ? kk - 1 : 0), -sg * p[u], 0u - one);
                        }
                    }
                }
            });
        }
    }
}

// one 128-point sub-chunk against the wave's 64 "i" points
template <int METRIC, int COV, int NW, bool BASE, bool CHECK>
__device__ __forceinline__ void vario_subchunk(vg_args_ptr a, vg_cptr ju0, vg_cptr ju1, vg_cptr ju2, vg_cptr jv,
                                               const double (&M)[VG_SLOTS], unsigned hmax_hi, double xa_lane,
                                               double xb_lane, double* ls, int e0, int same, double ax, double ay,
                                               double az, double rx, double ry, double av, long i, bool live, long js,
                                               long jlen) {
    // Four "j" points per round: their scalar loads merge (s_load_dwordx8 per array).  One 32-bit byte offset serves
    // the four arrays (scalar loads take base + offset register; ck_vario_begin rejects more than 2^28 points): four
    // 64-bit pointer increments and a 64-bit loop compare -- on the vector unit, there is no scalar one -- per round
    // were a quarter of the loop's scalar instructions.  (Fetching the next round's points a round ahead -- scalar
    // loads return out of order, so a wave can only wait for all of them, and that wait also covers the LDS atomics
    // issued in between -- was measured twice and dropped: 163 -> 199 ms on the first version of the kernel,
    // 87 -> 144 ms on this one.)
    constexpr int UU = 4;
    const unsigned n = (unsigned)jlen;
    unsigned ob = (unsigned)js * 8u;
    unsigned k = 0;
    for (; k + UU <= n; k += UU, ob += 8u * UU) {
        double bx[UU], by[UU], bz[UU], bv[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            bx[u] = vg_at(ju0, ob + 8u * u);
            by[u] = vg_at(ju1, ob + 8u * u);
            bz[u] = METRIC == CK_METRIC_HAVERSINE ? vg_at(ju2, ob + 8u * u) : 0.0;
            bv[u] = vg_at(jv, ob + 8u * u);
        }
        vario_round<METRIC, COV, NW, BASE, CHECK, UU>(a, M, hmax_hi, xa_lane, xb_lane, ls, e0, same, ax, ay, az, rx, ry, av, i,
                                                      live, js + k, bx, by, bz, bv);
    }
    for (; k < n; ++k, ob += 8u) {
        const double bx[1] = {vg_at(ju0, ob)}, by[1] = {vg_at(ju1, ob)},
                     bz[1] = {METRIC == CK_METRIC_HAVERSINE ? vg_at(ju2, ob) : 0.0}, bv[1] = {vg_at(jv, ob)};
        vario_round<METRIC, COV, NW, BASE, CHECK, 1>(a, M, hmax_hi, xa_lane, xb_lane, ls, e0, same, ax, ay, az, rx, ry, av, i,
                                                     live, js + k, bx, by, bz, bv);
    }
}

// The arguments live in device memory and are read through the constant address space where they are needed
// (scalar loads): as by-value kernel arguments their thirty pointers and sizes stayed in SGPRs for the whole kernel
// and pushed the hot loop's thresholds and "j" points out into VGPR lanes (209 spilled SGPRs, 98 VGPRs).
// (five waves per SIMD asked for: left alone the register allocator takes 101 VGPRs, four waves, 100 ms instead of 87;
// at six -- 80 VGPRs, a few spills -- nothing is gained, the vector ALU is the limit by then)
template <int METRIC, int COV>
__global__ __launch_bounds__(VG_TPB) __attribute__((amdgpu_waves_per_eu(VG_WAVES, 8))) void k_vario_bin(
    const VarioBinArgs* __restrict__ args) {
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
    // the wave's table: per lane one column of VG_SLOTS sums and, VG_PLANE doubles behind them, the cells of as many
    // counts (only its own lane ever touches a column, so the no-return atomics are plain read-modify-writes executed by
    // the LDS unit, in the lane's program order)
    __shared__ double ltab[VG_TPB / 64][2 * VG_PLANE];
    double* const ls = &ltab[wv][lane];
    vg_for<VG_SLOTS>([&](auto k) {
        ls[k.value * 64] = 0.0;
        *(unsigned*)(ls + VG_PLANE + k.value * 64) = 0u;   // (the count's cell, written with the type it is read with)
    });
    // the centre of every level's band (lane e - 1) and a bound on all their half-widths (its upper word, rounded up)
    const double xm_lane = lane < E ? 0.5 * (xa_lane + xb_lane) : INFINITY;
    double hmax = lane < E ? 2.0 * (xa_lane - xb_lane) : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) hmax = fmax(hmax, __shfl_xor(hmax, off));
    const unsigned hmax_hi = (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(hmax)) + 1u;
    int e0cur = -1;
    unsigned long long visited = 0;
    // bins e0cur .. e0cur + VG_SLOTS - 2 <- the table, summed over the wave
    auto flush = [&]() __attribute__((always_inline)) {
        if (e0cur >= 0) {
            vg_for<VG_SLOTS - 1>([&](auto k) {   // the last slot belongs to the follow-up window
                const double sk = wave_sum(ls[k.value * 64]);
                const unsigned ck = wave_sum_u(*(const unsigned*)(ls + VG_PLANE + k.value * 64));
                if (lane == 0 && e0cur + k.value < E) {
                    hsum[wv][e0cur + k.value] += sk;
                    hcnt[wv][e0cur + k.value] += (unsigned long long)ck;
                }
            });
            vg_for<VG_SLOTS>([&](auto k) {
                ls[k.value * 64] = 0.0;
                *(unsigned*)(ls + VG_PLANE + k.value * 64) = 0u;
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
                double M[VG_SLOTS];   // per compared level: the centre of its band
                vg_for<VG_SLOTS>([&](auto k) {
                    const int lev = e0 + k.value;   // its thresholds sit in lane lev - 1
                    M[k.value] = (lev >= 1 && lev <= nhigh) ? lane_value(xm_lane, lev - 1) : INFINITY;
                });
                const bool base = e0 == nlow;   // slot 0 = a level every pair passes (or the virtual level 0)
#define VG_RUN(NWV)                                                                                                       \
    if (check)                                                                                                            \
        vario_subchunk<METRIC, COV, NWV, true, true>(a, ju0, ju1, ju2, jv, M, hmax_hi, xa_lane, xb_lane, ls, e0, same, ax, \
                                                     ay, az, rx, ry, av, i, live, js, jlen);                              \
    else                                                                                                                  \
        vario_subchunk<METRIC, COV, NWV, true, false>(a, ju0, ju1, ju2, jv, M, hmax_hi, xa_lane, xb_lane, ls, e0, same, ax, \
                                                      ay, az, rx, ry, av, i, live, js, jlen);
                if (!base) {   // a follow-up window of a sub-chunk that spans more than VG_SLOTS - 1 levels
                    if (check)
                        vario_subchunk<METRIC, COV, VG_SLOTS - 1, false, true>(a, ju0, ju1, ju2, jv, M, hmax_hi, xa_lane, xb_lane,
                                                                               ls, e0, same, ax, ay, az, rx, ry, av, i, live, js,
                                                                               jlen);
                    else
                        vario_subchunk<METRIC, COV, VG_SLOTS - 1, false, false>(a, ju0, ju1, ju2, jv, M, hmax_hi, xa_lane, xb_lane,
                                                                                ls, e0, same, ax, ay, az, rx, ry, av, i, live, js,
                                                                                jlen);
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
                                   const double* jbsub) {
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
    a.jsb = jbsub;
    a.qcap = qcap;
    a.cmax = cmax;
    return a;
}

void ck_launch_vario_extent(hipStream_t s, int grid, int same, const double* iu, int64_t ni, const double* ju,
                            int64_t nj, double qcap, void* part, int rank, int world, const double* ib64,
                            const double* jb1024, const double* jbsub, double cmax, unsigned long long* best, double qwin_lo,
                            CkVarioPair* list, unsigned* count, unsigned cap) {
    // best: two 32-bit words of device memory (the first 8 of its 16 bytes), initialised here to "nothing seen yet"
    static const unsigned init[2] = {0x7fffffffu, 0xffffffffu};
    (void)hipMemcpyAsync(best, init, sizeof(init), hipMemcpyHostToDevice, s);
    k_vario_extent<<<dim3(grid), dim3(VG_TPB), 0, s>>>(
        vario_ext_args(same, iu, ni, ju, nj, qcap, cmax, rank, world, ib64, jb1024, jbsub), (VarioPartialExt*)part,
        (unsigned*)best, qwin_lo, list, count, cap);
}

void ck_launch_vario_collect(hipStream_t s, int grid, int same, const double* iu, int64_t ni, const double* ju,
                             int64_t nj, double qtop_lo, double qcap, double qbot_hi, CkVarioPair* list, unsigned* count,
                             unsigned cap, int rank, int world, const double* ib64, const double* jb1024,
                             const double* jbsub) {
    k_vario_collect<<<dim3(grid), dim3(VG_TPB), 0, s>>>(
        vario_ext_args(same, iu, ni, ju, nj, qcap, 0.0, rank, world, ib64, jb1024, jbsub), qtop_lo, qbot_hi, list, count, cap);
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
                         const double* jbsub, int grid, double* part_sum, unsigned long long* part_cnt, CkVarioPair* list,
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
    a.jsb = jbsub;
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

// Workgroups of the three pair passes.  Wave tiles are dealt out with a fixed stride (deterministic sums) and their
// cost varies (culled or not, window widths), so a wave should own many of them and a launch many more workgroups than
// are resident at once (5 four-wave workgroups per CU: 93 VGPRs, 27 KB of LDS): the hardware then hands the next
// workgroup to whichever CU is done, which balances the launch without making the sums depend on the timing.
int ck_vario_bin_grid(int64_t ni, int64_t nj) {
    const int64_t nIw = (ni + VG_IW - 1) / VG_IW, nJ = (nj + VG_JCHUNK - 1) / VG_JCHUNK;
    int64_t wgs = (nIw * nJ + (VG_TPB / 64) - 1) / (VG_TPB / 64);
    if (wgs < 1) wgs = 1;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    int per_cu = 80;   // measured at 1 M soundings, bin pass: 5 -> 98 ms, 20 -> 92, 40 -> 89, 80 -> 87
    if (const char* e = getenv("CK_VG_WGS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;   // for that measurement
    const int64_t cap = (int64_t)cus * per_cu;
    return (int)(wgs < cap ? wgs : cap);
}
