// ck_host.h -- the host-only part of libcokrige_hip.so: no HIP call, no device pointer.  Error text, the thread team,
// the Hilbert site order (stable radix sort on a few threads), the reference's own distance arithmetic on libm, and
// the variogram's level planning and tie decisions (every pair the kernels leave to the host).  Compiled into the
// product by hipcc as plain C++ and, for tests/test_host_sanitize.py, by g++ with -fsanitize=address,undefined and
// -fsanitize=thread (CPU only).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

// symbols of this file that are not part of include/cokrige.h stay inside the shared object
#define CK_HIDDEN __attribute__((visibility("hidden")))

#define CK_HOST_METRIC_HAVERSINE 0
#define CK_HOST_METRIC_EUCLID 1
#define CK_HOST_VG_MAXBINS 60   // == CK_VG_MAXBINS (ck_internal.h)

CK_HIDDEN int ck_fail(const std::string& msg);   // sets the thread-local error text, returns -1

// fn(thread, begin, end) over [0, n) on a few host threads when n is large
CK_HIDDEN int ck_host_parallel_threads(int64_t n);
template <class F>
static inline void ck_host_parallel(int64_t n, F fn) {
    const int nt = ck_host_parallel_threads(n);
    if (nt == 1) {
        fn(0, (int64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() { fn(t, n * t / nt, n * (t + 1) / nt); });
    for (auto& x : th) x.join();
}

// perm <- the indices 0..n-1 ordered along the Hilbert curve of order 16 through the box [lo, hi]^2 of the 2-column
// coordinates (stable: coincident sites keep the caller's order)
CK_HIDDEN void ck_host_hilbert_order(const double* xy, int64_t n, const double lo[2], const double hi[2],
                                     std::vector<int64_t>& perm);
// lo / hi are UPDATED (start them at +-1e300)
CK_HIDDEN void ck_host_bounding_box(const double* xy, int64_t n, double lo[2], double hi[2]);

// The reference's own distance arithmetic, on the host with libm -- bit for bit what src/fields.py:332-342 returns
CK_HIDDEN double ck_host_ref_distance(int metric, const double* a, const double* b);

// ---- variogram: thresholds and tie decisions (ck_api.hip: ck_vario_extent / ck_vario_bin) -------------------
// a pair the kernels leave to the host (indices in the order the device sees the points; lev: the level whose
// band the pair lies in, 0 for the candidates of the extent pass)
struct CkVarioPair {
    int i, j, lev, pad;
};
// distance -> the monotone q the kernels compare (ck_vario.hip): squared chord of the unit vectors | squared distance
CK_HIDDEN double ck_host_vario_q_of_dist(int metric, double d);
// rounding band of q around a threshold
CK_HIDDEN double ck_host_vario_band(int metric, double q);
// largest chord |u_i - u_j| of a pair with q <= qlim, with a safety margin
CK_HIDDEN double ck_host_vario_cmax(double qlim);

// Levels 1 .. E in ascending order: the inner edges below the cap, then the cap min(max_dist, last edge); clusters 1 .. EC
// of levels whose rounding bands overlap (one level for the device).  Index 0 of every array is a zero sentinel.
struct CkVarioLevels {
    int E, EC;
    double dthr[CK_HOST_VG_MAXBINS + 2];                                             // per level: the threshold distance
    int cfirst[CK_HOST_VG_MAXBINS + 2], clast[CK_HOST_VG_MAXBINS + 2];               // per cluster: its levels
    double cxa[CK_HOST_VG_MAXBINS + 2], cxb[CK_HOST_VG_MAXBINS + 2], cthr[CK_HOST_VG_MAXBINS + 2];   // per cluster: band, threshold
    double q_reach;                                                                  // largest q still inside the cap's band
};
// 0, or -1 with the error text set (edges too close to zero / below the resolution of the distances)
CK_HIDDEN int ck_host_vario_levels(int metric, double max_dist, const double* edges, int nb, CkVarioLevels* out);

// Extreme distances among candidate pairs, decided by the reference's arithmetic (src/fields.py:212, 394-395):
// *best_lo / *best_hi are UPDATED (start them at +inf / -1)
CK_HIDDEN void ck_host_vario_decide_extent(int metric, const double* ci, const double* cj, const CkVarioPair* cand,
                                           int64_t nc, double max_dist, double* best_lo, double* best_hi);
// The pairs inside the band of a level (cluster) were binned below it by the device; the reference's formula decides
// where they belong: sm[b] / cnt[b] (nb_total + 1 entries) are corrected in place.
CK_HIDDEN void ck_host_vario_fix(int metric, const double* ci, const double* cj, const double* vi, const double* vj,
                                 const CkVarioPair* fix, int64_t nf, const CkVarioLevels& lv, int covariogram,
                                 double* sm, long long* cnt);
