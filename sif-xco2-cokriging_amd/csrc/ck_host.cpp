// ck_host.cpp -- see ck_host.h.  Nothing in this file touches the GPU.
#include "ck_host.h"
#include "ck_tilemap.h"

#include <math.h>
#include <string.h>

#include <atomic>
#include <memory>
#include <utility>

#include "../../include/cokrige.h"

#define CK_HOST_DEG2RAD 0.017453292519943295   // numpy.radians' factor pi / 180 (== CK_DEG2RAD, ck_math.h)
#define CK_HOST_EARTH_RADIUS_KM 6371.0         // == CK_EARTH_RADIUS_KM

static thread_local std::string g_err;
int ck_fail(const std::string& msg) {
    g_err = msg;
    return -1;
}
extern "C" const char* ck_last_error(void) { return g_err.c_str(); }

int ck_host_parallel_threads(int64_t n) {
    const unsigned hw = std::thread::hardware_concurrency();
    return n < 100000 ? 1 : (int)std::max(1u, std::min(8u, hw ? hw : 1u));
}

// Position along the Hilbert curve of order 16 through the unit square (x, y in [0, 65536)).  The textbook loop -- per level:
// digit (3 rx) ^ ry; if ry == 0, complement the lower bits of both coordinates when rx == 1 and exchange them -- keeps nothing but
// "exchanged" and "complemented" from level to level (the two commute), so four levels at a time come from a table indexed by
// that state and a nibble of each coordinate: 4 look-ups per point instead of 16 dependent iterations (the prediction sites of
// every ck_aux_begin are ordered on the host: 10 000 points 0.54 -> 0.1 ms).
struct HilbertTable {
    uint16_t t[4 * 256];   // [state][x nibble][y nibble] -> (8 bits of the key) << 2 | next state;  state = exchanged | complemented << 1
    HilbertTable() {
        for (unsigned st = 0; st < 4; ++st)
            for (unsigned xn = 0; xn < 16; ++xn)
                for (unsigned yn = 0; yn < 16; ++yn) {
                    unsigned sw = st & 1, cp = st >> 1, d = 0;
                    for (int b = 3; b >= 0; --b) {
                        const unsigned bx = (xn >> b) & 1, by = (yn >> b) & 1;
                        const unsigned rx = (sw ? by : bx) ^ cp, ry = (sw ? bx : by) ^ cp;
                        d = d << 2 | ((3u * rx) ^ ry);
                        if (ry == 0) {
                            if (rx == 1) cp ^= 1;
                            sw ^= 1;
                        }
                    }
                    t[st << 8 | xn << 4 | yn] = (uint16_t)(d << 2 | cp << 1 | sw);
                }
    }
};
static const HilbertTable g_hilbert;

static uint64_t hilbert_key(uint32_t x, uint32_t y) {
    uint64_t d = 0;
    unsigned st = 0;
    for (int sh = 12; sh >= 0; sh -= 4) {
        const unsigned e = g_hilbert.t[st << 8 | ((x >> sh) & 15u) << 4 | ((y >> sh) & 15u)];
        d = d << 8 | (e >> 2);
        st = e & 3u;
    }
    return d;
}

void ck_host_hilbert_order(const double* xy, int64_t n, const double lo[2], const double hi[2], std::vector<int64_t>& perm) {
    const double sx = hi[0] > lo[0] ? 65536.0 / (hi[0] - lo[0]) : 0.0, sy = hi[1] > lo[1] ? 65536.0 / (hi[1] - lo[1]) : 0.0;
    auto key_of = [&](int64_t k) -> uint32_t {   // order-16 curve: the key fits 32 bits
        double fx = (xy[2 * k] - lo[0]) * sx, fy = (xy[2 * k + 1] - lo[1]) * sy;
        fx = fx >= 0.0 ? (fx < 65535.0 ? fx : 65535.0) : 0.0;   // also catches NaN
        fy = fy >= 0.0 ? (fy < 65535.0 ? fy : 65535.0) : 0.0;
        return (uint32_t)hilbert_key((uint32_t)fx, (uint32_t)fy);
    };
    perm.resize((size_t)n);
    if (n < 4096) {
        std::vector<std::pair<uint32_t, int64_t>> key((size_t)n);
        for (int64_t k = 0; k < n; ++k) key[(size_t)k] = {key_of(k), k};
        std::stable_sort(key.begin(), key.end(),
                         [](const std::pair<uint32_t, int64_t>& a, const std::pair<uint32_t, int64_t>& b) { return a.first < b.first; });
        for (int64_t k = 0; k < n; ++k) perm[(size_t)k] = key[(size_t)k].second;
        return;
    }
    // large sets (a million soundings of a variogram): a stable LSD radix sort of (key, index) in three 11-bit passes
    // on a few threads -- every thread counts and scatters its own contiguous piece, the pieces' bucket offsets are laid
    // out thread after thread, so equal keys keep the caller's order.  One team of threads runs all the phases (a
    // spinning barrier in between; spawning a team per phase cost more than the phases), on uninitialised buffers
    // first touched by the threads that use them.  Per million points: comparison sort 270 ms, two 16-bit passes on one
    // thread 13 ms; with this sort, the bounding box and the gather on the same threads ck_vario_begin as a whole went
    // from 23 to 10.5 ms.
    const int nt = ck_host_parallel_threads(n);
    const int NBK = 2048;
    std::unique_ptr<uint32_t[]> buf(new uint32_t[(size_t)4 * (size_t)n]);
    uint32_t *ka = buf.get(), *kb = ka + n, *ia = kb + n, *ib = ia + n;
    std::vector<int64_t> hist((size_t)nt * NBK);
    std::atomic<int> arrived{0}, generation{0};
    auto barrier = [&]() {
        const int g = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == nt) {
            arrived.store(0, std::memory_order_relaxed);
            generation.fetch_add(1, std::memory_order_acq_rel);
        } else {
            while (generation.load(std::memory_order_acquire) == g) std::this_thread::yield();
        }
    };
    auto work = [&](int t) {
        const int64_t b = n * t / nt, e = n * (t + 1) / nt;
        int64_t* hh = &hist[(size_t)t * NBK];
        uint32_t *sk = ka, *si = ia, *dk = kb, *di = ib;
        for (int64_t k = b; k < e; ++k) {
            sk[k] = key_of(k);
            si[k] = (uint32_t)k;
        }
        for (int pass = 0; pass < 3; ++pass) {
            const int sh = 11 * pass;
            for (int bk = 0; bk < NBK; ++bk) hh[bk] = 0;
            for (int64_t k = b; k < e; ++k) ++hh[(sk[k] >> sh) & (NBK - 1)];
            barrier();
            if (t == 0) {
                int64_t run = 0;
                for (int bk = 0; bk < NBK; ++bk)
                    for (int q = 0; q < nt; ++q) {
                        const int64_t c = hist[(size_t)q * NBK + bk];
                        hist[(size_t)q * NBK + bk] = run;
                        run += c;
                    }
            }
            barrier();
            for (int64_t k = b; k < e; ++k) {
                const int64_t p = hh[(sk[k] >> sh) & (NBK - 1)]++;
                dk[p] = sk[k];
                di[p] = si[k];
            }
            barrier();
            std::swap(sk, dk);
            std::swap(si, di);
        }
        for (int64_t k = b; k < e; ++k) perm[(size_t)k] = (int64_t)si[k];
    };
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
}

void ck_host_bounding_box(const double* xy, int64_t n, double lo[2], double hi[2]) {
    const int nt = ck_host_parallel_threads(n);
    std::vector<double> part((size_t)nt * 4);
    ck_host_parallel(n, [&](int t, int64_t b, int64_t e) {
        double l0 = 1e300, l1 = 1e300, h0 = -1e300, h1 = -1e300;
        for (int64_t k = b; k < e; ++k) {   // comparisons, not fmin / fmax (library calls: 12 ns per point); a NaN is skipped either way
            const double x = xy[2 * k], y = xy[2 * k + 1];
            l0 = x < l0 ? x : l0;
            h0 = x > h0 ? x : h0;
            l1 = y < l1 ? y : l1;
            h1 = y > h1 ? y : h1;
        }
        part[(size_t)t * 4] = l0;
        part[(size_t)t * 4 + 1] = l1;
        part[(size_t)t * 4 + 2] = h0;
        part[(size_t)t * 4 + 3] = h1;
    });
    for (int t = 0; t < nt; ++t) {
        lo[0] = fmin(lo[0], part[(size_t)t * 4]);
        lo[1] = fmin(lo[1], part[(size_t)t * 4 + 1]);
        hi[0] = fmax(hi[0], part[(size_t)t * 4 + 2]);
        hi[1] = fmax(hi[1], part[(size_t)t * 4 + 3]);
    }
}

// sklearn's haversine_distances (Cython on libm's sin / cos / asin / sqrt) of np.radians(X) times 6371, or scipy's
// cdist.  The variogram kernels leave every pair whose bin or retention they cannot decide beyond rounding to this
// function (tests/test_ref_distance.py checks it against sklearn / scipy bit by bit on the CPU).  No contraction of
// a * b + c into an fma: the reference's binaries have none.
#if defined(__clang__)
double ck_host_ref_distance(int metric, const double* a, const double* b) {
#pragma clang fp contract(off)
#else
__attribute__((optimize("fp-contract=off"))) double ck_host_ref_distance(int metric, const double* a, const double* b) {
#endif
    if (metric == CK_HOST_METRIC_EUCLID) {
        const double d0 = a[0] - b[0], d1 = a[1] - b[1];
        const double s0 = d0 * d0;
        const double s1 = d1 * d1;
        return sqrt(s0 + s1);
    }
    const double lat1 = a[0] * CK_HOST_DEG2RAD, lon1 = a[1] * CK_HOST_DEG2RAD, lat2 = b[0] * CK_HOST_DEG2RAD,
                 lon2 = b[1] * CK_HOST_DEG2RAD;
    const double sin_0 = sin(0.5 * (lat1 - lat2));
    const double sin_1 = sin(0.5 * (lon1 - lon2));
    const double c = cos(lat1) * cos(lat2) * sin_1 * sin_1;
    const double r = sin_0 * sin_0 + c;
    const double d = 2 * asin(sqrt(r));
    return d * CK_HOST_EARTH_RADIUS_KM;
}

extern "C" int ck_hilbert_order(const double* coords, int64_t n, int64_t* perm_out) {
    if (n < 0 || (n > 0 && (!coords || !perm_out))) return ck_fail("bad arguments");
    if (n >= (1LL << 32)) return ck_fail("at most 2^32 - 1 sites");
    if (n == 0) return 0;
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
    ck_host_bounding_box(coords, n, lo, hi);
    std::vector<int64_t> perm;
    ck_host_hilbert_order(coords, n, lo, hi, perm);
    memcpy(perm_out, perm.data(), (size_t)n * sizeof(int64_t));
    return 0;
}

// host-only: the tiles of one Cholesky trailing update in launch order (ck_tilemap.h; tests/test_tilemap.py)
extern "C" int64_t ck_debug_tile_map(int64_t nvalid, int J0, int Jstep, int nJ, int32_t* out3, int64_t cap) {
    if (nvalid <= 0 || J0 < 0 || Jstep < 1 || nJ < 0) return ck_fail("bad arguments");
    if (4LL * J0 >= (nvalid + 127) / 128) return ck_fail("block column J0 lies in the padding");
    const CkTileMap m = ck_tilemap_make(nvalid, J0, Jstep, nJ);
    if (out3)
        for (int64_t t = 0; t < m.total && t < cap; ++t) {
            int u, tm, tn;
            ck_tilemap_get(m, t, u, tm, tn);
            out3[3 * t] = J0 + u * Jstep;
            out3[3 * t + 1] = tm;
            out3[3 * t + 2] = tn;
        }
    return m.total;
}

// host-only: the same for a "tall" launch (k_tall_group_d): every block column's triangle tiles are followed by the
// aux_tile_rows x 4 tiles of the right-hand-side block below it; out4 = (block column, tile row, tile column, 1 if the
// tile belongs to the right-hand-side block)
extern "C" int64_t ck_debug_tall_map(int64_t nvalid, int J0, int nJ, int aux_tile_rows, int32_t* out4, int64_t cap) {
    if (nvalid <= 0 || J0 < 0 || nJ < 0 || aux_tile_rows < 0) return ck_fail("bad arguments");
    if (4LL * J0 >= (nvalid + 127) / 128) return ck_fail("block column J0 lies in the padding");
    const CkTileMap m = ck_tilemap_make(nvalid, J0, 1, nJ, aux_tile_rows);
    if (out4)
        for (int64_t t = 0; t < m.total && t < cap; ++t) {
            int u, tm, tn;
            const bool ax = ck_tilemap_get(m, t, u, tm, tn);
            out4[4 * t] = J0 + u;
            out4[4 * t + 1] = tm;
            out4[4 * t + 2] = tn;
            out4[4 * t + 3] = ax ? 1 : 0;
        }
    return m.total;
}

// host-only: workgroup -> (system, unit) of a batched launch over systems with counts[y] units each (ck_tilemap.h)
extern "C" int64_t ck_debug_run_map(const int32_t* counts, int n_sys, int32_t* out2, int64_t cap) {
    if (n_sys < 0 || (n_sys > 0 && !counts)) return ck_fail("bad arguments");
    for (int y = 1; y < n_sys; ++y)
        if (counts[y] > counts[y - 1]) return ck_fail("counts must not increase");
    const CkRunMap m = ck_runmap_make(n_sys, [&](int y) { return (int)counts[y]; });
    const int64_t total = m.nruns ? m.off[m.nruns] : 0;
    if (out2)
        for (int64_t b = 0; b < total && b < cap; ++b) {
            int y, t;
            const bool real = ck_runmap_get(m, (int)b, y, t);
            out2[2 * b] = real ? y : -1;
            out2[2 * b + 1] = t;
        }
    return total;
}

extern "C" int ck_ref_distance(int metric, const double* A, const double* B, int64_t n, double* out) {
    if (metric != CK_HOST_METRIC_HAVERSINE && metric != CK_HOST_METRIC_EUCLID) return ck_fail("unknown metric");
    if (n > 0 && (!A || !B || !out)) return ck_fail("null array");
    for (int64_t k = 0; k < n; ++k) out[k] = ck_host_ref_distance(metric, A + 2 * k, B + 2 * k);
    return 0;
}

double ck_host_vario_q_of_dist(int metric, double d) {
    if (!(d >= 0.0)) return 0.0;
    if (metric == CK_HOST_METRIC_EUCLID) return d * d;
    const long double a = (long double)d / (2.0L * CK_HOST_EARTH_RADIUS_KM);
    if (a >= 1.57079632679489661923L) return 4.0 + 1e-9;   // beyond half the circumference: everything
    const long double sn = sinl(a);
    return (double)(4.0L * sn * sn);
}

// Haversine: the unit vectors carry ~1e-16 absolute error per component, so q = |u_i - u_j|^2 carries ~2 sqrt(q) 3e-16
// (measured 5e-16 sqrt(q) over lattice pairs from 5 km to 6 000 km), and the reference's own d a few ulp; Euclid: a few ulp.
double ck_host_vario_band(int metric, double q) {
    return metric == CK_HOST_METRIC_HAVERSINE ? 6e-15 * sqrt(q) + 8e-15 * q : 8e-15 * q;
}

double ck_host_vario_cmax(double qlim) {
    const double m = sqrt(qlim) * (1.0 + 1e-9) + 1e-12;
    return m == m ? m : INFINITY;
}

int ck_host_vario_levels(int metric, double max_dist, const double* edges, int nb, CkVarioLevels* lv) {
    // Levels 1 .. E in ascending order: the inner edges below the cap, then the cap.  A pair's bin is the number of
    // levels it passes (d > threshold: pd.cut's right-closed intervals, src/fields.py:214-216); a pair that passes
    // level E is not retained (d > max_dist, :212, or beyond the last edge, where pd.cut yields no bin).
    const double dcap = fmin(max_dist, edges[nb]);
    double xa[CK_HOST_VG_MAXBINS + 2], xb[CK_HOST_VG_MAXBINS + 2], tq[CK_HOST_VG_MAXBINS + 2];
    double* dthr = lv->dthr;
    int E = 0;
    for (int e = 1; e < nb && edges[e] < dcap; ++e) dthr[++E] = edges[e];
    dthr[++E] = dcap;
    dthr[0] = xa[0] = xb[0] = tq[0] = 0.0;
    lv->q_reach = 0.0;
    for (int e = 1; e <= E; ++e) {
        tq[e] = ck_host_vario_q_of_dist(metric, dthr[e]);
        if (!(tq[e] > 0.0)) return ck_fail("variogram bin edge too close to zero");
        // the binning kernel's monotone x (ck_vario.hip): Euclid x = q; haversine x = q / 2 - 1 from a dot product,
        // which costs an absolute 1e-15 of q near x = -1 on top of the band of the difference form
        const double bnd = ck_host_vario_band(metric, tq[e]) + (metric == CK_HOST_METRIC_HAVERSINE ? 3e-15 : 0.0);
        const double qa = tq[e] + bnd, qb = tq[e] - bnd;
        if (metric == CK_HOST_METRIC_HAVERSINE) {
            xa[e] = (double)(0.5L * (long double)qa - 1.0L);
            xb[e] = (double)(0.5L * (long double)qb - 1.0L);
        } else {
            xa[e] = qa;
            xb[e] = qb;
        }
        if (!(xb[e] < xa[e])) return ck_fail("variogram bin edge below the resolution of the distances");
        lv->q_reach = qa;
    }
    // Levels whose bands overlap (bins narrower than the rounding of the distances: max_dist equal to the smallest lattice
    // distance makes linspace(lo, hi) a few 1e-15 wide, and the reference then bins by the last bits of its distances)
    // form ONE level for the device, with the union of their bands; every pair inside it goes to the host, which walks
    // it up through the cluster's edges with the reference's distance.  Device bin k = passed k clusters = real bin
    // clast[k].  A cluster of one level is the ordinary case.
    int EC = 0;
    lv->cfirst[0] = lv->clast[0] = 0;
    lv->cxa[0] = lv->cxb[0] = lv->cthr[0] = 0.0;
    for (int e = 1; e <= E; ++e) {
        if (EC >= 1 && !(xb[e] > lv->cxa[EC])) {   // overlaps the cluster so far
            lv->clast[EC] = e;
            lv->cxa[EC] = xa[e];
            lv->cthr[EC] = -1.0;   // not a single threshold: the device lists the pair also for the Euclidean metric
        } else {
            ++EC;
            lv->cfirst[EC] = lv->clast[EC] = e;
            lv->cxa[EC] = xa[e];
            lv->cxb[EC] = xb[e];
            lv->cthr[EC] = dthr[e];
        }
    }
    lv->E = E;
    lv->EC = EC;
    return 0;
}

void ck_host_vario_decide_extent(int metric, const double* ci, const double* cj, const CkVarioPair* cand, int64_t nc,
                                 double max_dist, double* best_lo, double* best_hi) {
    double tlo[8], thi[8];
    for (int t = 0; t < 8; ++t) {
        tlo[t] = INFINITY;
        thi[t] = -1.0;
    }
    ck_host_parallel(nc, [&](int t, int64_t a, int64_t b) {
        double l = INFINITY, u = -1.0;
        for (int64_t k = a; k < b; ++k) {
            const double d = ck_host_ref_distance(metric, &ci[2 * (size_t)cand[k].i], &cj[2 * (size_t)cand[k].j]);
            if (d <= max_dist) {               // src/fields.py:212
                if (d > u) u = d;              // :395
                if (d > 0.0 && d < l) l = d;   // :394
            }
        }
        tlo[t] = l;
        thi[t] = u;
    });
    for (int t = 0; t < 8; ++t) {
        if (thi[t] > *best_hi) *best_hi = thi[t];
        if (tlo[t] < *best_lo) *best_lo = tlo[t];
    }
}

void ck_host_vario_fix(int metric, const double* ci, const double* cj, const double* vi, const double* vj,
                       const CkVarioPair* fix, int64_t nf, const CkVarioLevels& lv, int covariogram, double* sm,
                       long long* cnt) {
    if (nf <= 0) return;
    const int NB1 = CK_HOST_VG_MAXBINS + 1;
    std::vector<double> dsum((size_t)8 * NB1, 0.0);
    std::vector<long long> dcnt((size_t)8 * NB1, 0);
    ck_host_parallel(nf, [&](int t, int64_t a, int64_t b) {
        double* ds = &dsum[(size_t)t * NB1];
        long long* dc = &dcnt[(size_t)t * NB1];
        for (int64_t k = a; k < b; ++k) {
            const CkVarioPair& p = fix[(size_t)k];
            if (p.lev < 1 || p.lev > lv.EC) continue;
            const double d = ck_host_ref_distance(metric, &ci[2 * (size_t)p.i], &cj[2 * (size_t)p.j]);
            const int from = lv.clast[p.lev - 1];   // where the device put it: below the cluster
            int to = from;
            for (int e = lv.cfirst[p.lev]; e <= lv.clast[p.lev] && d > lv.dthr[e]; ++e) to = e;
            if (to == from) continue;
            const double va = vi[(size_t)p.i], vb = vj[(size_t)p.j];
            const double cl = covariogram ? va * vb : 0.5 * ((va - vb) * (va - vb));   // src/fields.py:382-385
            ds[from] -= cl;
            dc[from] -= 1;
            if (to < lv.E) {   // above the cap: not retained
                ds[to] += cl;
                dc[to] += 1;
            }
        }
    });
    for (int t = 0; t < 8; ++t)
        for (int b = 0; b <= lv.E && b < NB1; ++b) {
            sm[b] += dsum[(size_t)t * NB1 + b];
            cnt[b] += dcnt[(size_t)t * NB1 + b];
        }
}
