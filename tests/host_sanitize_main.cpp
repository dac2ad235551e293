// TEST-ONLY driver for the host-side C++ of libcokrige_hip.so (csrc/ck_host.cpp, csrc/ck_model.cpp), built by
// tests/test_host_sanitize.py with g++ -fsanitize=address,undefined and, separately, -fsanitize=thread.  CPU only.
// Exercises: the Hilbert order (both the small comparison-sort path and the threaded radix sort on 1e6 points), the
// reference-distance function, the variogram's level planning (incl. bands that overlap) and its threaded tie
// decisions, and the table plan / fit over the parameter box.  Exit code 0 = every self-check passed.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../include/cokrige.h"
#include "ck_host.h"
#include "ck_model.h"

static int fails = 0;
#define CHECK(c)                                                   \
    do {                                                           \
        if (!(c)) {                                                \
            fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
            ++fails;                                               \
        }                                                          \
    } while (0)

static double urand(unsigned long long* s) {
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(*s >> 11) * (1.0 / 9007199254740992.0);
}

static void check_perm(const std::vector<double>& xy, const std::vector<int64_t>& perm) {
    const int64_t n = (int64_t)perm.size();
    std::vector<char> seen((size_t)n, 0);
    for (int64_t k = 0; k < n; ++k) {
        CHECK(perm[(size_t)k] >= 0 && perm[(size_t)k] < n);
        if (perm[(size_t)k] >= 0 && perm[(size_t)k] < n) {
            CHECK(!seen[(size_t)perm[(size_t)k]]);
            seen[(size_t)perm[(size_t)k]] = 1;
        }
    }
    (void)xy;
}

int main() {
    unsigned long long seed = 12345;
    // ---- Hilbert order --------------------------------------------------------------------------------------
    for (int64_t n : {(int64_t)0, (int64_t)1, (int64_t)100, (int64_t)4095, (int64_t)4096, (int64_t)250000, (int64_t)1000000}) {
        std::vector<double> xy((size_t)(2 * n));
        for (auto& v : xy) v = urand(&seed) * 50.0 - 20.0;
        if (n > 10) {   // coincident sites and a NaN: the order must stay a permutation
            xy[2] = xy[0];
            xy[3] = xy[1];
            xy[20] = NAN;
        }
        std::vector<int64_t> perm((size_t)n);
        CHECK(ck_hilbert_order(xy.data(), n, perm.data()) == 0);
        check_perm(xy, perm);
        std::vector<int64_t> again((size_t)n);
        CHECK(ck_hilbert_order(xy.data(), n, again.data()) == 0);   // deterministic whatever the threads do
        CHECK(perm == again);
    }
    CHECK(ck_hilbert_order(nullptr, 5, nullptr) != 0);
    CHECK(ck_last_error()[0] != 0);
    // ---- reference distance -----------------------------------------------------------------------------------
    {
        const int64_t n = 20000;
        std::vector<double> A((size_t)(2 * n)), B((size_t)(2 * n)), out((size_t)n);
        for (int64_t k = 0; k < n; ++k) {
            A[(size_t)(2 * k)] = 25.0 + 25.0 * urand(&seed);
            A[(size_t)(2 * k + 1)] = -120.0 + 50.0 * urand(&seed);
            B[(size_t)(2 * k)] = 25.0 + 25.0 * urand(&seed);
            B[(size_t)(2 * k + 1)] = -120.0 + 50.0 * urand(&seed);
        }
        for (int metric = 0; metric < 2; ++metric) {
            CHECK(ck_ref_distance(metric, A.data(), B.data(), n, out.data()) == 0);
            for (int64_t k = 0; k < n; ++k) CHECK(out[(size_t)k] >= 0.0 && out[(size_t)k] < 2.1e4);
            CHECK(ck_ref_distance(metric, A.data(), A.data(), n, out.data()) == 0);
            for (int64_t k = 0; k < n; ++k) CHECK(out[(size_t)k] == 0.0);
        }
        CHECK(ck_ref_distance(7, A.data(), B.data(), n, out.data()) != 0);
    }
    // ---- workgroup -> tile maps (ck_tilemap.h) ------------------------------------------------------------------------
    for (int64_t nvalid : {(int64_t)1, (int64_t)129, (int64_t)513, (int64_t)4600, (int64_t)40000, (int64_t)100096}) {
        const int nK = (int)((nvalid + 511) / 512);
        for (int J0 : {0, nK / 2, nK - 1})
            for (int Jstep : {1, 3}) {
                const int nJ = (nK - 1 - J0) / Jstep + 1;
                const int64_t total = ck_debug_tile_map(nvalid, J0, Jstep, nJ, nullptr, 0);
                CHECK(total >= 1);
                std::vector<int32_t> out((size_t)(3 * total));
                CHECK(ck_debug_tile_map(nvalid, J0, Jstep, nJ, out.data(), total) == total);
                for (int64_t t = 0; t < total; ++t) {
                    CHECK(out[(size_t)(3 * t)] >= J0 && out[(size_t)(3 * t)] < nK);
                    CHECK(out[(size_t)(3 * t + 2)] >= 0 && out[(size_t)(3 * t + 2)] <= 3 && out[(size_t)(3 * t + 2)] <= out[(size_t)(3 * t + 1)]);
                }
            }
    }
    CHECK(ck_debug_tile_map(1000, 2, 1, 1, nullptr, 0) < 0);
    {
        std::vector<int32_t> counts;
        for (int y = 0; y < 5000; ++y) counts.push_back((int32_t)((5000 - y) / 37));
        const int64_t total = ck_debug_run_map(counts.data(), (int)counts.size(), nullptr, 0);
        CHECK(total > 0);
        std::vector<int32_t> out((size_t)(2 * total));
        CHECK(ck_debug_run_map(counts.data(), (int)counts.size(), out.data(), total) == total);
        int64_t real = 0, want = 0;
        for (int64_t b = 0; b < total; ++b)
            if (out[(size_t)(2 * b)] >= 0 && out[(size_t)(2 * b + 1)] < counts[(size_t)out[(size_t)(2 * b)]]) ++real;
        for (int32_t c : counts) want += c;
        CHECK(real == want);
        counts[10] = counts[9] + 1;
        CHECK(ck_debug_run_map(counts.data(), (int)counts.size(), nullptr, 0) < 0);
    }
    // ---- variogram levels, clusters, tie decisions ------------------------------------------------------------------
    for (int metric = 0; metric < 2; ++metric) {
        const int nb = 30;
        double edges[31];
        const double top = metric == 0 ? 1500.0 : 0.8;
        for (int b = 0; b <= nb; ++b) edges[b] = top * b / nb;
        CkVarioLevels lv;
        CHECK(ck_host_vario_levels(metric, top * 0.9, edges, nb, &lv) == 0);
        CHECK(lv.E >= 1 && lv.E <= nb && lv.EC == lv.E);
        // edges a few 1e-15 apart: their bands overlap and form one cluster
        double tight[4] = {0.0, 100.0, 100.0 * (1.0 + 4e-16), 100.0 * (1.0 + 9e-16)};
        CHECK(ck_host_vario_levels(metric, 1e9, tight, 3, &lv) == 0);
        CHECK(lv.EC < lv.E);
        double bad[3] = {0.0, 1e-300, 1.0};
        CHECK(ck_host_vario_levels(metric, 1.0, bad, 2, &lv) != 0);
        // threaded decisions: 300k candidate pairs over 5 000 points
        const int64_t np_ = 5000, nc = 300000;
        std::vector<double> c((size_t)(2 * np_)), v((size_t)np_);
        for (int64_t k = 0; k < np_; ++k) {
            c[(size_t)(2 * k)] = metric == 0 ? 25.0 + 25.0 * urand(&seed) : urand(&seed);
            c[(size_t)(2 * k + 1)] = metric == 0 ? -120.0 + 50.0 * urand(&seed) : urand(&seed);
            v[(size_t)k] = urand(&seed) - 0.5;
        }
        CHECK(ck_host_vario_levels(metric, top * 0.9, edges, nb, &lv) == 0);
        std::vector<CkVarioPair> cand((size_t)nc);
        for (int64_t k = 0; k < nc; ++k) {
            cand[(size_t)k].i = (int)(urand(&seed) * np_);
            cand[(size_t)k].j = (int)(urand(&seed) * np_);
            cand[(size_t)k].lev = 1 + (int)(urand(&seed) * lv.EC);
            cand[(size_t)k].pad = 0;
        }
        double lo = INFINITY, hi = -1.0;
        ck_host_vario_decide_extent(metric, c.data(), c.data(), cand.data(), nc, top * 0.9, &lo, &hi);
        CHECK(hi > 0.0 && hi <= top * 0.9 && lo > 0.0 && lo <= hi);
        double lo1 = INFINITY, hi1 = -1.0;   // the same on one thread (few candidates per call)
        for (int64_t a = 0; a < nc; a += 50000)
            ck_host_vario_decide_extent(metric, c.data(), c.data(), cand.data() + a, 50000, top * 0.9, &lo1, &hi1);
        CHECK(lo1 == lo && hi1 == hi);
        std::vector<double> sm(CK_HOST_VG_MAXBINS + 1, 0.0), sm1(CK_HOST_VG_MAXBINS + 1, 0.0);
        std::vector<long long> cn(CK_HOST_VG_MAXBINS + 1, 0), cn1(CK_HOST_VG_MAXBINS + 1, 0);
        ck_host_vario_fix(metric, c.data(), c.data(), v.data(), v.data(), cand.data(), nc, lv, 0, sm.data(), cn.data());
        for (int64_t a = 0; a < nc; a += 50000)
            ck_host_vario_fix(metric, c.data(), c.data(), v.data(), v.data(), cand.data() + a, 50000, lv, 0, sm1.data(), cn1.data());
        long long moved = 0;
        for (int b = 0; b <= CK_HOST_VG_MAXBINS; ++b) {
            CHECK(cn[(size_t)b] == cn1[(size_t)b]);
            CHECK(fabs(sm[(size_t)b] - sm1[(size_t)b]) <= 1e-9 * (1.0 + fabs(sm[(size_t)b])));
            moved += cn[(size_t)b];
        }
        CHECK(moved <= 0);   // pairs only move up, or out above the cap
    }
    // ---- model constants, table plan and fit over the parameter box ------------------------------------------------
    {
        const double nus[] = {0.2, 0.39, 0.5, 0.999, 1.5, 2.5, 3.4999, 3.5};
        const double lens[] = {50.0, 460.0, 2000.0};
        std::vector<double> q((size_t)(CK_TAB_DEG + 1) * CK_TAB_STRIDE), f((size_t)(CK_TAB_DEG + 1) * CK_TAB_STRIDE),
            coef((size_t)(CK_TAB_DEG + 1) * CK_TAB_STRIDE);
        for (double nu : nus)
            for (double ls : lens)
                for (int metric = 0; metric < 2; ++metric) {
                    CkMatern m;
                    ck_matern_prepare(nu, metric == 0 ? ls : ls / 4000.0, 0.9, 0.02, &m);
                    int64_t base = 0;
                    const int n_int = ck_table_plan(&m, metric, 2.0, &base, q.data());
                    CHECK(n_int >= 0 && n_int <= CK_TAB_STRIDE);
                    if (n_int <= 0) continue;
                    for (int64_t k = 0; k < (int64_t)n_int * (CK_TAB_DEG + 1); ++k)
                        f[(size_t)k] = m.amp * ck_matern_rho_scaled(m, ck_s_of_q(m, metric, q[(size_t)k]));
                    ck_table_fit(f.data(), n_int, base, coef.data());
                    // the polynomial reproduces a node value of a middle interval
                    const int it = n_int / 2;
                    const double qa = ck_table_edge(base + it), qb = ck_table_edge(base + it + 1), qq = 0.5 * (qa + qb);
                    int iv = 0;
                    const double y = ck_table_y(qq, &iv, (int)base);
                    CHECK(iv == it);
                    const double got = ck_table_poly(coef.data(), iv, y);
                    const double want = m.amp * ck_matern_rho_scaled(m, ck_s_of_q(m, metric, qq));
                    CHECK(fabs(got - want) <= 1e-11 * fabs(m.amp));
                }
        CkMatern b3[3];
        const double sig[2] = {0.99, 0.81}, nu3[3] = {0.39, 0.695, 1.0}, l3[3] = {460, 460, 460}, nug[2] = {0.02, 0.025};
        ck_model_prepare(2, sig, nu3, l3, nug, -0.19, b3);
        CHECK(b3[0].amp > 0 && b3[1].amp < 0 && b3[2].nugget == 0.025);
    }
    if (fails) {
        fprintf(stderr, "%d check(s) failed\n", fails);
        return 1;
    }
    printf("host sanitize driver: all checks passed\n");
    return 0;
}
