// TEST-ONLY shim: compiles the device math header (csrc/ck_math.h) for the host
// with g++ so that the K_nu / Matern / haversine arithmetic can be checked
// against scipy and mpmath on a machine without a GPU.  Never linked into the
// product library.
#include "ck_model.h"

extern "C" {
void shim_prepare(double nu, double len_scale, double amp, double nugget, CkMatern* m) {
    ck_matern_prepare(nu, len_scale, amp, nugget, m);
}
int shim_sizeof_matern() { return (int)sizeof(CkMatern); }
void shim_consts(double nu, double* out8) {
    CkMatern m;
    ck_matern_prepare(nu, 1.0, 1.0, 0.0, &m);
    out8[0] = m.mu; out8[1] = m.gam1; out8[2] = m.gam2; out8[3] = m.gampl; out8[4] = m.gammi;
    out8[5] = m.fact; out8[6] = m.lnpref; out8[7] = (double)m.nl;
}
// rho(s) for an array of scaled lags
void shim_rho_scaled(double nu, const double* s, long n, double* out) {
    CkMatern m;
    ck_matern_prepare(nu, 1.0, 1.0, 0.0, &m);
    for (long i = 0; i < n; ++i) out[i] = ck_matern_rho_scaled(m, s[i]);
}
// covariance entries for lags h
void shim_cov(double nu, double len_scale, double amp, double nugget, int add_nugget, const double* h, long n,
              double* out) {
    CkMatern m;
    ck_matern_prepare(nu, len_scale, amp, nugget, &m);
    for (long i = 0; i < n; ++i) out[i] = ck_cov_entry(m, h[i], add_nugget);
}
// unscaled K_nu(x) (x <= 2: Temme; x > 2: CF2 * exp(-x)), iteration counts for diagnostics
void shim_kv(double nu, const double* x, long n, double* out) {
    CkMatern m;
    ck_matern_prepare(nu, 1.0, 1.0, 0.0, &m);
    for (long i = 0; i < n; ++i) {
        double k0, k1;
        double sc = 1.0;
        if (x[i] <= 2.0) ck_temme(m, x[i], &k0, &k1);
        else { ck_cf2(m, x[i], &k0, &k1); sc = exp(-x[i]); }
        double v = m.mu, xi2 = 2.0 / x[i];
        for (int j = 0; j < m.nl; ++j) { v += 1.0; double kn = v * xi2 * k1 + k0; k0 = k1; k1 = kn; }
        out[i] = k0 * sc;
    }
}
void shim_haversine(const double* a, long na, const double* b, long nb, double* out) {
    for (long i = 0; i < na; ++i) {
        double la = a[2 * i] * CK_DEG2RAD, lo = a[2 * i + 1] * CK_DEG2RAD, ca = cos(la);
        for (long j = 0; j < nb; ++j) {
            double lb = b[2 * j] * CK_DEG2RAD, lob = b[2 * j + 1] * CK_DEG2RAD;
            out[i * nb + j] = ck_haversine_km(la, lo, ca, lb, lob, cos(lb));
        }
    }
}
}

// ---- table path on the host: plan -> node values (host evaluator) -> fit -> probe error ---------
extern "C" double shim_table_max_err(double nu, double len_scale, int metric, double qbox, int* n_int_out,
                                     double* q_lo_out, double* q_hi_out) {
    CkMatern m;
    ck_matern_prepare(nu, len_scale, 1.0, 0.0, &m);
    static double q[(CK_TAB_DEG + 1) * CK_TAB_STRIDE], f[(CK_TAB_DEG + 1) * CK_TAB_STRIDE],
        coef[(CK_TAB_DEG + 1) * CK_TAB_STRIDE];
    int64_t base = 0;
    int n_int = ck_table_plan(&m, metric, qbox, &base, q);
    const int ND = CK_TAB_DEG + 1;
    for (int k = 0; k < n_int * ND; ++k) f[k] = m.amp * ck_matern_rho_scaled(m, ck_s_of_q(m, metric, q[k]));
    ck_table_fit(f, n_int, base, coef);
    *n_int_out = n_int;
    *q_lo_out = ck_table_edge(base);
    *q_hi_out = ck_table_edge(base + n_int);
    double worst = 0;   // same measure as k_table_check: |err| / (amp max(rho, 1e-6))
    for (int it = 0; it < n_int; ++it)
        for (int j = 0; j < 16; ++j) {
            const double qa = ck_table_edge(base + it), qb = ck_table_edge(base + it + 1);
            const double qq = qa + (0.015625 + 0.0625 * j) * (qb - qa);
            int iv;
            const double y = ck_table_y(qq, &iv, (int)base);
            if (iv != it) return 1e9;
            const double got = ck_table_poly(coef, iv, y);
            const double ref = ck_matern_rho_scaled(m, ck_s_of_q(m, metric, qq));
            worst = fmax(worst, fabs(got - ref) / fmax(ref, 1e-6));
        }
    return worst;
}
