// ck_model.cpp -- host-side preparation of the per-block Matern constants.
//
// Everything here depends on the model parameters only (11 scalars for a
// bivariate model, src/model.py:122-130), never on the data, so it runs once
// per ck_set_model() on the host in extended precision.
#include "ck_model.h"

#include <math.h>
#include <stdint.h>
#include <string.h>

// Taylor coefficients of 1/Gamma(z) = sum_{k>=1} c_k z^k  (Abramowitz & Stegun
// 6.1.34; Wrench 1968).  Used only for |mu| < 0.02, where the closed formulas
// for Gamma_1 cancel; tests/test_host_math.py checks the result against mpmath
// over the whole interval, which also checks these digits.
static const long double RG[] = {
    0.0L,
    1.0000000000000000000000L,
    0.5772156649015328606065L,
    -0.6558780715202538810770L,
    -0.0420026350340952355290L,
    0.1665386113822914895017L,
    -0.0421977345555443367482L,
    -0.0096219715278769735621L,
    0.0072189432466630995424L,
    -0.0011651675918590651121L,
    -0.0002152416741149509728L,
    0.0001280502823881161862L,
    -0.0000201348547807882387L,
    -0.0000012504934821426707L,
    0.0000011330272319816959L,
    -0.0000002056338416977607L,
    0.0000000061160951044814L,
};

extern "C" void ck_matern_prepare(double nu, double len_scale, double amp, double nugget, CkMatern* m) {
    memset(m, 0, sizeof(*m));
    m->nu = nu;
    m->len_scale = len_scale;
    m->sqrt2nu = sqrt(2.0 * nu);
    // model.py:377-378 evaluates (1 - nu) * log(2) - gammaln(nu) in double
    m->lnpref = (1.0 - nu) * log(2.0) - lgamma(nu);
    m->amp = amp;
    m->nugget = nugget;
    m->kind = CK_KIND_GENERAL;
    if (nu == 0.5) m->kind = CK_KIND_HALF;
    if (nu == 1.5) m->kind = CK_KIND_3HALF;
    if (nu == 2.5) m->kind = CK_KIND_5HALF;
    if (nu == 3.5) m->kind = CK_KIND_7HALF;

    const int nl = (int)floor(nu + 0.5);
    const long double mu = (long double)nu - (long double)nl;
    m->nl = nl;
    m->mu = (double)mu;
    m->mu2 = (double)(mu * mu);
    m->a1 = (double)(0.25L - mu * mu);

    long double gampl, gammi, gam1, gam2;
    if (fabsl(mu) < 0.02L) {
        // 1/Gamma(1 + z) = sum_k c_k z^(k-1);  Gamma_1 = -(c2 + c4 mu^2 + ...),
        // Gamma_2 = c1 + c3 mu^2 + ...
        const long double m2 = mu * mu;
        long double g1 = 0, g2 = 0, pw = 1;
        for (int k = 1; k + 1 <= 16; k += 2) {
            g2 += RG[k] * pw;
            g1 -= RG[k + 1] * pw;
            pw *= m2;
        }
        gam1 = g1;
        gam2 = g2;
        gampl = gam2 - mu * gam1;
        gammi = gam2 + mu * gam1;
    } else {
        gampl = 1.0L / tgammal(1.0L + mu);
        gammi = 1.0L / tgammal(1.0L - mu);
        gam1 = (gammi - gampl) / (2.0L * mu);
        gam2 = 0.5L * (gammi + gampl);
    }
    m->gam1 = (double)gam1;
    m->gam2 = (double)gam2;
    m->gampl = (double)gampl;
    m->gammi = (double)gammi;
    const long double pimu = 3.14159265358979323846264338327950288L * mu;
    m->fact = (fabsl(pimu) < 1e-10L) ? 1.0 : (double)(pimu / sinl(pimu));

    for (int i = 1; i <= CK_TEMME_MAXIT; ++i) {
        const long double li = (long double)i;
        m->t_r[i] = (double)(1.0L / (li * li - mu * mu));
        m->t_p[i] = (double)(1.0L / (li - mu));
        m->t_q[i] = (double)(1.0L / (li + mu));
        m->t_i[i] = (double)(1.0L / li);
    }
    for (int i = 1; i <= CK_CF2_MAXIT; ++i) {
        const long double li = (long double)i;
        const long double a = -((0.25L - mu * mu) + li * (li - 1.0L));
        m->c_a[i] = (double)a;
        m->c_ra[i] = (a != 0.0L) ? (double)(1.0L / a) : 0.0;
        m->c_i[i] = (double)(1.0L / li);
    }
}

// Bivariate (or univariate) model -> the three (one) blocks, order 11, 12, 22.
// amp/nugget follow src/model.py:193-207:  C_ii = sigma_i^2 rho (+ nugget at 0),
// C_12 = rho12 * (sigma1 * sigma2) * rho, no nugget.
extern "C" void ck_model_prepare(int n_procs, const double* sigma, const double* nu, const double* len_scale,
                                 const double* nugget, double rho12, CkMatern* out3) {
    if (n_procs == 1) {
        ck_matern_prepare(nu[0], len_scale[0], sigma[0] * sigma[0], nugget[0], &out3[0]);
        out3[1] = out3[0];
        out3[2] = out3[0];
        return;
    }
    ck_matern_prepare(nu[0], len_scale[0], sigma[0] * sigma[0], nugget[0], &out3[0]);
    ck_matern_prepare(nu[1], len_scale[1], rho12 * (sigma[0] * sigma[1]), 0.0, &out3[1]);
    ck_matern_prepare(nu[2], len_scale[2], sigma[1] * sigma[1], nugget[1], &out3[2]);
}


// ---------------------------------------------------------------------------------------
// tabulated correlation: interval plan and per-interval Chebyshev fit (ck_math.h
// "Tabulated correlation").  Host-side, long double; the node VALUES come from the device.
// ---------------------------------------------------------------------------------------
#define CK_ND (CK_TAB_DEG + 1)

extern "C" double ck_table_edge(int64_t interval_index) {
    const uint64_t u = (uint64_t)interval_index << CK_TAB_SHIFT;
    double d;
    memcpy(&d, &u, 8);
    return d;
}

static long double cheb_node(int j) {
    const long double PI = 3.14159265358979323846264338327950288L;
    return cosl(PI * (j + 0.5L) / CK_ND);
}

// Decide the q-range of the table for one block and emit the Chebyshev nodes (n_int x CK_ND values
// of q, interval-major).  Lower end: scaled lag 1/64 (closer pairs take the exact formulas);
// upper end: q = 2 on the sphere (10 007 km) or 4x the squared bounding-box diagonal of the data.
extern "C" int ck_table_plan(const CkMatern* m, int metric, double qbox_euclid, int64_t* base_out, double* q_nodes) {
    const double d_lo = (1.0 / 64.0) * m->len_scale / m->sqrt2nu;
    double q_lo_raw, q_hi;
    if (metric == CK_METRIC_HAVERSINE) {
        const double c = 2.0 * sin(0.5 * d_lo / CK_EARTH_RADIUS_KM);
        q_lo_raw = c * c;
        q_hi = 2.0;   // a quarter of the way round; beyond it (asin's branch point sits at q = 4) -> exact
    } else {
        q_lo_raw = d_lo * d_lo;
        const double want = fmax(4.0 * qbox_euclid, 4.0 * q_lo_raw);
        q_hi = 1.0;
        while (q_hi < want) q_hi *= 2.0;
        while (q_hi * 0.5 >= want) q_hi *= 0.5;
    }
    uint64_t ul, uh;
    memcpy(&ul, &q_lo_raw, 8);
    memcpy(&uh, &q_hi, 8);
    int64_t base = (int64_t)(ul >> CK_TAB_SHIFT);
    const int64_t top = (int64_t)(uh >> CK_TAB_SHIFT);
    if (top - base > CK_TAB_MAXINT) base = top - CK_TAB_MAXINT;
    const int n_int = (int)(top - base);
    if (n_int <= 0) return 0;
    long double node[CK_ND];   // the same eight node positions in every interval (cosl once, not once per node and interval)
    for (int j = 0; j < CK_ND; ++j) node[j] = cheb_node(j);
    for (int it = 0; it < n_int; ++it) {
        const long double qa = ck_table_edge(base + it), qb = ck_table_edge(base + it + 1);
        for (int j = 0; j < CK_ND; ++j)
            q_nodes[it * CK_ND + j] = (double)(qa + (node[j] + 1.0L) * 0.5L * (qb - qa));
    }
    *base_out = base;
    return n_int;
}

// node values f (n_int x CK_ND, interval-major) -> coefficients of the polynomial in
// y = q - centre, k-major with stride CK_TAB_STRIDE (coef[k * CK_TAB_STRIDE + interval]; the
// caller provides CK_ND * CK_TAB_STRIDE doubles).  The fit is a Chebyshev interpolant in the
// normalised x = 2 y / width, converted to monomials and rescaled by (2 / width)^k -- widths are
// powers of two, so the rescaling is exact.
extern "C" void ck_table_fit(const double* f, int n_int, int64_t base, double* coef_kmajor) {
    const long double PI = 3.14159265358979323846264338327950288L;
    long double Tm[CK_ND][CK_ND];   // Tm[k][i]: coefficient of x^i in the Chebyshev polynomial T_k
    for (int k = 0; k < CK_ND; ++k)
        for (int i = 0; i < CK_ND; ++i) Tm[k][i] = 0;
    Tm[0][0] = 1;
    Tm[1][1] = 1;
    for (int k = 2; k < CK_ND; ++k)
        for (int i = 0; i < CK_ND; ++i) Tm[k][i] = (i > 0 ? 2 * Tm[k - 1][i - 1] : 0) - Tm[k - 2][i];
    // the cosines of the discrete Chebyshev transform are the same for every interval: 64 cosl calls here instead of 64 per
    // interval (3 blocks x ~700 intervals of them were 15 ms of a handle's first assemble); same values, same coefficients
    long double cs[CK_ND][CK_ND];
    for (int k = 0; k < CK_ND; ++k)
        for (int j = 0; j < CK_ND; ++j) cs[k][j] = cosl(PI * k * (j + 0.5L) / CK_ND);
    for (size_t i = 0; i < (size_t)CK_ND * CK_TAB_STRIDE; ++i) coef_kmajor[i] = 0.0;
    for (int it = 0; it < n_int; ++it) {
        long double ck[CK_ND];
        for (int k = 0; k < CK_ND; ++k) {
            long double acc = 0;
            for (int j = 0; j < CK_ND; ++j) acc += (long double)f[it * CK_ND + j] * cs[k][j];
            ck[k] = acc * (k == 0 ? 1.0L : 2.0L) / CK_ND;
        }
        const long double scale = 2.0L / ((long double)ck_table_edge(base + it + 1) - (long double)ck_table_edge(base + it));
        long double pw = 1.0L;
        for (int i = 0; i < CK_ND; ++i) {
            long double a = 0;
            for (int k = 0; k < CK_ND; ++k) a += ck[k] * Tm[k][i];
            coef_kmajor[(size_t)i * CK_TAB_STRIDE + it] = (double)(a * pw);
            pw *= scale;
        }
    }
}
