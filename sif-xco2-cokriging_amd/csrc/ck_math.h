// ck_math.h -- scalar FP64 arithmetic of the cokriging hot path.
//
// Everything the covariance-assembly kernels evaluate per matrix entry lives
// here: pairwise distance (great-circle / Euclidean) and the Matern
// correlation with a modified Bessel function K_nu of arbitrary real order.
//
// Reference semantics (91Mrwu/sif-xco2-cokriging):
//   distance      src/fields.py:318-342  (haversine * 6371 km | Euclidean cdist)
//   correlation   src/model.py:354-385   (rho = 1 at h == 0; log-domain
//                 prefactor * K_nu; non-finite -> 0; clamp >= 0)
//   covariance    src/model.py:193-207   (sigma^2 rho + nugget where h == 0;
//                 cross: rho12 sigma1 sigma2 rho, no nugget)
//
// The reference delegates K_nu to scipy.special.kv (AMOS zbesk).  Here K_nu is
// computed with Temme's series (x <= 2) and Steed's continued fraction CF2
// (x > 2) for the fractional order mu = nu - round(nu), |mu| <= 1/2, followed
// by the stable upward recurrence -- the classical scheme of Temme (1975) /
// Thompson & Barnett (1987).  Everything that depends on the order only
// (gamma-function constants, loop reciprocals) is prepared once per model on
// the host (ck_matern_prepare in ck_model.cpp) so the per-entry code has a
// single true division per CF2 step and none in the Temme loop.
//
// The functions are `CK_HD` so that tests can compile this very header with
// g++ and check it against scipy/mpmath on the CPU (tests/host_math_shim.cpp);
// the shipped library only ever instantiates them in device code.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CK_HD __host__ __device__ __forceinline__
#else
#define CK_HD inline
#endif

#define CK_EARTH_RADIUS_KM 6371.0     /* src/fields.py:17 */
#define CK_DEG2RAD (M_PI / 180.0)     /* numpy.radians: x * (pi/180) */

#define CK_METRIC_HAVERSINE 0
#define CK_METRIC_EUCLID 1

#define CK_KIND_GENERAL 0
#define CK_KIND_HALF 1       /* nu = 0.5 : exp(-s)                        */
#define CK_KIND_3HALF 2      /* nu = 1.5 : (1 + s) exp(-s)                */
#define CK_KIND_5HALF 3      /* nu = 2.5 : (1 + s + s^2/3) exp(-s)        */
#define CK_KIND_7HALF 4      /* nu = 3.5 : (1 + s + 2s^2/5 + s^3/15) e^-s */

#define CK_TEMME_MAXIT 24
#define CK_CF2_MAXIT 96

// One Matern block = one (nu, len_scale) pair with its amplitude.  A bivariate
// model has three: (1,1), (1,2), (2,2)  (src/model.py:122-130).
struct CkMatern {
    double nu, len_scale;
    double sqrt2nu;          // sqrt(2 nu)
    double lnpref;           // (1 - nu) ln 2 - lgamma(nu)            (model.py:377-378)
    double amp;              // sigma_i^2  |  rho12 * sigma1 * sigma2 (model.py:194, 203-206)
    double nugget;           // added where h == 0 (auto blocks only, model.py:195-196)
    double mu, mu2;          // nu = nl + mu, |mu| <= 1/2
    double gam1, gam2;       // Temme's Gamma_1(mu), Gamma_2(mu)
    double gampl, gammi;     // 1/Gamma(1 + mu), 1/Gamma(1 - mu)
    double fact;             // pi mu / sin(pi mu)
    double a1;               // 1/4 - mu^2  (CF2)
    int32_t nl;              // round(nu)
    int32_t kind;            // CK_KIND_*
    // order-only reciprocals, index i = 1 .. MAXIT
    double t_r[CK_TEMME_MAXIT + 1];    // 1 / (i^2 - mu^2)
    double t_p[CK_TEMME_MAXIT + 1];    // 1 / (i - mu)
    double t_q[CK_TEMME_MAXIT + 1];    // 1 / (i + mu)
    double t_i[CK_TEMME_MAXIT + 1];    // 1 / i
    double c_a[CK_CF2_MAXIT + 1];      // a_i = -(a1 + i (i - 1))
    double c_ra[CK_CF2_MAXIT + 1];     // 1 / a_i
    double c_i[CK_CF2_MAXIT + 1];      // 1 / i
};

// --------------------------------------------------------------------------
// distances
// --------------------------------------------------------------------------
// Great-circle distance in km.  Inputs are radians (lat, lon) plus cos(lat),
// all prepared per site.  Formula of sklearn's HaversineDistance, which the
// reference calls (src/fields.py:332-336):
//   r = sin^2((lat1-lat2)/2) + cos lat1 cos lat2 sin^2((lon1-lon2)/2)
//   d = 6371 * 2 asin(sqrt(r))
// Bit-identical coordinates give exactly 0.0 (the nugget rule keys on that).
CK_HD double ck_haversine_km(double lat1, double lon1, double cos1,
                             double lat2, double lon2, double cos2) {
    double s0 = sin(0.5 * (lat1 - lat2));
    double s1 = sin(0.5 * (lon1 - lon2));
    double r = s0 * s0 + cos1 * cos2 * s1 * s1;
    return 2.0 * asin(sqrt(r)) * CK_EARTH_RADIUS_KM;
}

// scipy cdist default (src/fields.py:340-342)
CK_HD double ck_euclid(double x1, double y1, double x2, double y2) {
    double d0 = x1 - x2, d1 = y1 - y2;
    return sqrt(d0 * d0 + d1 * d1);
}

// --------------------------------------------------------------------------
// K_mu(x), K_{mu+1}(x) for |mu| <= 1/2
// --------------------------------------------------------------------------
// Temme's series, 0 < x <= 2.  Returns unscaled K values.
CK_HD void ck_temme(const CkMatern& m, double x, double* kmu, double* kmu1) {
    const double x2 = 0.5 * x;
    const double d = -log(x2);
    const double e = m.mu * d;
    double E = exp(e);
    double Ei = 1.0 / E;
    double ch = 0.5 * (E + Ei);
    double shc;  // sinh(e) / e
    if (fabs(e) < 0.5) {
        const double e2 = e * e;
        shc = 1.0 + e2 * (1.0 / 6 + e2 * (1.0 / 120 + e2 * (1.0 / 5040 + e2 * (1.0 / 362880 +
              e2 * (1.0 / 39916800 + e2 * (1.0 / 6227020800.0 + e2 * (1.0 / 1307674368000.0)))))));
    } else {
        shc = 0.5 * (E - Ei) / e;
    }
    double ff = m.fact * (m.gam1 * ch + m.gam2 * shc * d);
    double sum = ff;
    double p = 0.5 * E / m.gampl;
    double q = 0.5 * Ei / m.gammi;
    double c = 1.0;
    const double dd = x2 * x2;
    double sum1 = p;
    for (int i = 1; i <= CK_TEMME_MAXIT; ++i) {
        ff = (i * ff + p + q) * m.t_r[i];
        c *= dd * m.t_i[i];
        p *= m.t_p[i];
        q *= m.t_q[i];
        const double del = c * ff;
        sum += del;
        sum1 += c * (p - i * ff);
        if (fabs(del) < fabs(sum) * 1e-17) break;
    }
    *kmu = sum;
    *kmu1 = sum1 * (2.0 / x);
}

// Steed's CF2, x > 2.  Returns K scaled by exp(x):  k = K * e^x.
CK_HD void ck_cf2(const CkMatern& m, double x, double* kmu_s, double* kmu1_s) {
    double b = 2.0 * (1.0 + x);
    double d = 1.0 / b;
    double h = d, delh = d;
    double q1 = 0.0, q2 = 1.0;
    double q = m.a1, c = m.a1;
    double s = 1.0 + q * delh;
    for (int i = 2; i <= CK_CF2_MAXIT; ++i) {
        const double a = m.c_a[i];
        c = -a * c * m.c_i[i];
        const double qnew = (q1 - b * q2) * m.c_ra[i];
        q1 = q2;
        q2 = qnew;
        q += c * qnew;
        b += 2.0;
        d = 1.0 / (b + a * d);
        delh = (b * d - 1.0) * delh;
        h += delh;
        const double dels = q * delh;
        s += dels;
        if (fabs(dels) < fabs(s) * 1e-17) break;
    }
    h = m.a1 * h;
    const double k0 = sqrt(M_PI / (2.0 * x)) / s;
    *kmu_s = k0;
    *kmu1_s = k0 * (m.mu + x + 0.5 - h) / x;
}

// --------------------------------------------------------------------------
// Matern correlation as a function of the scaled lag s = sqrt(2 nu) h / ell > 0
// (src/model.py:372-385).  h == 0 is handled by the caller.
// --------------------------------------------------------------------------
CK_HD double ck_matern_rho_scaled(const CkMatern& m, double s) {
    double rho;
    switch (m.kind) {
    case CK_KIND_HALF:
        rho = exp(-s);
        break;
    case CK_KIND_3HALF:
        rho = (1.0 + s) * exp(-s);
        break;
    case CK_KIND_5HALF:
        rho = (1.0 + s + s * s * (1.0 / 3.0)) * exp(-s);
        break;
    case CK_KIND_7HALF:
        rho = (1.0 + s + s * s * (0.4 + s * (1.0 / 15.0))) * exp(-s);
        break;
    default: {
        double k0, k1;
        const double ls = log(s);
        double ex;  // exponent of the prefactor (with -s folded in when K is scaled)
        if (s <= 2.0) {
            ck_temme(m, s, &k0, &k1);
            ex = m.lnpref + m.nu * ls;
        } else {
            ck_cf2(m, s, &k0, &k1);
            ex = m.lnpref + m.nu * ls - s;
        }
        // upward recurrence K_{v+1} = (2 v / x) K_v + K_{v-1}, v = mu + 1, ...
        const double xi2 = 2.0 / s;
        double v = m.mu;
        for (int i = 0; i < m.nl; ++i) {
            v += 1.0;
            const double kn = v * xi2 * k1 + k0;   // K_{v+1} from K_v (=k1) and K_{v-1} (=k0)
            k0 = k1;
            k1 = kn;
        }
        rho = exp(ex) * k0;
        break;
    }
    }
    // model.py:382-384: non-finite -> 0, then clamp at >= 0
    if (!(fabs(rho) <= 1.79769313486231570815e308)) rho = 0.0;
    return rho > 0.0 ? rho : 0.0;
}

// --------------------------------------------------------------------------
// Tabulated covariance (fast assembly path)
// --------------------------------------------------------------------------
// For one Matern block, C(q) = amp * rho is tabulated as a function of q, the SQUARED chord
// (haversine: q = |u_i - u_j|^2 on the unit sphere, d = 2 R asin(sqrt(q)/2); Euclid: q = d^2).
// q-space is cut at the double's exponent and top 5 mantissa bits (32 sub-intervals per octave),
// so the interval index is one shift of the bit pattern and the interval CENTRE is one
// and-or of it; on every interval C is a degree-7 polynomial in y = q - centre (exact in
// floating point).  The nearest singularity of rho(q) is q = 0, 65 half-widths away, so the
// Chebyshev interpolant converges like 130^-k: degree 7 is at the rounding floor of the
// leading term.  One entry costs an and-or, a subtract, 8 LDS reads and 7 FMA -- no exp, no
// Bessel function, no sqrt/asin/sin.  Accuracy is ABSOLUTE in units of amp (what the
// factorisation of Sigma responds to): far in the tail, where rho < 1e-6, the relative error
// of rho itself grows (1e-12 at rho = 1e-14) while the absolute one keeps shrinking.
// Coefficients are k-major with a fixed stride (coef[k * CK_TAB_STRIDE + interval]): lanes on
// neighbouring intervals read neighbouring LDS banks and k becomes an immediate offset.
#ifndef CK_TAB_DEG
#define CK_TAB_DEG 7
#endif
#define CK_TAB_SHIFT 47            /* bits >> 47 = exponent and top 5 mantissa bits */
#define CK_TAB_MAXINT 768          /* 24 octaves; 8 x 769 doubles = 48 KB of LDS */
// k-stride in doubles.  NOT a multiple of 64: with a 512-byte-multiple stride hipcc fuses the
// reads of two coefficients into ds_read2st64_b64, which runs at half the rate of two
// ds_read_b64 and banks modulo 32 instead of 64 (MI355X_MICROARCH.md, LDS table).
#define CK_TAB_STRIDE 769

struct CkTable {
    double q_lo, q_hi;     // table covers q_lo <= q < q_hi (interval aligned); outside: exact evaluator
    int32_t base;          // (bits(q_lo) >> CK_TAB_SHIFT)
    int32_t n_int;         // number of intervals
    int32_t enabled;       // 0: use the exact evaluator for this block
    int32_t pad_;
    double max_rel_err;    // measured against the exact evaluator when the table was built (see k_table_check)
};

// interval index of q relative to `base` (may lie outside [0, n_int): the caller decides) and
// y = q - centre of q's interval
CK_HD double ck_table_y(double q, int* interval, int base) {
    union { double d; uint64_t u; } v, c;
    v.d = q;
    const uint32_t hi = (uint32_t)(v.u >> 32);
    *interval = (int)(hi >> (CK_TAB_SHIFT - 32)) - base;
    c.u = (uint64_t)((hi & ~((1u << (CK_TAB_SHIFT - 32)) - 1u)) | (1u << (CK_TAB_SHIFT - 33))) << 32;
    return q - c.d;
}

CK_HD double ck_table_poly(const double* coef, int interval, double y) {
    double p = coef[CK_TAB_DEG * CK_TAB_STRIDE + interval];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = CK_TAB_DEG - 1; k >= 0; --k) p = p * y + coef[k * CK_TAB_STRIDE + interval];
    return p;
}

// scaled lag from the squared chord q
CK_HD double ck_s_of_q(const CkMatern& m, int metric, double q) {
    const double d = metric == CK_METRIC_HAVERSINE ? 2.0 * asin(0.5 * sqrt(q)) * CK_EARTH_RADIUS_KM : sqrt(q);
    return m.sqrt2nu * (d / m.len_scale);
}

// covariance entry for lag h >= 0 (src/model.py:193-207)
CK_HD double ck_cov_entry(const CkMatern& m, double h, int add_nugget) {
    if (h == 0.0) return add_nugget ? m.amp + m.nugget : m.amp;
    const double s = m.sqrt2nu * (h / m.len_scale);
    return m.amp * ck_matern_rho_scaled(m, s);
}
