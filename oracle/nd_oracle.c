/*
 * oracle/nd_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference's per-pixel compute path, used only
 * as the checker in tests/, in __graft_entry__.smoke() and as the
 * `cpu_baseline` leg of bench.py.  Nothing under nd_amd/ imports, links or
 * executes anything in this directory; the product path is the HIP library
 * (nd_amd/csrc) and fails loudly without it.
 *
 * What is restated (reference = jnhansen/nd, paths relative to /root/reference):
 *   omnibus  : nd/_change.pyx:20-77, 133-151, 224-287  (types from the shipped
 *              generated C, nd/_change.c:2926-2975, 3501-3590, 6063-6091, 8836-8975)
 *   nlmeans  : nd/_filters.pyx:15-40, 299-420          (nd/_filters.c:2590-2625, 3381-3692)
 *   convolve : scipy.ndimage.convolve as called from nd/filters.py:256-267
 *              (third-party: scipy, un-pinned in the reference's setup.py:126;
 *              scipy 1.15.3 in this image; algorithm = NI_Correlate with the
 *              kernel flipped, origin shifted for even sizes, taps with
 *              |w| <= DBL_EPSILON dropped, double accumulation, cast on store)
 *
 * Third-party arithmetic that is NOT in /root/reference and NOT in this image:
 *   GSL `gsl_cdf_chisq_P` (call sites nd/_change.pyx:147-148; GSL is an
 *   un-pinned system library, reference setup.py:60-95 skips nd._change when
 *   it is missing).  Its published definition is restated here:
 *   gsl_cdf_chisq_P(x, nu) = gsl_cdf_gamma_P(x, nu/2, 2): 0 for x <= 0,
 *   1 - Q(a, x/2) for x/2 > a, else P(a, x/2), with P/Q the regularised
 *   incomplete gamma functions.
 *
 * PARITY PINNING (see DESIGN.md "Oracle"):
 *   nlmeans  : pinned against the reference itself (nd/_filters.pyx compiled
 *              unmodified into oracle/_ref by oracle/build_ref.py) -> tests/golden.
 *   convolve : pinned against scipy.ndimage.convolve itself (present here).
 *   omnibus  : nd/_change.pyx cannot be built here (needs GSL + CythonGSL,
 *              both absent; no stand-ins are written).  Pinned by the
 *              reference's own known-answer test (nd/tests/test_change_omnibus.py:6-19),
 *              its NaN-robustness test (nd/tests/test_change_common.py:21-32),
 *              by scipy.stats.chi2.cdf -- the CDF the reference itself uses at
 *              its other call site nd/_change.pyx:123-124 -- and by mpmath.
 *              The float p-values of the GSL call sites are therefore pinned
 *              through the function's definition, not through a GSL run.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off, optional -fopenmp).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <float.h>

/* ---- nd/_change.pyx:20-39 (nd/_change.c:2926-2975) ---------------------- */
static double orc_f(double p, double k, double n)
{
    (void)n;
    return ((k - 1.0) * (p * p));
}

static double orc_rho(double p, double k, double n)
{
    return (1.0 - ((((2.0 * (p * p)) - 1.0) / ((6.0 * (k - 1.0)) * p))
                   * ((k / n) - (1.0 / (n * k)))));
}

static double orc_omega2(double p, double k, double n, double rho)
{
    return (((((p * p) * ((p * p) - 1.0)) / (24.0 * (rho * rho)))
             * ((k / (n * n)) - (1.0 / ((n * k) * (n * k)))))
            - ((((p * p) * (k - 1.0)) / 4.0)
               * ((1.0 - (1.0 / rho)) * (1.0 - (1.0 / rho)))));
}

/* ---- regularised incomplete gamma, generic real a > 0 ------------------- */
static double orc_lgamma(double x)
{
    int sign;
    return lgamma_r(x, &sign);
}

static double orc_gamma_prefactor(double a, double x)
{
    /* x^a e^-x / Gamma(a) */
    return exp((a * log(x) - x) - orc_lgamma(a));
}

static double orc_gamma_P_series(double a, double x)
{
    double ap = a, del = 1.0 / a, sum = del;
    int n;
    for (n = 1; n < 100000; n++) {
        ap += 1.0;
        del *= x / ap;
        sum += del;
        if (fabs(del) < fabs(sum) * 1e-17) break;
    }
    return sum * orc_gamma_prefactor(a, x);
}

static double orc_gamma_Q_cf(double a, double x)
{
    /* modified Lentz evaluation of the continued fraction for Q(a,x) */
    const double tiny = 1e-300;
    double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
    int i;
    for (i = 1; i < 100000; i++) {
        double an = -(double)i * ((double)i - a), del;
        b += 2.0;
        d = an * d + b;
        if (fabs(d) < tiny) d = tiny;
        c = b + an / c;
        if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        del = d * c;
        h *= del;
        if (fabs(del - 1.0) < 1e-16) break;
    }
    return orc_gamma_prefactor(a, x) * h;
}

double oracle_gammainc_P(double a, double x)
{
    if (!(a > 0.0) || isnan(x) || x < 0.0) return NAN;
    if (x == 0.0) return 0.0;
    if (x < a + 1.0) return orc_gamma_P_series(a, x);
    return 1.0 - orc_gamma_Q_cf(a, x);
}

double oracle_gammainc_Q(double a, double x)
{
    if (!(a > 0.0) || isnan(x) || x < 0.0) return NAN;
    if (x == 0.0) return 1.0;
    if (x < a + 1.0) return 1.0 - orc_gamma_P_series(a, x);
    return orc_gamma_Q_cf(a, x);
}

/* gsl_cdf_chisq_P(x, nu), call sites nd/_change.pyx:147-148.  GSL is a linked system library of
 * the reference (setup.py:69; the shipped C was generated against gsl 2.5, nd/_change.c:22) and is
 * absent from this image, so the function is restated from GSL's published source, branch by
 * branch (GNU Scientific Library 2.5, files as named; the same text in 2.4 - 2.7):
 *
 *   cdf/chisq.c   gsl_cdf_chisq_P (x, nu)      = gsl_cdf_gamma_P (x, nu / 2, 2.0)
 *   cdf/gamma.c   gsl_cdf_gamma_P (x, a, b):     y = x / b;
 *                                                if (x <= 0.0) return 0.0;                  [B1]
 *                                                if (y > a) P = 1 - gsl_sf_gamma_inc_Q (a, y);  [B2]
 *                                                else       P = gsl_sf_gamma_inc_P (a, y);      [B3]
 *   specfunc/gamma_inc.c  gsl_sf_gamma_inc_P_e / _Q_e evaluate the regularised incomplete gamma
 *                 function by series (x small against a), continued fraction (gamma_inc_Q_CF,
 *                 a <= x <= 1e6), a uniform asymptotic form (a >= 1e6) and, for x > 1e6,
 *                 gamma_inc_Q_large_x = gamma_inc_D (a, x) * (a / x) * sum, with
 *                 gamma_inc_D (a, x) = exp (a ln x - x - lnGamma (a + 1))  (a < 10; a log1pmx form
 *                 of the same quantity otherwise).  Every one of them evaluates the SAME function
 *                 Q(a, y) / P(a, y) to ~1e-15: the restatement below uses a series and a continued
 *                 fraction and is pinned against scipy / mpmath to 2e-13, so it agrees with GSL to
 *                 that level wherever GSL is accurate -- nine orders inside the 1e-5 budget.
 *
 * Edge semantics, each pinned by tests/test_oracle_omnibus.py::test_chisq_gsl_branches:
 *   x <= 0 (incl. -inf, -0.0)  ->  0.0        branch B1, taken before anything is evaluated
 *   x = NaN                    ->  NaN        B1 and B2 compare false, B3: gsl_sf_gamma_inc_P_e's
 *                                             series on NaN (no domain error: `x < 0` is false)
 *   x = +inf                   ->  NaN        B2: y = inf > a; gamma_inc_Q_large_x -> gamma_inc_D:
 *                                             exp (a ln(inf) - inf - ...) = exp (inf - inf) = NaN
 *                                             (the a >= 10 form computes log(1 + mu) - mu with
 *                                             mu = inf: again inf - inf), so P = 1 - NaN = NaN.
 *                                             scipy's gammainc returns 1 here; the reference's
 *                                             decision `P > alpha` is false for NaN: no change.
 *   large finite x             ->  1 - Q      B2 (y > a): Q underflows to 0 smoothly, P -> 1
 * (GSL's error handler is not involved in any of these: gsl_sf_gamma_inc_Q/_P only raise for
 * a < 0 or x < 0, which B1 and the callers' a = f / 2 > 0 exclude.) */
static double orc_cdf_chisq_P(double x, double nu)
{
    double a = nu / 2.0;
    double y = x / 2.0;
    if (x <= 0.0) return 0.0;
    if (isnan(x) || isinf(x) || !(a > 0.0)) return NAN;
    if (y > a) return 1.0 - oracle_gammainc_Q(a, y);
    return oracle_gammainc_P(a, y);
}

double oracle_cdf_chisq_P(double x, double nu) { return orc_cdf_chisq_P(x, nu); }
double oracle_f(double p, double k, double n) { return orc_f(p, k, n); }
double oracle_rho(double p, double k, double n) { return orc_rho(p, k, n); }
double oracle_omega2(double p, double k, double n, double rho) { return orc_omega2(p, k, n, rho); }

/* ---- nd/_filters.pyx:15-40  _idx, EDGE_MODE_REFLECT (whole-sample) ------ */
static ptrdiff_t orc_idx(ptrdiff_t i, ptrdiff_t shape)
{
    if (i < 0) return -i;
    else if (i >= shape) return 2 * shape - 2 - i;
    else return i;
}

/* ---- nd/_filters.pyx:299-314  find_weight ------------------------------- */
static double orc_find_weight(double weight_sum, double sq_weight_sum, double n, int *err)
{
    double rt;
    *err = 0;
    /* shipped C raises ZeroDivisionError for sq_weight_sum == 0
     * (nd/_filters.c:2603-2606), ValueError('No solution') below */
    if (sq_weight_sum == 0) { *err = 2; return 0.0; }
    if ((n - 1.0) > ((weight_sum * weight_sum) / sq_weight_sum)) { *err = 1; return 0.0; }
    rt = sqrt(((((n * weight_sum) * weight_sum) - ((n * n) * sq_weight_sum))
               + (n * sq_weight_sum)));
    if ((n - 1.0) == 0) { *err = 2; return 0.0; }
    return (weight_sum + rt) / (n - 1.0);
}

double oracle_find_weight(double W, double W2, double n, int *err)
{
    return orc_find_weight(W, W2, n, err);
}

/* ---- scipy.ndimage border handling (ni_support.c NI_InitFilterOffsets) --- */
/* mode: 0 reflect (d c b a | a b c d | d c b a)  [scipy default, the only
 * one the reference's tests exercise], 1 constant (returns -1 = use cval),
 * 2 nearest, 3 mirror, 4 wrap. */
static int64_t orc_extend(int64_t cc, int64_t len, int mode)
{
    if (cc >= 0 && cc < len) return cc;
    switch (mode) {
    case 0:
        if (len <= 1) return 0;
        if (cc < 0) {
            int64_t sz2 = 2 * len;
            if (cc < -sz2) cc += sz2 * (-cc / sz2);
            return cc < -len ? cc + sz2 : -cc - 1;
        } else {
            int64_t sz2 = 2 * len;
            cc -= sz2 * (cc / sz2);
            if (cc >= len) cc = sz2 - cc - 1;
            return cc;
        }
    case 1:
        return -1;
    case 2:
        return cc < 0 ? 0 : len - 1;
    case 3:
        if (len <= 1) return 0;
        if (cc < 0) {
            int64_t sz2 = 2 * len - 2;
            cc = sz2 * (-cc / sz2) + cc;
            return cc <= 1 - len ? cc + sz2 : -cc;
        } else {
            int64_t sz2 = 2 * len - 2;
            cc -= sz2 * (cc / sz2);
            if (cc >= len) cc = sz2 - cc;
            return cc;
        }
    case 4:
        if (len <= 1) return 0;
        if (cc < 0) {
            cc += len * (-cc / len);
            if (cc < 0) cc += len;
            return cc;
        } else {
            cc -= len * (cc / len);
            return cc;
        }
    }
    return -1;
}

static int64_t orc_reflect_half(int64_t cc, int64_t len) { return orc_extend(cc, len, 0); }

int64_t oracle_extend(int64_t cc, int64_t len, int mode) { return orc_extend(cc, len, mode); }

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define SFX(x) CAT(x, _f32)
#include "nd_oracle_impl.h"
#undef REAL
#undef SFX

#define REAL double
#define SFX(x) CAT(x, _f64)
#include "nd_oracle_impl.h"
#undef REAL
#undef SFX

/* Generic-mode correlate (modes other than reflect, cval) -- double only path
 * selection is done in oracle.py; both dtypes go through this. */
#define DEFINE_CORRELATE_MODE(REAL_T, NAME)                                              \
int NAME(const REAL_T *in, REAL_T *out, const int64_t A[4], const int64_t si[4],        \
         const int64_t so[4], int64_t ntaps, const int64_t *offs, const double *w,      \
         int mode, double cval)                                                          \
{                                                                                        \
    for (int64_t i0 = 0; i0 < A[0]; i0++)                                                \
    for (int64_t i1 = 0; i1 < A[1]; i1++)                                                \
    for (int64_t i2 = 0; i2 < A[2]; i2++)                                                \
    for (int64_t i3 = 0; i3 < A[3]; i3++) {                                              \
        double tmp = 0.0;                                                                \
        for (int64_t t = 0; t < ntaps; t++) {                                            \
            int64_t j0 = orc_extend(i0 + offs[4 * t + 0], A[0], mode);                   \
            int64_t j1 = orc_extend(i1 + offs[4 * t + 1], A[1], mode);                   \
            int64_t j2 = orc_extend(i2 + offs[4 * t + 2], A[2], mode);                   \
            int64_t j3 = orc_extend(i3 + offs[4 * t + 3], A[3], mode);                   \
            if (j0 < 0 || j1 < 0 || j2 < 0 || j3 < 0) tmp += w[t] * cval;                \
            else tmp += w[t] * (double)in[j0 * si[0] + j1 * si[1] + j2 * si[2] + j3 * si[3]]; \
        }                                                                                \
        out[i0 * so[0] + i1 * so[1] + i2 * so[2] + i3 * so[3]] = (REAL_T)tmp;            \
    }                                                                                    \
    return 0;                                                                            \
}
DEFINE_CORRELATE_MODE(float, oracle_correlate_mode_f32)
DEFINE_CORRELATE_MODE(double, oracle_correlate_mode_f64)

int oracle_abi_version(void) { return 1; }
