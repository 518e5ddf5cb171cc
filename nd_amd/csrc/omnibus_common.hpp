// nd_amd/csrc/omnibus_common.hpp -- pieces shared by the dual-pol (omnibus.hip) and full-pol
// (omnibus_c3.hip) omnibus kernels: the per-test constant table, the chi-square pair, the
// approximate logarithm of the screen, and the host code that builds the table (rho, omega2,
// decision bounds).  p = 2 follows nd/_change.pyx:20-39, 133-151 to the letter; p = 3 is the same
// formulas with p = 3 (the reference hard-codes p = 2, nd/_change.pyx:51,99,135).
#pragma once

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.hpp"

namespace nd_amd {

// ---- constants of one omnibus test over j matrices (host-computed in double) ------------
struct OmniTabEntry {
    double m2rho;    // -2.0 * (double)(T)rho(p, j, n)        nd/_change.pyx:75-76
    double pklogk;   // (double)(p * j) * log((double)j)      nd/_change.pyx:74
    double omega2;   // omega2(p, j, n, rho) from double rho  nd/_change.pyx:139
    double lgam;     // lgamma(a + 1), a = f/2 = (j - 1) p^2 / 2
    double zlo;      // fast-reject bound: z < zlo  =>  P <= alpha for certain (see omni_bounds)
    double zlo_a;    // the same bound for z_approx (hardware f32 log2), widened by its error
    double zhi;      // fast-accept bound: zhi < z < inf  =>  P > alpha for certain
    double zhi_a;    // the same for z_approx
};

constexpr int kTabArgs = 96;   // largest k whose table travels as a kernel argument
struct OmniTab {
    OmniTabEntry e[kTabArgs + 1];
};

// 2/m, i.e. the reciprocal of the (half-)integer m/2, for the incomplete-gamma recurrences.
// For even m this is bit-for-bit 1/(m/2).
constexpr int kInvTab = 4096;
struct InvTab {
    double v[kInvTab];
    constexpr InvTab() : v()
    {
        v[0] = 0.0;
        for (int i = 1; i < kInvTab; ++i) v[i] = 2.0 / (double)i;
    }
};
static __constant__ InvTab c_inv2 = InvTab();

__device__ __forceinline__ double inv_half(int m2)   // 1 / (m2 / 2)
{
    return m2 < kInvTab ? c_inv2.v[m2] : 2.0 / (double)m2;
}

// P1 = P(a, z/2), P2 = P(a + 2, z/2) for a = a2/2 (integer or half-integer), N values in lockstep.
//   t_a = x^a e^-x / Gamma(a+1)
//   x <  a+1 :  P(a,x) = t_a * sum_{n>=0} x^n / ((a+1)...(a+n)),  P(a+2,x) = t_a * sum_{n>=2} ...
//   x >= a+1 :  Q(a,x) = t_{a-1} * sum_m (a-1)...(a-m) / x^m  over the factors >= 1
//               (+ erfc(sqrt x) when a is a half-integer),     Q(a+2,x) = Q(a,x) + t_a + t_{a+1}
// Both sums have decreasing positive terms; one loop serves both, four terms per trip, the
// reciprocals of the next trip fetched while this one computes.
// a2 = 4 (j-1) for dual pol (a integer: the gsl_cdf_chisq_P(z, f), (z, f+4) pair of
// nd/_change.pyx:147-148), 9 (j-1) for full pol.
template <int N>
__device__ __forceinline__ void chisq_pair(const double (&z)[N], int a2, double lgam_a1,
                                           double (&P1)[N], double (&P2)[N])
{
    double x[N], ta[N], rx[N], u1[N], term[N], sum[N];
    bool lower[N], ok[N];
    const double a = 0.5 * (double)a2;
    const double ap1 = a + 1.0;
    const double inv_ap1 = inv_half(a2 + 2);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ok[i] = (z[i] > 0.0) && (z[i] < INFINITY);
        x[i] = ok[i] ? 0.5 * z[i] : 1.0;
        lower[i] = x[i] < ap1;
        ta[i] = exp(fma(a, log(x[i]), -x[i]) - lgam_a1);
        rx[i] = 1.0 / x[i];
        u1[i] = x[i] * inv_ap1;
        term[i] = lower[i] ? u1[i] : 1.0;
        sum[i] = lower[i] ? 0.0 : 1.0;
    }
    double inv_cur[4], inv_nxt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) inv_cur[u] = inv_half(a2 + 4 + 2 * u);
    for (int n = 1; n < 4000000; n += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) inv_nxt[u] = inv_half(a2 + 2 * (n + 4 + u) + 2);
        bool more = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d2 = a2 - 2 * (n + u);                 // 2 (a - m)
            const double up = d2 >= 2 ? 0.5 * (double)d2 : 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double ratio = lower[i] ? x[i] * inv_cur[u] : up * rx[i];
                term[i] = term[i] * ratio;
                sum[i] = sum[i] + term[i];
                if (u == 3) more = more || (term[i] > 1e-17 * sum[i]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) inv_cur[u] = inv_nxt[u];
        if (!__any(more)) break;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double p1, p2;
        if (lower[i]) {
            p1 = ta[i] * ((1.0 + u1[i]) + sum[i]);
            p2 = ta[i] * sum[i];
        } else {
            double qa = (ta[i] * a * rx[i]) * sum[i];
            if (a2 & 1) qa = qa + erfc(sqrt(x[i]));
            p1 = 1.0 - qa;
            p2 = 1.0 - (qa + ta[i] + ta[i] * u1[i]);
        }
        if (!ok[i]) {
            // gsl_cdf_chisq_P (GSL cdf/gamma.c: gsl_cdf_gamma_P): x <= 0 -> 0 before anything is
            // evaluated; NaN stays NaN; +inf -> NaN (y > a -> 1 - gsl_sf_gamma_inc_Q, whose
            // large-x form evaluates exp(a ln x - x - ...) = exp(inf - inf); the branch list is
            // next to orc_cdf_chisq_P in oracle/nd_oracle.c)
            p1 = p2 = (z[i] <= 0.0) ? 0.0 : NAN;
        }
        P1[i] = p1;
        P2[i] = p2;
    }
}

// Cheap stand-in for ln used only to screen: ln x = (exponent + log2(mantissa)) ln 2 with the
// mantissa's log2 from the hardware v_log_f32 (1 ulp on [0.5, 1), i.e. <= 6e-8 absolute).  NaN,
// +-inf and zero arguments propagate exactly as in the double evaluation, so an approximate z is
// NaN or infinite precisely when the exact one is.
__device__ __forceinline__ double approx_ln(double x)
{
    int e;
    const double m = frexp(x, &e);
    return ((double)e + (double)__log2f((float)m)) * 0.6931471805599453;
}

// P = P1 + omega2 (P2 - P1) with the reference's rounding points (nd/_change.c:6087-6089)
template <typename T>
__device__ __forceinline__ T combine_P(double P1, double P2, double omega2)
{
    const T p1 = (T)P1, p2 = (T)P2;
    const T d = p2 - p1;
    return (T)((double)p1 + (omega2 * (double)d));
}

// =========================================================================================
// host side
// =========================================================================================
static inline double host_rho(double p, double k, double n)
{
    return (1.0 - ((((2.0 * (p * p)) - 1.0) / ((6.0 * (k - 1.0)) * p)) *
                   ((k / n) - (1.0 / (n * k)))));
}

static inline double host_omega2(double p, double k, double n, double rho)
{
    return (((((p * p) * ((p * p) - 1.0)) / (24.0 * (rho * rho))) *
             ((k / (n * n)) - (1.0 / ((n * k) * (n * k))))) -
            ((((p * p) * (k - 1.0)) / 4.0) * ((1.0 - (1.0 / rho)) * (1.0 - (1.0 / rho)))));
}

// host twin of chisq_pair (N = 1, no table), used only to place the decision bounds
static inline void host_chisq_pair(double z, int a2, double lgam_a1, double *P1, double *P2)
{
    if (!(z > 0.0)) {
        *P1 = *P2 = (z <= 0.0) ? 0.0 : NAN;
        return;
    }
    if (!(z < INFINITY)) {
        *P1 = *P2 = NAN;
        return;
    }
    const double a = 0.5 * (double)a2;
    const double x = 0.5 * z;
    const bool lower = x < a + 1.0;
    const double ta = exp((a * log(x) - x) - lgam_a1);
    const double u1 = x / (a + 1.0);
    double term = lower ? u1 : 1.0, sum = lower ? 0.0 : 1.0;
    for (int n = 1; n < 4000000; ++n) {
        const int d2 = a2 - 2 * n;
        const double ratio = lower ? x / (a + 1.0 + (double)n) : (d2 >= 2 ? 0.5 * (double)d2 : 0.0) / x;
        term *= ratio;
        sum += term;
        if (!(term > 1e-17 * sum)) break;
    }
    if (lower) {
        *P1 = ta * ((1.0 + u1) + sum);
        *P2 = ta * sum;
    } else {
        double qa = (ta * a / x) * sum;
        if (a2 & 1) qa += erfc(sqrt(x));
        *P1 = 1.0 - qa;
        *P2 = 1.0 - (qa + ta + ta * u1);
    }
}

// Decision bounds of the test over j matrices: every z' < zlo has P(z') <= alpha for certain and
// every finite z' > zhi has P(z') > alpha for certain, so the chi-square pair is only needed for
// zlo <= z <= zhi (and for z = +inf, whose P is NaN).
//   P(z) = P1 + omega2 (P2 - P1) is non-decreasing in z when 0 <= omega2 <= 1 (a mixture of two
//   chi-square CDFs).  The kernel's P differs from the exact one by the roundings to T of P1, P2,
//   their difference and the result, plus ~1e-13 from the series: bounded by `margin` below.
//   zlo = the z where the exact P equals alpha - margin, stepped down by 1e-9 relative;
//   zhi = the z where it equals alpha + margin, stepped up by 1e-9 relative.
//   Outside 0 <= omega2 <= 1 (e.g. n = 1, small j), or when a target leaves (0, 1), the bound is
//   -inf / +inf: every non-NaN z is evaluated exactly.
template <typename T>
static void omni_bounds(int j, int a2, double omega2, double lgam, double alpha, double *zlo,
                        double *zhi)
{
    *zlo = -INFINITY;    // evaluate everything exactly
    *zhi = INFINITY;     // never accept without evaluating
    if (j < 2) return;
    if (!(omega2 >= 0.0 && omega2 <= 1.0) || !(alpha == alpha)) return;
    const double ulp = sizeof(T) == 4 ? 5.9604644775390625e-08 : 1.1102230246251565e-16;
    // the device's double evaluation: ~1e-13 from the series, plus the prefactor
    // exp(a ln x - x - lgamma(a + 1)), whose exponent carries ~a (1 + ln a) eps of absolute
    // rounding error -- negligible at a = 2 (k - 1) <= 100, 2e-10 for series of 10^4 dates
    const double a_half = 0.5 * (double)a2;
    const double margin = 16.0 * ulp * (1.0 + 2.0 * omega2) + 1e-11 +
                          8.0 * a_half * (1.0 + log(a_half + 2.0)) * 1.1102230246251565e-16;
    auto Pz = [&](double z) {
        double p1, p2;
        host_chisq_pair(z, a2, lgam, &p1, &p2);
        return p1 + omega2 * (p2 - p1);
    };
    // smallest z (to 1e-15 relative) with exact P(z) >= target, as a bracketing pair lo < hi
    auto quantile = [&](double target, double *lo_out, double *hi_out) -> bool {
        double lo = 0.0, hi = 2.0 * (double)a2 + 64.0;
        int guard = 0;
        while (Pz(hi) < target && guard++ < 64) hi *= 2.0;
        if (guard >= 64) return false;
        for (int it = 0; it < 200; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (Pz(mid) < target)
                lo = mid;
            else
                hi = mid;
            if (hi - lo <= 1e-15 * hi) break;
        }
        *lo_out = lo;
        *hi_out = hi;
        return true;
    };
    double lo, hi;
    const double tlo = alpha - margin;
    if (tlo >= 1.0) {
        *zlo = INFINITY;                      // P <= 1 < alpha: nothing can fire
    } else if (tlo >= 0.0) {
        if (quantile(tlo, &lo, &hi))
            *zlo = lo * (1.0 - 1e-9);
        else
            *zlo = INFINITY;                  // target unreachable in double
    }
    const double thi = alpha + margin;
    if (thi > 0.0 && thi < 1.0 - 1e-9 && quantile(thi, &lo, &hi)) *zhi = hi * (1.0 + 1e-9);
}

// a2 = 2a = f = (j - 1) p^2
static inline int omni_a2(int j, int p) { return (j - 1) * p * p; }

template <typename T>
static OmniTabEntry make_entry(int j, uint32_t n_looks, double alpha, int pol)
{
    OmniTabEntry e;
    const double p = (double)pol, k = (double)j, n = (double)n_looks;
    const double rho = host_rho(p, k, n);
    const T rho_t = (T)rho;
    e.m2rho = -2.0 * (double)rho_t;
    const T pk = (T)pol * (T)j;                       // `p * k` in `floating`, nd/_change.c:3580
    e.pklogk = (double)pk * log(k);
    e.omega2 = host_omega2(p, k, n, rho);
    const int a2 = omni_a2(j, pol);
    e.lgam = lgamma(0.5 * (double)a2 + 1.0);
    omni_bounds<T>(j, a2, e.omega2, e.lgam, alpha, &e.zlo, &e.zhi);
    // bounds for the f32-log2 screen: ten times its worst-case error outside [zlo, zhi]
    // (|z_approx - z| <= |m2rho| n (j + 1) 1e-7)
    const double aerr = 1e-6 * fabs(e.m2rho) * n * (k + 1.0);
    e.zlo_a = (e.zlo > -INFINITY && e.zlo < INFINITY) ? e.zlo - (aerr + 1e-9 * fabs(e.zlo)) : e.zlo;
    e.zhi_a = (e.zhi < INFINITY) ? e.zhi + (aerr + 1e-9 * fabs(e.zhi)) : INFINITY;
    if (!(aerr == aerr) || !(aerr < INFINITY)) {   // rho is NaN/inf for j = 1: exact path only
        e.zlo = e.zlo_a = -INFINITY;
        e.zhi = e.zhi_a = INFINITY;
    }
    return e;
}

// ---- constants of the in-register search's screen (omnibus.hip: dense_search) ----------------
// The search decides a test from  L2 = log2(prod of determinants) - j * log2(det of sum),
// z = z0 + c * L2 with z0 = m2rho n pklogk and c = m2rho n ln 2 (< 0 for rho > 0), evaluated as
//   x = L2 - R  in float, relative to a reference point R = re + rf near the decision:
//   x < a  =>  the test fires for certain;   x > b  =>  it cannot fire;   otherwise: undecided,
// the pixel is handed to pass B (exact evaluation).  Error budget of the device's x (see
// dense_search): j * (6e-8 hardware log2 + 1.5e-8 fixed point) per determinant, j * 6e-8 for the
// determinant of the sum, < 6e-6 float32 arithmetic  =>  < 9e-6 at j = 24; `mg` below is more than
// twice that, plus the rounding of z to T that the exact bounds zlo / zhi refer to.
struct DenseScreenEntry {
    int re;
    float rf, a, b;
};
constexpr int kDenseMax = 128;
struct DenseScreen {
    DenseScreenEntry e[kDenseMax + 1];
};

template <typename T>
static DenseScreenEntry make_dense_entry(const OmniTabEntry &t, int j, uint32_t n_looks)
{
    DenseScreenEntry d;
    d.re = 0;
    d.rf = 0.f;
    d.a = -INFINITY;    // never "fires for certain"
    d.b = INFINITY;     // never "cannot fire"  => every test of this j is handed over
    const double c = t.m2rho * (double)n_looks * 0.6931471805599453;
    const double z0 = t.m2rho * (double)n_looks * t.pklogk;
    if (j < 2 || !(c < 0.0) || !(c > -INFINITY) || !(z0 == z0) || !(fabs(z0) < INFINITY)) return d;
    if (t.zlo == INFINITY) {          // P <= 1 < alpha: nothing can fire
        d.b = -INFINITY;
        return d;
    }
    const double eps = sizeof(T) == 4 ? 1.1920928955078125e-07 : 2.220446049250313e-16;
    const bool hi_ok = t.zhi < INFINITY && t.zhi > -INFINITY;
    const bool lo_ok = t.zlo > -INFINITY && t.zlo < INFINITY;
    // beyond 32 dates the float32 arithmetic of x works on magnitudes up to j (ulp 7.6e-6 at 64) and
    // the fixed-point sum of the mantissa logs passes 2^24 before it is converted: the budget grows
    // to ~2.5e-5 at j = 64 and ~6e-5 at j = 128, the margin with it (4.6e-5 / 1.7e-4)
    const double mg0 = 2e-5 + 4e-7 * (double)j + (j > 32 ? 1e-6 * (double)(j - 32) : 0.0);
    double Lhi = 0, Llo = 0;
    if (hi_ok) {
        const double zr = 2.0 * eps * fabs(t.zhi);                 // (T) rounding of z
        Lhi = (t.zhi + zr - z0) / c;                               // z > zhi + zr  <=>  L2 < Lhi
        Lhi -= mg0 + 1e-12 * fabs(Lhi);
    }
    if (lo_ok) {
        const double zr = 2.0 * eps * fabs(t.zlo);
        Llo = (t.zlo - zr - z0) / c;                               // z < zlo - zr  <=>  L2 > Llo
        Llo += mg0 + 1e-12 * fabs(Llo);
    }
    if (!hi_ok && !lo_ok) return d;
    const double R = hi_ok ? Lhi : Llo;
    if (!(fabs(R) < 5e8)) return d;
    const double fl = floor(R);
    d.re = (int)fl;
    d.rf = (float)(R - fl);
    if (hi_ok) {
        d.a = 0.f;
        d.b = lo_ok ? (float)(Llo - R) + 1e-6f : INFINITY;
        if (lo_ok && !(Llo >= R)) {      // cannot happen (zlo <= zhi); be safe: exact only
            d.a = -INFINITY;
            d.b = INFINITY;
        }
    } else {
        d.a = -INFINITY;
        d.b = 0.f;
    }
    return d;
}

template <typename T>
static DenseScreen make_dense_screen(const std::vector<OmniTabEntry> &tab, int k, uint32_t n_looks)
{
    DenseScreen s;
    memset(&s, 0, sizeof(s));
    for (int j = 0; j <= kDenseMax; ++j) {
        s.e[j].a = -INFINITY;
        s.e[j].b = INFINITY;
        if (j >= 1 && j <= k) s.e[j] = make_dense_entry<T>(tab[(size_t)j], j, n_looks);
    }
    return s;
}

typedef float f2_t __attribute__((ext_vector_type(2)));

// ---- constants of the streaming search (omnibus.hip: omnibus_c2_stream_kernel) -----------------
// The kernel walks the dates last to first, so the global test it meets at step jj = 1, 2, ... is
// the one over jj dates whatever k is: entry jj is read with one scalar load from the kernel's
// argument segment (a wave-uniform index into a by-value argument compiles to s_load_dwordx8), no
// vector instruction, no LDS.  jf / cj / mj are the wave-uniform factors of the rounding band of
// that test (see the kernel), precomputed so that they cost no conversions on the device.
struct StreamEntry {
    int re;
    float rf, a, b;     // as DenseScreenEntry
    float jf;           // (float) jj
    float cj;           // 1.46 * 5 u * jj:  rel = cj * s11 s22 / det
    float mj;           // 1.01 * jj:        band = mj * rel
    float pad;
};
// The 2- and 3-date marginal tests are decided without logarithms: with L2 = log2(prod det_t) -
// j log2 det(sum),  L2 < Lhi  <=>  prod det_t < 2^Lhi det(sum)^j.  The products of two or three
// determinants are formed in `floating` (each determinant inside [dlo, dhi], so the product is a
// normal number with j - 1 roundings), det(sum)^j with j - 1 roundings, the constant with one:
// <= 6 half-ulps, 4.3e-7 in log2 units at float32; the constants carry a margin of 4e-6.
//   prod < ca_j * det(sum)^j  =>  the test fires for certain;  prod > cb_j * det(sum)^j  =>  it cannot.
// A right-hand side that underflows is harmless (the product is a normal number, larger than
// anything that underflows: both verdicts are then true statements); overflow is excluded by
// det(sum) < dhi and ca, cb <= 64.
template <int NJ>
struct StreamScreen {
    StreamEntry e[NJ + 1];
    f2_t ca, cb;          // .x: the 2-date test, .y: the 3-date test (pairs: operands of packed multiplications)
    float dlo, dhi;       // a date's determinant and those of the 2- / 3-date sums: strictly inside (dlo, dhi)
    // the same constants in a longer table (entries beyond NJ: for the caller to fill)
    template <int NJ2>
    StreamScreen<NJ2> widen() const
    {
        static_assert(NJ2 >= NJ, "widen");
        StreamScreen<NJ2> w;
        memset(&w, 0, sizeof(w));
        for (int j = 0; j <= NJ; ++j) w.e[j] = e[j];
        w.ca = ca;
        w.cb = cb;
        w.dlo = dlo;
        w.dhi = dhi;
        return w;
    }
};

template <typename T>
static void stream_marginal_bounds(const OmniTabEntry &t, int j, uint32_t n_looks, float *ca, float *cb)
{
    *ca = 0.f;            // never "fires for certain"
    *cb = INFINITY;       // never "cannot fire"
    const double c = t.m2rho * (double)n_looks * 0.6931471805599453;
    const double z0 = t.m2rho * (double)n_looks * t.pklogk;
    if (j < 2 || !(c < 0.0) || !(c > -INFINITY) || !(z0 == z0) || !(fabs(z0) < INFINITY)) return;
    if (t.zlo == INFINITY) {          // P <= 1 < alpha: nothing can fire
        *cb = 0.f;
        return;
    }
    const double eps = sizeof(T) == 4 ? 1.1920928955078125e-07 : 2.220446049250313e-16;
    const double mgp = 4e-6;
    if (t.zhi < INFINITY && t.zhi > -INFINITY) {
        const double zr = 2.0 * eps * fabs(t.zhi);
        double L = (t.zhi + zr - z0) / c;
        L -= mgp + 1e-12 * fabs(L);
        if (L == L && L > -100.0) {
            if (L > 6.0) L = 6.0;                                // a weaker claim, still true
            float v = (float)exp2(L);
            if ((double)v > exp2(L)) v = nextafterf(v, 0.f);     // rounded down
            *ca = v;
        }
    }
    if (t.zlo > -INFINITY && t.zlo < INFINITY) {
        const double zr = 2.0 * eps * fabs(t.zlo);
        double L = (t.zlo - zr - z0) / c;
        L += mgp + 1e-12 * fabs(L);
        if (L == L && L <= 6.0) {
            if (L < -100.0) L = -100.0;                          // a weaker claim, still true
            float v = (float)exp2(L);
            if ((double)v < exp2(L)) v = nextafterf(v, INFINITY);   // rounded up
            *cb = v;
        }
    }
    if (!(*ca <= *cb)) {              // cannot happen (zlo <= zhi); be safe: exact only
        *ca = 0.f;
        *cb = INFINITY;
    }
}

template <typename T, int NJ>
static StreamScreen<NJ> make_stream_screen(const std::vector<OmniTabEntry> &tab, const DenseScreen &scr,
                                           int k, uint32_t n_looks)
{
    StreamScreen<NJ> s;
    memset(&s, 0, sizeof(s));
    const float cu = (sizeof(T) == 4 ? 5.9604645e-08f : 1.1102230e-16f) * 7.5f;   // 1.46 * 5 u, rounded up
    for (int j = 0; j <= NJ; ++j) {
        s.e[j].re = scr.e[j].re;
        s.e[j].rf = scr.e[j].rf;
        s.e[j].a = scr.e[j].a;
        s.e[j].b = scr.e[j].b;
        s.e[j].jf = (float)j;
        s.e[j].cj = cu * (float)j;
        s.e[j].mj = (float)j * 1.01f;
    }
    float ca2 = 0.f, ca3 = 0.f, cb2 = INFINITY, cb3 = INFINITY;
    if (k >= 2) stream_marginal_bounds<T>(tab[2], 2, n_looks, &ca2, &cb2);
    if (k >= 3) stream_marginal_bounds<T>(tab[3], 3, n_looks, &ca3, &cb3);
    s.ca.x = ca2;
    s.ca.y = ca3;
    s.cb.x = cb2;
    s.cb.y = cb3;
    // float32: the determinants of a date and of the 2- / 3-date sums inside 2^+-36 (a product of
    // three stays a normal number, 2^+-108; a cube times a constant <= 64 stays finite); float64:
    // 2^+-100.  Pixels outside go to the exact pass.  The running double product of the determinants
    // then moves by at most 36 (100) binary orders per date: see the kernel's range check.
    if (sizeof(T) == 4) {
        s.dlo = 1.4551915228366852e-11f;     // 2^-36
        s.dhi = 68719476736.f;               // 2^36
    } else {
        s.dlo = 7.888609052210118e-31f;      // 2^-100
        s.dhi = 1.2676506002282294e30f;      // 2^100
    }
    return s;
}

// ---- device side of the screen (shared by the dual-pol and the full-pol kernels) ---------------
// log2 of a positive finite x as (exponent, log2 of the mantissa in [0.5, 1)): the mantissa's
// log2 comes from the hardware v_log_f32 (<= 1 ulp of a value in [-1, 0], i.e. <= 6e-8 absolute).
__device__ __forceinline__ void log2_parts(float x, int &e, float &m)
{
    m = __log2f(__builtin_frexpf(x, &e));
}
__device__ __forceinline__ void log2_parts(double x, int &e, float &m)
{
    m = __log2f((float)__builtin_frexp(x, &e));
    // (float) of a mantissa just below 1 may round to 1: log2 = 0, error < 1e-7 as budgeted
}

constexpr float kLogFix = 33554432.f;          // 2^25: fixed-point scale of the mantissa logs

struct ScreenRegs {
    int re;
    float rf, a, b;
};
__device__ __forceinline__ ScreenRegs screen_regs_load(const DenseScreenEntry *scr_lds, const int lane)
{
    const DenseScreenEntry e = scr_lds[lane + 1];          // entries 1 .. 64 in lanes 0 .. 63
    ScreenRegs r;
    r.re = e.re;
    r.rf = e.rf;
    r.a = e.a;
    r.b = e.b;
    return r;
}
__device__ __forceinline__ DenseScreenEntry screen_entry(const ScreenRegs &r, const int j)
{
    DenseScreenEntry c;
    c.re = __builtin_amdgcn_readlane(r.re, j - 1);
    c.rf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.rf), j - 1));
    c.a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.a), j - 1));
    c.b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.b), j - 1));
    return c;
}

// Lm: sum of the dates' fixed-point mantissa logs (each in [-2^25, 0]): int up to 64 dates, long long
// beyond
template <typename T, typename LmT>
__device__ __forceinline__ float dense_x(const T dets, const bool ok, const int Le, const LmT Lm,
                                         const int jj, const DenseScreenEntry &c)
{
    int es;
    float ms;
    log2_parts(ok ? dets : (T)1, es, ms);
    const int E = (Le - __mul24(jj, es)) - c.re;
    const float F = (float)Lm * (1.0f / kLogFix);
    return (float)E + ((F - c.rf) - (float)jj * ms);
}

// ---- bit masks over the dates (position t = date t) ------------------------------------------
// 32 and 64 dates: plain integers.  Up to 128 dates: two 64-bit words.
struct Bits128 {
    unsigned long long lo, hi;
};
// Up to 192 dates (round 6: the two-pass chain search of series of 129 .. 192 dates): three words.
struct Bits192 {
    unsigned long long w0, w1, w2;
};
template <typename M>
__device__ __forceinline__ M mask_zero()
{
    return (M)0;
}
template <>
__device__ __forceinline__ Bits192 mask_zero<Bits192>()
{
    return Bits192{0ull, 0ull, 0ull};
}
template <>
__device__ __forceinline__ Bits128 mask_zero<Bits128>()
{
    return Bits128{0ull, 0ull};
}
// set bit i (any i, per lane) where `on`
template <typename M>
__device__ __forceinline__ void mask_set(M &m, const int i, const bool on)
{
    m |= on ? ((M)1 << i) : (M)0;
}
__device__ __forceinline__ void mask_set(Bits128 &m, const int i, const bool on)
{
    const unsigned long long b = on ? (1ull << (i & 63)) : 0ull;
    m.lo |= i < 64 ? b : 0ull;
    m.hi |= i < 64 ? 0ull : b;
}
template <typename M>
__device__ __forceinline__ bool mask_bit(const M &m, const int i)
{
    return ((m >> i) & (M)1) != 0;
}
__device__ __forceinline__ bool mask_bit(const Bits128 &m, const int i)
{
    return (((i < 64 ? m.lo : m.hi) >> (i & 63)) & 1ull) != 0ull;
}
// m = 2 m + bit: the streaming search meets the dates last to first, so after the last push the
// bit of date t sits at position t
template <typename M>
__device__ __forceinline__ void mask_push(M &m, const bool bit)
{
    m = (m + m) + (M)(bit ? 1 : 0);
}
// 32 bits: one add-with-carry whose carry-in is the condition's lane mask (the compiler's own
// selection is v_cndmask + v_lshl_or).  The carry-out lands in a scalar pair nobody reads.
__device__ __forceinline__ void mask_push(unsigned &m, const bool bit)
{
    const unsigned long long cin = __builtin_amdgcn_ballot_w64(bit);
    unsigned long long cout;
    unsigned r;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(cout) : "v"(m), "s"(cin));
    m = r;
}
// 64 bits: two of them, the carry of the low word (a lane mask in a scalar pair) feeding the high one
__device__ __forceinline__ void mask_push(unsigned long long &m, const bool bit)
{
    const unsigned long long cin = __builtin_amdgcn_ballot_w64(bit);
    unsigned long long c1, c2;
    const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
    unsigned rlo, rhi;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(rlo), "=s"(c1) : "v"(lo), "s"(cin));
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(rhi), "=s"(c2) : "v"(hi), "s"(c1));
    m = ((unsigned long long)rhi << 32) | rlo;
}
__device__ __forceinline__ void mask_push(Bits128 &m, const bool bit)
{
    m.hi = (m.hi + m.hi) + (m.lo >> 63);
    m.lo = (m.lo + m.lo) + (bit ? 1ull : 0ull);
}
// bits 0 .. n - 1 (0 <= n <= width)
template <typename M>
__device__ __forceinline__ M mask_low(const int n)
{
    return n >= (int)(8 * sizeof(M)) ? ~(M)0 : (((M)1 << n) - (M)1);
}
template <typename M>
__device__ __forceinline__ void mask_keep_low(M &m, const int n)
{
    m &= mask_low<M>(n);
}
__device__ __forceinline__ void mask_keep_low(Bits128 &m, const int n)
{
    m.lo &= mask_low<unsigned long long>(n < 64 ? n : 64);
    m.hi &= mask_low<unsigned long long>(n < 64 ? 0 : n - 64);
}
// bits where neither "fires" nor "cannot fire" is set
template <typename M>
__device__ __forceinline__ M mask_undecided(const M &f, const M &c)
{
    return (M) ~(f | c);
}
__device__ __forceinline__ Bits128 mask_undecided(const Bits128 &f, const Bits128 &c)
{
    return Bits128{~(f.lo | c.lo), ~(f.hi | c.hi)};
}
__device__ __forceinline__ int mask_ctz(const unsigned m) { return __builtin_ctz(m); }
__device__ __forceinline__ int mask_ctz(const unsigned long long m) { return __builtin_ctzll(m); }
// ---- three-word masks ----
__device__ __forceinline__ void mask_set(Bits192 &m, const int i, const bool on)
{
    const unsigned long long b = on ? (1ull << (i & 63)) : 0ull;
    m.w0 |= i < 64 ? b : 0ull;
    m.w1 |= (i >= 64 && i < 128) ? b : 0ull;
    m.w2 |= i >= 128 ? b : 0ull;
}
__device__ __forceinline__ bool mask_bit(const Bits192 &m, const int i)
{
    return (((i < 64 ? m.w0 : (i < 128 ? m.w1 : m.w2)) >> (i & 63)) & 1ull) != 0ull;
}
__device__ __forceinline__ void mask_push(Bits192 &m, const bool bit)
{
    m.w2 = (m.w2 + m.w2) + (m.w1 >> 63);
    m.w1 = (m.w1 + m.w1) + (m.w0 >> 63);
    m.w0 = (m.w0 + m.w0) + (bit ? 1ull : 0ull);
}
__device__ __forceinline__ void mask_keep_low(Bits192 &m, const int n)
{
    m.w0 &= mask_low<unsigned long long>(n < 64 ? (n < 0 ? 0 : n) : 64);
    m.w1 &= mask_low<unsigned long long>(n < 64 ? 0 : (n < 128 ? n - 64 : 64));
    m.w2 &= mask_low<unsigned long long>(n < 128 ? 0 : n - 128);
}
__device__ __forceinline__ Bits192 mask_undecided(const Bits192 &f, const Bits192 &c)
{
    return Bits192{~(f.w0 | c.w0), ~(f.w1 | c.w1), ~(f.w2 | c.w2)};
}
__device__ __forceinline__ unsigned mask_nibble(const Bits192 &m, const int q)
{
    return (unsigned)((q < 16 ? m.w0 : (q < 32 ? m.w1 : m.w2)) >> (4 * (q & 15))) & 0xFu;
}
// bits 4 q .. 4 q + 3
template <typename M>
__device__ __forceinline__ unsigned mask_nibble(const M &m, const int q)
{
    return (unsigned)(m >> (4 * q)) & 0xFu;
}
__device__ __forceinline__ unsigned mask_nibble(const Bits128 &m, const int q)
{
    return (unsigned)((q < 16 ? m.lo : m.hi) >> (4 * (q & 15))) & 0xFu;
}

// The rows of the change map of a wave's 64 pixels are 64 k contiguous bytes.  A lane storing its
// own row writes 4-byte pieces k bytes apart -- k / 4 store instructions that each touch every
// line of the span (measured on the streaming search: 0.24 ms of a 1.3 ms launch for 0.4 GB).
// Through a wave-private LDS image (16 k words) the same bytes leave as 16-byte pieces of
// consecutive lanes.  Needs all 64 pixels, k a multiple of 4 and a 16-byte aligned span.
template <typename MT>
__device__ __forceinline__ void store_change_rows_wave(uint8_t *wob, uint32_t *img, const int k,
                                                       const MT &mask, const int lane)
{
    const int kq = k >> 2;
    for (int q = 0; q < kq; ++q)
        img[lane * kq + q] = (mask_nibble(mask, q) * 0x00204081u) & 0x01010101u;     // bit i -> byte i
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same wave: LDS operations complete in order
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 *src = reinterpret_cast<const u4 *>(img);
    u4 *dst = reinterpret_cast<u4 *>(wob);
    for (int c = lane; c < 4 * k; c += 64) __builtin_nontemporal_store(src[c], dst + c);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the image may be written again behind this
}
__device__ __forceinline__ bool change_rows_wave_ok(const uint8_t *wob, const int k, const int wnp)
{
    return wnp == 64 && (k & 3) == 0 && ((uintptr_t)wob & 15) == 0;
}

template <typename T, typename MT>
__device__ __forceinline__ void screen_decide(const float x, const float m2, const bool sane,
                                              const DenseScreenEntry &c, const int t,
                                              MT &fbits, MT &ibits)
{
    const bool fires = sane && (x + m2 < c.a);
    const bool cant = sane && (x - m2 > c.b);
    mask_set(fbits, t, fires);
    mask_set(ibits, t, !(fires || cant));
}


// small cache of per-call tables: they depend only on (k, n_looks, alpha, dtype, p)
struct TabKey {
    int k, dtype, pol;
    uint32_t n;
    double alpha;
};
struct TabCacheEntry {
    TabKey key;
    std::vector<OmniTabEntry> tab;
};
std::vector<OmniTabEntry> get_table_impl(int k, uint32_t n_looks, double alpha, int dtype, int pol);

template <typename T>
static std::vector<OmniTabEntry> get_table(int k, uint32_t n_looks, double alpha, int pol)
{
    return get_table_impl(k, n_looks, alpha, sizeof(T) == 4 ? ND_AMD_F32 : ND_AMD_F64, pol);
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace nd_amd
