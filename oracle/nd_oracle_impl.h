/*
 * oracle/nd_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (see nd_oracle.c header).
 *
 * Type-generic body of the oracle, included twice by nd_oracle.c with
 *   REAL = float  / SFX(x) = x##_f32
 *   REAL = double / SFX(x) = x##_f64
 * mirroring Cython's fused type `floating` (nd/_change.pyx:8, nd/_filters.pyx:3).
 * Every place where the reference rounds to `floating` is an explicit (REAL)
 * cast or a REAL-typed variable here; everything else is double, exactly as in
 * the C that Cython generates (nd/_change.c:3501-3590, 6063-6091;
 * nd/_filters.c:3381-3692).
 */

/* ---- nd/_change.pyx:46-77  _z(ts, n) ----------------------------------- */
/* ts is a k x 4 strided view: element (i, v) at ts[v][i * st], columns
 * [C11, C12re, C12im, C22] (nd/_change.pyx:48, nd/change.py:66). */
static REAL SFX(orc_z)(const REAL *c11, const REAL *c12r, const REAL *c12i,
                       const REAL *c22, ptrdiff_t st, size_t k, unsigned int n)
{
    REAL p = 2;                          /* :51 dual pol */
    REAL c11sum = 0, c22sum = 0, c12rsum = 0, c12isum = 0;   /* :53 */
    REAL det_of_sum;
    double prod_of_dets = 1.0;           /* :55 DOUBLE */
    REAL rho;
    double logQ;
    REAL z;
    size_t i;

    for (i = 0; i < k; i++) {            /* :64-69 */
        REAL a = c11[(ptrdiff_t)i * st], b = c12r[(ptrdiff_t)i * st];
        REAL c = c12i[(ptrdiff_t)i * st], d = c22[(ptrdiff_t)i * st];
        /* determinant in `floating`, promoted to double for the product
         * (nd/_change.c:3559: float*float - (powf(.,2)+powf(.,2))) */
        REAL det = (a * d) - ((b * b) + (c * c));
        prod_of_dets = prod_of_dets * (double)det;
        c11sum = c11sum + a;
        c12rsum = c12rsum + b;
        c12isum = c12isum + c;
        c22sum = c22sum + d;
    }
    /* :72 */
    det_of_sum = (c11sum * c22sum) - ((c12rsum * c12rsum) + (c12isum * c12isum));
    /* :74, C types as in nd/_change.c:3580 */
    logQ = (double)n * ((((double)(p * (REAL)k)) * log((double)k)
                         + log(prod_of_dets))
                        - ((double)k * log((double)det_of_sum)));
    rho = (REAL)orc_rho((double)p, (double)k, (double)n);   /* :75, rounded to floating */
    z = (REAL)((-2.0 * (double)rho) * logQ);                /* :76 */
    return z;
}

/* ---- nd/_change.pyx:133-151  single_pixel_omnibus(ts, n) ---------------- */
static REAL SFX(orc_single_pixel_omnibus)(const REAL *c11, const REAL *c12r,
                                          const REAL *c12i, const REAL *c22,
                                          ptrdiff_t st, size_t k, unsigned int n,
                                          REAL *z_out)
{
    double p = 2;
    double f, rho, omega2;
    REAL z, P1, P2, result;

    f = orc_f(p, (double)k, (double)n);
    rho = orc_rho(p, (double)k, (double)n);
    omega2 = orc_omega2(p, (double)k, (double)n, rho);
    z = SFX(orc_z)(c11, c12r, c12i, c22, st, k, n);
    P1 = (REAL)orc_cdf_chisq_P((double)z, f);          /* :147 */
    P2 = (REAL)orc_cdf_chisq_P((double)z, f + 4.0);    /* :148 */
    /* :150; (P2 - P1) is a floating-floating subtraction (nd/_change.c:6089) */
    result = (REAL)((double)P1 + (omega2 * (double)(REAL)(P2 - P1)));
    if (z_out) *z_out = z;
    return result;
}

/* ---- nd/_change.pyx:224-257  single_pixel_change_detection -------------- */
static void SFX(orc_single_pixel_change_detection)(
    const REAL *c11, const REAL *c12r, const REAL *c12i, const REAL *c22,
    ptrdiff_t st, ptrdiff_t k, unsigned char *result, double alpha,
    unsigned int n, REAL *z0, REAL *p0)
{
    ptrdiff_t l, j, r = 0;
    REAL p_H0_l, p_H0_lj;
    int change;

    if (k < 1) return;
    l = 0;
    for (;;) {
        REAL zz;
        /* Test global hypothesis H0_l on ts[l:] (:238-240) */
        p_H0_l = SFX(orc_single_pixel_omnibus)(c11 + l * st, c12r + l * st,
                                               c12i + l * st, c22 + l * st,
                                               st, (size_t)(k - l), n, &zz);
        if (l == 0) {
            if (z0) *z0 = zz;
            if (p0) *p0 = p_H0_l;
        }
        change = ((double)p_H0_l > alpha);
        if (!change) break;
        /* marginal hypotheses (:246-254) */
        for (j = 2; j < k - l + 1; j++) {
            p_H0_lj = SFX(orc_single_pixel_omnibus)(c11 + l * st, c12r + l * st,
                                                    c12i + l * st, c22 + l * st,
                                                    st, (size_t)j, n, NULL);
            change = ((double)p_H0_lj > alpha);
            r = j - 1;
            if (change) {
                result[l + r] = 1;
                break;
            }
        }
        l = l + r;                        /* :255 */
        if (l >= k - 1) break;            /* :256 */
    }
}

/* ---- nd/_change.pyx:263-287  change_detection --------------------------- */
/* Planes + element strides instead of the (y,x,time,4) strided view that
 * nd/change.py:66-67 builds; same values, same order. `change` is (y,x,time)
 * C-order uint8, zeroed here like np.zeros at :275. */
int SFX(oracle_omnibus_c2)(const REAL *c11, const REAL *c12r, const REAL *c12i,
                           const REAL *c22, int64_t ny, int64_t nx, int64_t k,
                           int64_t sy, int64_t sx, int64_t st,
                           unsigned int n, double alpha,
                           unsigned char *change, REAL *z_out, REAL *p_out,
                           int nthreads)
{
    int64_t iy;
    if (ny < 0 || nx < 0 || k < 0) return -1;
    memset(change, 0, (size_t)ny * (size_t)nx * (size_t)k);
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 2) num_threads(nthreads)
#endif
    for (iy = 0; iy < ny; iy++) {
        int64_t ix;
        for (ix = 0; ix < nx; ix++) {
            ptrdiff_t off = (ptrdiff_t)(iy * sy + ix * sx);
            int64_t pix = iy * nx + ix;
            REAL zz = 0, pp = 0;
            if (k < 1) continue;
            SFX(orc_single_pixel_change_detection)(
                c11 + off, c12r + off, c12i + off, c22 + off, (ptrdiff_t)st,
                (ptrdiff_t)k, change + pix * k, alpha, n, &zz, &pp);
            if (z_out) z_out[pix] = zz;
            if (p_out) p_out[pix] = pp;
        }
    }
    return 0;
}

/* ---- generic-p restatement (p = 2: dual pol, p = 3: full pol) ------------------------------
 * The reference hard-codes p = 2 (nd/_change.pyx:51, 99, 135); `_f`, `_rho`, `_omega2` are already
 * written for any p (:20-39).  These functions apply the same formulas and the same rounding
 * points to p x p Hermitian covariance matrices given as p*p real planes:
 *   p = 2: [C11, C12re, C12im, C22]   (must equal the functions above bit for bit)
 *   p = 3: [C11, C22, C33, C12re, C12im, C13re, C13im, C23re, C23im]
 * det(C) for p = 3:  abc - a|z|^2 - b|y|^2 - c|x|^2 + 2 Re(x z conj(y)),  x = C12, y = C13, z = C23,
 * evaluated left to right in `floating`.  There is no reference implementation for p = 3:
 * parity for it is UNPINNED. */
static REAL SFX(orc_det_p)(const REAL *v, int pol)
{
    if (pol == 2) return (v[0] * v[3]) - ((v[1] * v[1]) + (v[2] * v[2]));
    {
        const REAL a = v[0], b = v[1], c = v[2];
        const REAL xr = v[3], xi = v[4], yr = v[5], yi = v[6], zr = v[7], zi = v[8];
        const REAL re = (((xr * zr) - (xi * zi)) * yr) + (((xr * zi) + (xi * zr)) * yi);
        return (((((a * b) * c) - (a * ((zr * zr) + (zi * zi)))) - (b * ((yr * yr) + (yi * yi))))
                - (c * ((xr * xr) + (xi * xi)))) + ((REAL)2 * re);
    }
}

static REAL SFX(orc_omnibus_p)(const REAL *const *pl, ptrdiff_t off, ptrdiff_t st, size_t k,
                               unsigned int n, int pol, REAL *z_out)
{
    const int nv = pol * pol;
    REAL p = (REAL)pol, sums[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, v[9];
    double prod_of_dets = 1.0, logQ, f, rho_d, omega2;
    REAL det_of_sum, rho, z, P1, P2, result;
    size_t i;
    int c;
    for (i = 0; i < k; i++) {
        for (c = 0; c < nv; c++) v[c] = pl[c][off + (ptrdiff_t)i * st];
        prod_of_dets = prod_of_dets * (double)SFX(orc_det_p)(v, pol);
        if (pol == 2) {        /* summation order of nd/_change.pyx:66-69 (irrelevant: independent sums) */
            for (c = 0; c < nv; c++) sums[c] = sums[c] + v[c];
        } else {
            for (c = 0; c < nv; c++) sums[c] = sums[c] + v[c];
        }
    }
    det_of_sum = SFX(orc_det_p)(sums, pol);
    logQ = (double)n * ((((double)(p * (REAL)k)) * log((double)k) + log(prod_of_dets))
                        - ((double)k * log((double)det_of_sum)));
    rho_d = orc_rho((double)pol, (double)k, (double)n);
    rho = (REAL)rho_d;
    z = (REAL)((-2.0 * (double)rho) * logQ);
    f = orc_f((double)pol, (double)k, (double)n);
    omega2 = orc_omega2((double)pol, (double)k, (double)n, rho_d);
    P1 = (REAL)orc_cdf_chisq_P((double)z, f);
    P2 = (REAL)orc_cdf_chisq_P((double)z, f + 4.0);
    result = (REAL)((double)P1 + (omega2 * (double)(REAL)(P2 - P1)));
    if (z_out) *z_out = z;
    return result;
}

/* nd/_change.pyx:224-287 for p x p matrices; planes share one set of element strides */
int SFX(oracle_omnibus_pol)(const REAL *const *planes, int pol, int64_t ny, int64_t nx, int64_t k,
                            int64_t sy, int64_t sx, int64_t st, unsigned int n, double alpha,
                            unsigned char *change, REAL *z_out, REAL *p_out, int nthreads)
{
    int64_t iy;
    if (ny < 0 || nx < 0 || k < 0 || (pol != 2 && pol != 3)) return -1;
    memset(change, 0, (size_t)ny * (size_t)nx * (size_t)k);
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 2) num_threads(nthreads)
#endif
    for (iy = 0; iy < ny; iy++) {
        int64_t ix;
        for (ix = 0; ix < nx; ix++) {
            const ptrdiff_t off = (ptrdiff_t)(iy * sy + ix * sx);
            unsigned char *result = change + (iy * nx + ix) * k;
            ptrdiff_t l = 0, j, r = 0;
            if (k < 1) continue;
            for (;;) {
                REAL zz, pg, pm;
                pg = SFX(orc_omnibus_p)(planes, off + l * (ptrdiff_t)st, (ptrdiff_t)st,
                                        (size_t)(k - l), n, pol, &zz);
                if (l == 0) {
                    if (z_out) z_out[iy * nx + ix] = zz;
                    if (p_out) p_out[iy * nx + ix] = pg;
                }
                if (!((double)pg > alpha)) break;
                for (j = 2; j < k - l + 1; j++) {
                    pm = SFX(orc_omnibus_p)(planes, off + l * (ptrdiff_t)st, (ptrdiff_t)st,
                                            (size_t)j, n, pol, NULL);
                    r = j - 1;
                    if ((double)pm > alpha) {
                        result[l + r] = 1;
                        break;
                    }
                }
                l = l + r;
                if (l >= k - 1) break;
            }
        }
    }
    return 0;
}

/* Marginal/global P of one series segment, for fixtures and unit pins. */
REAL SFX(oracle_single_pixel_omnibus)(const REAL *ts4 /* k x 4 C-order */,
                                      int64_t k, unsigned int n, REAL *z_out)
{
    return SFX(orc_single_pixel_omnibus)(ts4 + 0, ts4 + 1, ts4 + 2, ts4 + 3, 4,
                                         (size_t)k, n, z_out);
}

/* ---- nd/_filters.pyx:320-420  _pixelwise_nlmeans_3d --------------------- */
/* arr/out are (N0,N1,N2,V) views with element strides s[4] (the reference
 * takes arbitrary-stride memoryviews).  patch_mode: see the comment on the
 * patch-loop bounds below.  neff_policy: 0 = find_weight errors
 * are swallowed and the self weight becomes 0 (shipped Cython-0.29 C,
 * nd/_filters.c:2592+), 1 = stop and return 1 ("No solution", what a
 * Cython>=3 build of the same .pyx does). */
int SFX(oracle_nlmeans3d)(const REAL *arr, REAL *out, const int64_t N[3],
                          int64_t nvars, const int64_t s[4], const int64_t so[4],
                          const unsigned int r[3], const unsigned int f[3],
                          double sigma, double h, double n_eff, int neff_policy,
                          int patch_mode, int nthreads)
{
    const ptrdiff_t N0 = N[0], N1 = N[1], N2 = N[2];
    /* Patch-loop bounds, nd/_filters.pyx:374-376 `range(-f[i], f[i] + 1)` with
     * f an `unsigned int` memoryview and d a Py_ssize_t.  The C that Cython
     * emits (nd/_filters.c:3539-3553) starts the loop at `-(unsigned int)f[i]`,
     * i.e. 2^32 - f[i] once widened to the 64-bit Py_ssize_t, so on LP64 the
     * patch loops run ZERO times whenever any f[i] > 0 (dsquare stays 0 and
     * every neighbour gets weight exp(0) = 1) and exactly once when f == 0.
     * patch_mode 0 restates that literally (this is what the compiled
     * reference does, pinned by oracle/_ref); patch_mode 1 is the signed
     * range(-f, f+1) the source text intends. */
    const ptrdiff_t dlo0 = patch_mode ? -(ptrdiff_t)f[0] : (ptrdiff_t)(unsigned int)(0u - f[0]);
    const ptrdiff_t dlo1 = patch_mode ? -(ptrdiff_t)f[1] : (ptrdiff_t)(unsigned int)(0u - f[1]);
    const ptrdiff_t dlo2 = patch_mode ? -(ptrdiff_t)f[2] : (ptrdiff_t)(unsigned int)(0u - f[2]);
    /* :337 unsigned-int product assigned to `floating` */
    REAL dsq_norm = (REAL)(((unsigned int)nvars * (2 * f[0] + 1)) * (2 * f[1] + 1)
                           * (2 * f[2] + 1));
    int status = 0;
    ptrdiff_t p0;
    if (nvars > 64) return -2;
    if (nthreads < 1) nthreads = 1;

    /* :344-348 zero the output first */
    for (ptrdiff_t i0 = 0; i0 < N0; i0++)
        for (ptrdiff_t i1 = 0; i1 < N1; i1++)
            for (ptrdiff_t i2 = 0; i2 < N2; i2++)
                for (ptrdiff_t v = 0; v < nvars; v++)
                    out[i0 * so[0] + i1 * so[1] + i2 * so[2] + v * so[3]] = 0;

#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (p0 = 0; p0 < N0; p0++) {
        REAL weighted_sum[64];
        for (ptrdiff_t p1 = 0; p1 < N1; p1++) {
            for (ptrdiff_t p2 = 0; p2 < N2; p2++) {
                double total_weight = 0, total_sq_weight = 0, max_weight = 0;
                double weight, dsquare;
                int st_local;
#ifdef _OPENMP
#pragma omp atomic read
#endif
                st_local = status;
                if (st_local) continue;
                for (ptrdiff_t v = 0; v < nvars; v++) weighted_sum[v] = 0;

                for (ptrdiff_t q0 = p0 - (ptrdiff_t)r[0]; q0 < p0 + (ptrdiff_t)r[0] + 1; q0++)
                for (ptrdiff_t q1 = p1 - (ptrdiff_t)r[1]; q1 < p1 + (ptrdiff_t)r[1] + 1; q1++)
                for (ptrdiff_t q2 = p2 - (ptrdiff_t)r[2]; q2 < p2 + (ptrdiff_t)r[2] + 1; q2++) {
                    if (p0 == q0 && p1 == q1 && p2 == q2) continue;   /* :368 */
                    dsquare = 0;
                    for (ptrdiff_t d0 = dlo0; d0 < (ptrdiff_t)f[0] + 1; d0++)
                    for (ptrdiff_t d1 = dlo1; d1 < (ptrdiff_t)f[1] + 1; d1++)
                    for (ptrdiff_t d2 = dlo2; d2 < (ptrdiff_t)f[2] + 1; d2++) {
                        ptrdiff_t pa = orc_idx(p0 + d0, N0) * s[0] + orc_idx(p1 + d1, N1) * s[1]
                                     + orc_idx(p2 + d2, N2) * s[2];
                        ptrdiff_t qa = orc_idx(q0 + d0, N0) * s[0] + orc_idx(q1 + d1, N1) * s[1]
                                     + orc_idx(q2 + d2, N2) * s[2];
                        for (ptrdiff_t v = 0; v < nvars; v++) {
                            /* :377-386 floating diff, floating square, double sum */
                            REAL df = arr[pa + v * s[3]] - arr[qa + v * s[3]];
                            dsquare = dsquare + (double)(REAL)(df * df);
                        }
                    }
                    dsquare = dsquare / (double)dsq_norm;                 /* :388 */
                    {
                        double t = dsquare - (2.0 * (sigma * sigma));     /* :391 */
                        double m = (0 > t) ? 0.0 : t;                     /* max(t, 0) */
                        weight = exp((-m) / (h * h));
                    }
                    total_weight = total_weight + weight;
                    total_sq_weight = total_sq_weight + (weight * weight);
                    if (weight > max_weight) max_weight = weight;
                    {
                        ptrdiff_t qq = orc_idx(q0, N0) * s[0] + orc_idx(q1, N1) * s[1]
                                     + orc_idx(q2, N2) * s[2];
                        for (ptrdiff_t v = 0; v < nvars; v++)   /* :399-403 floating += double */
                            weighted_sum[v] = (REAL)((double)weighted_sum[v]
                                                     + (weight * (double)arr[qq + v * s[3]]));
                    }
                }

                if (n_eff < 0) {                       /* :406-411 */
                    if (max_weight == 0) max_weight = 1;
                    weight = max_weight;
                } else {                               /* :413 */
                    int err = 0;
                    weight = orc_find_weight(total_weight, total_sq_weight, n_eff, &err);
                    if (err) {
                        if (neff_policy == 1) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
                            status = 1;
                            continue;
                        }
                        weight = 0.0;
                    }
                }
                total_weight = total_weight + weight;  /* :417 */
                {
                    ptrdiff_t pp = p0 * s[0] + p1 * s[1] + p2 * s[2];
                    ptrdiff_t po = p0 * so[0] + p1 * so[1] + p2 * so[2];
                    for (ptrdiff_t v = 0; v < nvars; v++) {
                        weighted_sum[v] = (REAL)((double)weighted_sum[v]
                                                 + (weight * (double)arr[pp + v * s[3]]));
                        out[po + v * so[3]] = (REAL)((double)weighted_sum[v] / total_weight);
                    }
                }
            }
        }
    }
    return status;
}

/* ---- scipy.ndimage.convolve as called at nd/filters.py:256-267 ---------- */
/* in/out: up to 4-D (A0..A3) with element strides; footprint entries are the
 * non-zero taps of the already flipped kernel in C (row-major) order with
 * per-axis input offsets (see orc_build_footprint in nd_oracle.c).  Per
 * output: double tmp = 0; tmp += w * (double)in[...] in footprint order;
 * out = (REAL)tmp  (scipy NI_Correlate, CASE_CORRELATE_POINT). */
int SFX(oracle_correlate_fp)(const REAL *in, REAL *out, const int64_t A[4],
                             const int64_t si[4], const int64_t so[4],
                             int64_t ntaps, const int64_t *offs /* ntaps x 4 */,
                             const double *w, int nthreads)
{
    int64_t i0;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads) collapse(3)
#endif
    for (i0 = 0; i0 < A[0]; i0++)
        for (int64_t i1 = 0; i1 < A[1]; i1++)
            for (int64_t i2 = 0; i2 < A[2]; i2++)
                for (int64_t i3 = 0; i3 < A[3]; i3++) {
                    double tmp = 0.0;
                    for (int64_t t = 0; t < ntaps; t++) {
                        int64_t j0 = orc_reflect_half(i0 + offs[4 * t + 0], A[0]);
                        int64_t j1 = orc_reflect_half(i1 + offs[4 * t + 1], A[1]);
                        int64_t j2 = orc_reflect_half(i2 + offs[4 * t + 2], A[2]);
                        int64_t j3 = orc_reflect_half(i3 + offs[4 * t + 3], A[3]);
                        tmp += w[t] * (double)in[j0 * si[0] + j1 * si[1] + j2 * si[2] + j3 * si[3]];
                    }
                    out[i0 * so[0] + i1 * so[1] + i2 * so[2] + i3 * so[3]] = (REAL)tmp;
                }
    return 0;
}
