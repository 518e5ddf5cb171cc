// nd_amd/csrc/nlmeans.hip -- pixelwise non-local means in up to three filter dimensions plus a
// variable axis, for gfx950.  Replaces nd._filters._pixelwise_nlmeans_3d
// (nd/_filters.pyx:320-420, caller nd/filters.py:462).
//
// Generic kernel: one thread per output pixel p; the lane index runs along the axis with the
// smallest input stride so neighbouring lanes read neighbouring addresses.  The neighbour loop
// keeps the reference's order (q0 outer, q2 inner) because `weighted_sum` is rounded to
// `floating` after every addition (nd/_filters.c:3646).
//
// Arithmetic per the generated C (nd/_filters.c:3381-3692):
//   dsquare (double) += (T)((T)(a - b) * (T)(a - b));  dsquare /= (double)(T)dsq_norm;
//   weight = exp(-max(dsquare - 2 sigma^2, 0) / h^2)         (double)
//   weighted_sum[v] = (T)((double)weighted_sum[v] + weight * (double)arr[q, v])
//   out[p, v] = (T)((double)weighted_sum[v] / total_weight)
// Edges: whole-sample reflection of p+d, q+d and q (nd/_filters.pyx:15-40), applied in GLOBAL
// coordinates so a y-tile that carries its halo gives the untiled result.
//
// patch_mode 0 reproduces the compiled reference on LP64: the patch loops start at
// (Py_ssize_t)(unsigned)(-f[i]) = 2^32 - f[i] and therefore do not execute when f[i] > 0
// (nd/_filters.c:3539-3553); patch_mode 1 starts them at -f[i].
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace nd_amd {

template <typename T>
struct NlmArgs {
    const T *arr;
    T *out;
    int64_t N[3];          // tile shape
    int64_t G[3];          // global shape
    int64_t toff[3];       // tile offset in global coordinates
    int64_t clo[3], chi[3];   // written range (tile coordinates)
    int64_t si[4], so[4];
    int order[3];          // thread-index decomposition: order[0] fastest
    int64_t r[3];
    int64_t dlo[3], dhi[3];   // patch loop bounds [dlo, dhi)
    int nvars;
    T dsq_norm;
    double two_sigma2, h2, n_eff;
    int neff_policy;
    int32_t *status;
    int64_t total;
};

__device__ __forceinline__ int64_t nlm_idx(int64_t i, int64_t shape)
{
    // nd/_filters.pyx:34-40 EDGE_MODE_REFLECT
    if (i < 0) return -i;
    if (i >= shape) return 2 * shape - 2 - i;
    return i;
}

template <typename T, int VMAX>
__global__ void __launch_bounds__(256) nlmeans_generic_kernel(const NlmArgs<T> a)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.total) return;
    int64_t p[3];
    {
        int64_t rem = idx;
        const int o0 = a.order[0], o1 = a.order[1], o2 = a.order[2];
        const int64_t e0 = a.chi[o0] - a.clo[o0], e1 = a.chi[o1] - a.clo[o1];
        p[o0] = a.clo[o0] + rem % e0;
        rem /= e0;
        p[o1] = a.clo[o1] + rem % e1;
        rem /= e1;
        p[o2] = a.clo[o2] + rem;
    }
    // global coordinates of p
    const int64_t g0 = p[0] + a.toff[0], g1 = p[1] + a.toff[1], g2 = p[2] + a.toff[2];

    double total_weight = 0.0, total_sq_weight = 0.0, max_weight = 0.0;
    T ws[VMAX];
#pragma unroll
    for (int v = 0; v < VMAX; ++v) ws[v] = 0;

    for (int64_t q0 = g0 - a.r[0]; q0 < g0 + a.r[0] + 1; ++q0)
    for (int64_t q1 = g1 - a.r[1]; q1 < g1 + a.r[1] + 1; ++q1)
    for (int64_t q2 = g2 - a.r[2]; q2 < g2 + a.r[2] + 1; ++q2) {
        if (q0 == g0 && q1 == g1 && q2 == g2) continue;
        double dsquare = 0.0;
        for (int64_t d0 = a.dlo[0]; d0 < a.dhi[0]; ++d0)
        for (int64_t d1 = a.dlo[1]; d1 < a.dhi[1]; ++d1)
        for (int64_t d2 = a.dlo[2]; d2 < a.dhi[2]; ++d2) {
            const int64_t pa = (nlm_idx(g0 + d0, a.G[0]) - a.toff[0]) * a.si[0] +
                               (nlm_idx(g1 + d1, a.G[1]) - a.toff[1]) * a.si[1] +
                               (nlm_idx(g2 + d2, a.G[2]) - a.toff[2]) * a.si[2];
            const int64_t qa = (nlm_idx(q0 + d0, a.G[0]) - a.toff[0]) * a.si[0] +
                               (nlm_idx(q1 + d1, a.G[1]) - a.toff[1]) * a.si[1] +
                               (nlm_idx(q2 + d2, a.G[2]) - a.toff[2]) * a.si[2];
#pragma unroll
            for (int v = 0; v < VMAX; ++v) {
                if (v < a.nvars) {
                    const T df = a.arr[pa + v * a.si[3]] - a.arr[qa + v * a.si[3]];
                    const T sq = df * df;
                    dsquare = dsquare + (double)sq;
                }
            }
        }
        dsquare = dsquare / (double)a.dsq_norm;
        const double t = dsquare - a.two_sigma2;
        const double m = (0.0 > t) ? 0.0 : t;
        const double weight = exp((-m) / a.h2);
        total_weight = total_weight + weight;
        total_sq_weight = total_sq_weight + (weight * weight);
        if (weight > max_weight) max_weight = weight;
        const int64_t qq = (nlm_idx(q0, a.G[0]) - a.toff[0]) * a.si[0] +
                           (nlm_idx(q1, a.G[1]) - a.toff[1]) * a.si[1] +
                           (nlm_idx(q2, a.G[2]) - a.toff[2]) * a.si[2];
#pragma unroll
        for (int v = 0; v < VMAX; ++v) {
            if (v < a.nvars)
                ws[v] = (T)((double)ws[v] + (weight * (double)a.arr[qq + v * a.si[3]]));
        }
    }

    double weight;
    if (a.n_eff < 0) {
        if (max_weight == 0) max_weight = 1;
        weight = max_weight;
    } else {
        // find_weight, nd/_filters.pyx:299-314
        const double n = a.n_eff;
        bool err = (total_sq_weight == 0);
        if (!err) err = (n - 1.0) > ((total_weight * total_weight) / total_sq_weight);
        if (!err) err = ((n - 1.0) == 0);
        if (err) {
            if (a.neff_policy == 1) {
                if (a.status) atomicExch(a.status, 1);
                return;
            }
            weight = 0.0;
        } else {
            const double rt = sqrt(((((n * total_weight) * total_weight) -
                                     ((n * n) * total_sq_weight)) +
                                    (n * total_sq_weight)));
            weight = (total_weight + rt) / (n - 1.0);
        }
    }
    total_weight = total_weight + weight;
    const int64_t pp = p[0] * a.si[0] + p[1] * a.si[1] + p[2] * a.si[2];
    const int64_t po = p[0] * a.so[0] + p[1] * a.so[1] + p[2] * a.so[2];
#pragma unroll
    for (int v = 0; v < VMAX; ++v) {
        if (v < a.nvars) {
            ws[v] = (T)((double)ws[v] + (weight * (double)a.arr[pp + v * a.si[3]]));
            a.out[po + v * a.so[3]] = (T)((double)ws[v] / total_weight);
        }
    }
}

// =========================================================================================
// LDS-tiled forms for the common 2-D case: search and patch windows over axes 0 and 1 only
// (r[2] = f[2] = 0, every index of axis 2 filtered on its own), axis 1 contiguous in memory
// (planar [var][time][y][x] stacks viewed as (y, x, time, var)), float32.
//
// A block of 256 threads owns a 64 x TY tile of one axis-2 slice.  The tile plus its (r+f) halo is
// staged once per variable into LDS with the reference's whole-sample reflection (`_idx`) already
// applied in GLOBAL coordinates, so every later access is a plain LDS read at
// (row + dy, col + dx).
//
// patch_mode 0 with f > 0 (what the compiled reference computes): every neighbour weight is
//   exactly exp(-0) = 1, so the output is the running float32 sum of the window in the
//   reference's visiting order (rows outer, columns inner, centre last) divided by the count.
//   nlmeans_window_kernel: each thread owns 4 adjacent pixels of a row and reads each window row
//   once into registers.
// patch_mode 1 (true patch distances): nlmeans_patch_kernel.  Each thread owns one column of
//   TYW rows.  For a search offset (dy, dx) the squared difference summed over one patch ROW
//   (2F+1 columns, float32) is computed once per image row, and the patch sum is a sliding sum
//   of 2F+1 such rows down the column (double): O(2F+1) work per (pixel, offset) instead of
//   (2F+1)^2, no barrier inside the offset loop.  The neighbour weight is
//   exp(-max(d2/norm - 2 sigma^2, 0)/h^2) with d2 differing from the reference's double sum by
//   <= ~1e-6 relative (float32 row sums) -- inside the 1e-5 budget; weighted sums keep the
//   reference's order and per-step rounding to float32.
// =========================================================================================
struct NlmTiledArgs {
    const float *arr;
    float *out;
    int64_t N0, N1, N2;        // tile shape (axis 0 = rows, axis 1 = contiguous columns, axis 2 = slices)
    int64_t G0, G1;            // global extents of axes 0, 1
    int64_t off0, off1;        // tile offset in global coordinates
    int64_t clo0, chi0, clo1, chi1, clo2, chi2;   // written range
    int64_t si0, si2, si3;     // input strides (axis 1 stride is 1)
    int64_t so0, so2, so3;
    int r0, r1, f0, f1;
    int rz;                    // window radius along axis 2 (0: axis 2 is a plain slice axis)
    int64_t Gz, offz;          // global extent / tile offset of axis 2
    int nvars;
    int tiles_x, tiles_y;
    float dsq_norm;
    double two_sigma2, h2, n_eff;
    int neff_policy;
    int32_t *status;
};

__device__ __forceinline__ int nlm_reflect_i(int64_t i, int64_t shape)
{
    if (i < 0) return (int)(-i);
    if (i >= shape) return (int)(2 * shape - 2 - i);
    return (int)i;
}

// stage (rows x cols) of variable v, slice i2, top-left global coordinate (gy0, gx0), into lds
// (T = double: the float64 window kernel; `arr` / `out` of the arguments are then double arrays and
// every stride counts doubles)
template <typename T>
__device__ __forceinline__ void nlm_stage_t(const NlmTiledArgs &a, T *lds, int *ymap, int *xmap,
                                            int rows, int cols, int64_t gy0, int64_t gx0, int64_t i2,
                                            int v, int tid, bool *nonfinite = nullptr)
{
    for (int i = tid; i < rows + cols; i += 256) {
        // Positions that feed a written pixel reflect into the tile (checked on the host); the
        // clamps only keep the staging of never-used corner positions inside the allocation.
        if (i < rows) {
            int m = nlm_reflect_i(gy0 + i, a.G0) - (int)a.off0;
            m = m < 0 ? 0 : (m >= (int)a.N0 ? (int)a.N0 - 1 : m);
            ymap[i] = m;
        } else {
            int m = nlm_reflect_i(gx0 + (i - rows), a.G1) - (int)a.off1;
            m = m < 0 ? 0 : (m >= (int)a.N1 ? (int)a.N1 - 1 : m);
            xmap[i - rows] = m;
        }
    }
    __syncthreads();
    const T *base = reinterpret_cast<const T *>(a.arr) + i2 * a.si2 + (int64_t)v * a.si3;
    const int n = rows * cols;
    for (int e0 = 0; e0 < n; e0 += 256 * 8) {
        T buf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256 + tid;
            if (e < n) {
                const int rr = e / cols, cc = e - rr * cols;
                buf[u] = base[(int64_t)ymap[rr] * a.si0 + xmap[cc]];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256 + tid;
            if (e < n) {
                lds[e] = buf[u];
                if (nonfinite) *nonfinite = *nonfinite || !(fabs((double)buf[u]) < INFINITY);
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void nlm_stage(const NlmTiledArgs &a, float *lds, int *ymap, int *xmap,
                                          int rows, int cols, int64_t gy0, int64_t gx0, int64_t i2,
                                          int v, int tid, bool *nonfinite = nullptr)
{
    nlm_stage_t<float>(a, lds, ymap, xmap, rows, cols, gy0, gx0, i2, v, tid, nonfinite);
}

__device__ __forceinline__ double nlm_self_weight(double total_weight, double total_sq_weight,
                                                  double max_weight, double n_eff, int policy,
                                                  int32_t *status, bool *fail)
{
    *fail = false;
    if (n_eff < 0) {
        if (max_weight == 0) max_weight = 1;
        return max_weight;
    }
    const double n = n_eff;
    bool err = (total_sq_weight == 0);
    if (!err) err = (n - 1.0) > ((total_weight * total_weight) / total_sq_weight);
    if (!err) err = ((n - 1.0) == 0);
    if (err) {
        if (policy == 1) {
            if (status) atomicExch(status, 1);
            *fail = true;
        }
        return 0.0;
    }
    const double rt = sqrt(((((n * total_weight) * total_weight) - ((n * n) * total_sq_weight)) +
                            (n * total_sq_weight)));
    return (total_weight + rt) / (n - 1.0);
}

// find_weight's discriminant n W^2 - n^2 W2 + n W2 relative to its leading term.  Where the
// neighbours' effective sample size W^2 / W2 is within ~5 % of n_eff - 1 the two large terms
// cancel, and the float32 weights of the fast kernels (relative error ~2e-7) would come back
// amplified beyond the 1e-5 budget -- or decide "no solution" differently from the reference.
// Those pixels take the exact per-pixel path (double weights, the reference's order).  Also true
// for NaN and for the neighbourhood of the no-solution boundary.  FAR on the no-solution side
// (n_eff - 1 more than 5 % above W^2 / W2, e.g. an n_eff beyond the number of neighbours: every pixel of
// the raster) the verdict does not depend on 2e-7 of the weights either: find_weight raises, the
// reference's self weight is 0 (nlm_self_weight), and the fast path's sums stand -- up to round 6 all of
// those pixels were recomputed one by one (r = 3, n_eff = 50: 2.5 against 0.29 ms without n_eff).
__device__ __forceinline__ bool nlm_neff_ill(double tw, double tsq, double n)
{
    const double lead = (n * tw) * tw;
    const double disc = (lead - ((n * n) * tsq)) + (n * tsq);
    if (disc >= 0.05 * lead) return false;
    if ((tsq > 0.0) && ((n - 1.0) * tsq > 1.05 * (tw * tw))) return false;
    return true;
}

// ---- patch_mode 0, f > 0: uniform weights ------------------------------------------------
constexpr int kWinTX = 128, kWinTY = 32;     // 4 px per thread along x, 32 x 8 threads -> 128 x 8 ... x4 rows

// T = float, or double for float64 arrays: the running sums are `floating`, as in the reference
// (nd/_filters.pyx:336,341), the self term and the quotient double
template <typename T, int R1MAX>
__global__ void __launch_bounds__(256) nlmeans_window_kernel(const NlmTiledArgs a)
{
    extern __shared__ __align__(16) unsigned char nd_smem_n[];
    const int tid = threadIdx.x;
    const int r0 = a.r0, r1 = a.r1, rz = a.rz;
    const int cols = kWinTX + 2 * r1, rows = kWinTY + 2 * r0;
    const int psz = rows * cols, nz = 2 * rz + 1;
    T *lds = reinterpret_cast<T *>(nd_smem_n);                  // [nz][rows][cols]
    int *ymap = reinterpret_cast<int *>(lds + nz * psz);
    int *xmap = ymap + rows;

    int64_t b = blockIdx.x;
    const int tx = (int)(b % a.tiles_x);
    b /= a.tiles_x;
    const int ty = (int)(b % a.tiles_y);
    const int64_t i2 = a.clo2 + b / a.tiles_y;
    const int64_t y0 = a.clo0 + (int64_t)ty * kWinTY, x0 = a.clo1 + (int64_t)tx * kWinTX;

    const int lx = (tid % 32) * 4, ly = (tid / 32) * 4;          // 4 x 4 pixels per thread
    const double nq = (double)(nz * (2 * r0 + 1) * (2 * r1 + 1) - 1);
    bool fail;
    const double wself = nlm_self_weight(nq, nq, nq > 0 ? 1.0 : 0.0, a.n_eff, a.neff_policy,
                                         a.status, &fail);
    const double total = nq + wself;

    for (int v = 0; v < a.nvars; ++v) {
        __syncthreads();
        // the window planes along axis 2 (one plane when rz = 0), whole-sample reflection at the
        // global ends of that axis
        for (int dz = 0; dz < nz; ++dz) {
            int64_t zz = nlm_reflect_i(a.offz + i2 + dz - rz, a.Gz) - a.offz;
            zz = zz < 0 ? 0 : (zz >= a.N2 ? a.N2 - 1 : zz);
            nlm_stage_t<T>(a, lds + dz * psz, ymap, xmap, rows, cols, a.off0 + y0 - r0, a.off1 + x0 - r1,
                           zz, v, tid);
        }
#pragma unroll
        for (int py = 0; py < 4; ++py) {
            const int yy = ly + py;
            T ws[4] = {(T)0, (T)0, (T)0, (T)0};
            // the reference's visiting order: axis-2 offset outermost when it is a window axis
            // (it is then the FIRST filter dimension, see nlm_try_tiled), rows, columns innermost
            for (int dz = 0; dz < nz; ++dz) {
                for (int dy = 0; dy < 2 * r0 + 1; ++dy) {
                    const T *row = lds + dz * psz + (yy + dy) * cols + lx;
                    // window columns 0 .. 2 r1 for pixel 0, shifted by i for pixel i
                    T w[2 * R1MAX + 4];
#pragma unroll
                    for (int c = 0; c < 2 * R1MAX + 4; ++c)
                        if (c < 2 * r1 + 4) w[c] = row[c];
                    const bool centre_row = (dy == r0) && (dz == rz);
#pragma unroll
                    for (int dx = 0; dx < 2 * R1MAX + 1; ++dx) {
                        if (dx < 2 * r1 + 1 && !(centre_row && dx == r1)) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) ws[i] = ws[i] + w[dx + i];
                        }
                    }
                }
            }
            const int64_t y = y0 + yy;
            if (y < a.chi0 && !fail) {
                const T *crow = lds + rz * psz + (yy + r0) * cols + lx + r1;
                T *o = reinterpret_cast<T *>(a.out) + i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t x = x0 + lx + i;
                    if (x < a.chi1) {
                        // self term last (nd/_filters.pyx:417-420), weights are exactly 1 / wself
                        const T s = (T)((double)ws[i] + (wself * (double)crow[i]));
                        o[x] = (T)((double)s / total);
                    }
                }
            }
        }
    }
}

// ---- patch_mode 0, f > 0, rolling form -----------------------------------------------------
// One block owns a 128 x 32 tile of ONE variable and walks the third axis.  The planes the window
// needs along that axis live in an LDS ring of 2 rz + 1 slots (one slot when the axis is a plain
// slice axis), so every plane is staged once per tile instead of once per output slice, and the
// global loads of the next plane are in flight (held in registers) while the current slice is
// computed.  A thread owns 4 x 4 pixels and reads each staged row ONCE for all four of its output
// rows; two vertically adjacent outputs receive the same input element at the same step of their
// (reference-ordered) running sums, so they share one packed float32 addition.
typedef float nlm_f32x2 __attribute__((ext_vector_type(2)));
constexpr int kRollRowsPerWave = 13;          // (32 + 2 * 10) rows over 4 waves
constexpr int kRollR0Max = 10;

// res[i] = (float)((double)s[i] / total) for four sums, without the divisions: s * (1/total) is
// within 3 ulp of the correctly rounded quotient, so both round to the same float unless the
// product sits within a few ulp of a float rounding boundary or leaves the normal float range --
// then (any lane of the wave, any of the four: a wave-uniform and very rare branch) divide.  The
// branch must stay a branch: if-converted, every output pays a double-precision division.
__device__ __forceinline__ void nlm_div_rounded4(const float s[4], double total, double inv_total,
                                                 float res[4])
{
    double q[4];
    bool risky = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        q[i] = (double)s[i] * inv_total;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(q[i]);
        const unsigned lo = (unsigned)bits & 0x1fffffffu;
        const unsigned e = (unsigned)(bits >> 52) & 0x7ffu;
        risky |= (lo - 0x0ffffff8u) <= 16u || (e - (1023u - 125u)) > 251u;
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(risky) != 0ull, 0)) {
        asm volatile("" ::: "memory");          // not speculated, not if-converted
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = (double)s[i] / total;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) res[i] = (float)q[i];
}

// R0T >= 0: the row radius is the compile-time constant R0T (the row loop unrolls and every
// validity test folds away); R0T = -1: taken from the arguments.
template <int R1, int R0T>
__global__ void __launch_bounds__(256) nlmeans_window_roll_kernel(const NlmTiledArgs a)
{
    constexpr int NW = ((2 * R1 + 4) + 3) / 4 * 4;     // floats a thread reads from one staged row
    constexpr int COLSP = kWinTX - 4 + NW;             // row pitch (multiple of 4, >= COLS)
    extern __shared__ __align__(16) unsigned char nd_smem_r[];
    float *lds = reinterpret_cast<float *>(nd_smem_r); // [nzr][rows][COLSP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = R0T >= 0 ? R0T : a.r0, rz = a.rz;
    const int rows = kWinTY + 2 * r0, psz = rows * COLSP, nzr = 2 * rz + 1;

    int64_t b = blockIdx.x;
    const int tx = (int)(b % a.tiles_x);
    b /= a.tiles_x;
    const int ty = (int)(b % a.tiles_y);
    const int v = (int)(b / a.tiles_y);
    const int64_t y0 = a.clo0 + (int64_t)ty * kWinTY, x0 = a.clo1 + (int64_t)tx * kWinTX;
    const int lx = (tid % 32) * 4, ly = (tid / 32) * 4;

    const double nq = (double)(nzr * (2 * r0 + 1) * (2 * R1 + 1) - 1);
    bool fail;
    const double wself = nlm_self_weight(nq, nq, nq > 0 ? 1.0 : 0.0, a.n_eff, a.neff_policy,
                                         a.status, &fail);
    const double total = nq + wself;
    const double inv_total = 1.0 / total;

    // source column of each staged column this lane handles (whole-sample reflection in global
    // coordinates; the clamps only keep never-used corner positions inside the allocation)
    int xoff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int m = nlm_reflect_i(a.off1 + x0 - R1 + lane + 64 * j, a.G1) - (int)a.off1;
        xoff[j] = m < 0 ? 0 : (m >= (int)a.N1 ? (int)a.N1 - 1 : m);
    }
    const float *base = a.arr + (int64_t)v * a.si3;
    const int64_t gy0 = a.off0 + y0 - r0;

    float pre[kRollRowsPerWave][3];
    auto load_plane = [&](int64_t p) {
        const float *bp = base + p * a.si2;
#pragma unroll
        for (int u = 0; u < kRollRowsPerWave; ++u) {
            const int rr = wave + 4 * u;
            if (rr < rows) {
                int m = nlm_reflect_i(gy0 + rr, a.G0) - (int)a.off0;
                m = m < 0 ? 0 : (m >= (int)a.N0 ? (int)a.N0 - 1 : m);
                const float *rp = bp + (int64_t)m * a.si0;
                pre[u][0] = rp[xoff[0]];
                pre[u][1] = rp[xoff[1]];
                if (lane < 2 * R1) pre[u][2] = rp[xoff[2]];
            }
        }
    };
    auto store_plane = [&](int slot) {
        float *sp = lds + slot * psz;
#pragma unroll
        for (int u = 0; u < kRollRowsPerWave; ++u) {
            const int rr = wave + 4 * u;
            if (rr < rows) {
                float *rp = sp + rr * COLSP + lane;
                rp[0] = pre[u][0];
                rp[64] = pre[u][1];
                if (lane < 2 * R1) rp[128] = pre[u][2];
            }
        }
    };
    // local plane that stands for logical slice q of the window
    auto zmap = [&](int64_t q) {
        int64_t zz = (int64_t)nlm_reflect_i(a.offz + q, a.Gz) - a.offz;
        return zz < 0 ? (int64_t)0 : (zz >= a.N2 ? a.N2 - 1 : zz);
    };

    {
        const int64_t lo = a.clo2 - rz < 0 ? 0 : a.clo2 - rz;
        const int64_t hi = a.clo2 + rz > a.N2 - 1 ? a.N2 - 1 : a.clo2 + rz;
        for (int64_t p = lo; p <= hi; ++p) {
            load_plane(p);
            store_plane((int)(p % nzr));
        }
    }
    __syncthreads();

    const bool vec_ok = ((((uintptr_t)a.out) & 15) == 0) && ((a.so0 | a.so2 | a.so3 | x0) & 3) == 0;

    for (int64_t i2 = a.clo2; i2 < a.chi2; ++i2) {
        const int64_t nxt = i2 + rz + 1;
        const bool has_next = (nxt <= a.N2 - 1) && (i2 + 1 < a.chi2);
        if (has_next) load_plane(nxt);

        nlm_f32x2 acc[2][4];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[pr][i] = (nlm_f32x2){0.f, 0.f};

        // One staged row feeds every output row whose window holds it.  The two outputs of a pair
        // (rows 2 pr and 2 pr + 1 of the thread's patch) see input row ry as their window rows
        // du = ry - 2 pr and dl = du - 1; where both are inside the window the additions are
        // packed.  The window centre must not enter the sum: its lane adds +0.0 instead, which
        // leaves a running sum that started at +0.0 unchanged bit for bit.
        auto row_step = [&](const float *P, int ry, bool cplane, auto full) {
            constexpr bool FULL = decltype(full)::value;     // both outputs of both pairs hold this row
            float w[NW];
            const float4 *rp = reinterpret_cast<const float4 *>(P + ry * COLSP);
#pragma unroll
            for (int c = 0; c < NW / 4; ++c) {
                const float4 t = rp[c];
                w[4 * c + 0] = t.x;
                w[4 * c + 1] = t.y;
                w[4 * c + 2] = t.z;
                w[4 * c + 3] = t.w;
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int du = ry - 2 * pr, dl = du - 1;
                const bool vu = FULL || (du >= 0 && du <= 2 * r0), vl = FULL || (dl >= 0 && dl <= 2 * r0);
                const bool cu = cplane && du == r0, cl = cplane && dl == r0;
                if (vu && vl) {
#pragma unroll
                    for (int dx = 0; dx < 2 * R1 + 1; ++dx)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float x = w[dx + i];
                            const nlm_f32x2 t = {(dx == R1 && cu) ? 0.f : x, (dx == R1 && cl) ? 0.f : x};
                            acc[pr][i] = acc[pr][i] + t;
                        }
                } else if (vu) {
#pragma unroll
                    for (int dx = 0; dx < 2 * R1 + 1; ++dx)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[pr][i].x = acc[pr][i].x + ((dx == R1 && cu) ? 0.f : w[dx + i]);
                } else if (vl) {
#pragma unroll
                    for (int dx = 0; dx < 2 * R1 + 1; ++dx)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[pr][i].y = acc[pr][i].y + ((dx == R1 && cl) ? 0.f : w[dx + i]);
                }
            }
        };
        // the reference's visiting order: third-axis offset outermost when it is a window axis,
        // then rows, columns innermost; the centre is added last
        for (int dz = 0; dz < nzr; ++dz) {
            const float *P = lds + (int)(zmap(i2 + dz - rz) % nzr) * psz + ly * COLSP + lx;
            const bool cplane = (dz == rz);
            if (R0T >= 0) {
#pragma unroll
                for (int ry = 0; ry < 4 + 2 * (R0T >= 0 ? R0T : 0); ++ry)
                    row_step(P, ry, cplane, std::false_type{});
            } else {
                // rows 3 .. 2 r0 lie inside the window of all four output rows: no validity tests
                const int nrow = 4 + 2 * r0;
#pragma unroll
                for (int ry = 0; ry < 3; ++ry) row_step(P, ry, cplane, std::false_type{});
#pragma unroll 2
                for (int ry = 3; ry <= 2 * r0; ++ry) row_step(P, ry, cplane, std::true_type{});
#pragma unroll 1
                for (int ry = (2 * r0 + 1 > 3 ? 2 * r0 + 1 : 3); ry < nrow; ++ry)
                    row_step(P, ry, cplane, std::false_type{});
            }
        }

        if (!fail) {
            const float *C = lds + (int)(i2 % nzr) * psz + (ly + r0) * COLSP + lx + R1;
#pragma unroll
            for (int py = 0; py < 4; ++py) {
                const int64_t y = y0 + ly + py;
                if (y < a.chi0) {
                    float res[4], sv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float ws = (py & 1) ? acc[py >> 1][i].y : acc[py >> 1][i].x;
                        // self term last (nd/_filters.pyx:417-420), weights are exactly 1 / wself
                        sv[i] = (float)((double)ws + (wself * (double)C[py * COLSP + i]));
                    }
                    nlm_div_rounded4(sv, total, inv_total, res);
                    float *o = a.out + i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x0 + lx;
                    if (vec_ok && x0 + lx + 3 < a.chi1) {
                        *reinterpret_cast<float4 *>(o) = make_float4(res[0], res[1], res[2], res[3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (x0 + lx + i < a.chi1) o[i] = res[i];
                    }
                }
            }
        }
        __syncthreads();                       // every wave is done with the oldest plane
        if (has_next) store_plane((int)(nxt % nzr));
        __syncthreads();
    }
}

// ---- patch_mode 0, f > 0, window of THREE planes along the third axis, streaming form ---------
// The tutorial's filter (NLMeansFilter(dims=('time','y','x'), r=(1,3,3), f=1)) has a window of
// three dates.  The ring kernel above keeps those three planes in LDS (62 KB per block at R = 3:
// two blocks per CU) and reads every staged row once per output slice.  Here ONE plane is consumed
// at a time and serves the three output slices whose windows hold it: the plane of step s is the
// dz = 0 plane of slice q + 1, the dz = 1 plane of slice q and the dz = 2 plane of slice q - 1
// (q = first slice - 1 + s; the staged sequence zmap(f - 1), zmap(f), ..., zmap(l + 1) holds for
// every slice its three planes in the reference's visiting order, whole-sample reflection at the
// ends of the axis included, because slice i's planes (dz = 1, 2) are slice i + 1's planes
// (dz = 0, 1)).  A thread keeps three sets of 4 x 4 running sums:
//   ab[py][i] = (.x: slice q - 1, finishing; .y: slice q) -- the two receive the SAME element at
//               the same step of their sums (only the window centre of slice q is replaced by
//               +0.0), so every addition is packed, with no unpaired border rows;
//   c[pr][i]  = slice q + 1, starting, packed over vertically adjacent outputs as in the ring form.
// Staging is LDS-DMA (buffer_load_dword ... lds: memory -> LDS without passing through registers,
// per-lane source offsets carry the reflection) into the second of two plane slots while the first
// is consumed; one barrier per step.  41 KB of LDS and < 128 VGPRs: three blocks per CU, and a
// third of the ring form's LDS reads.
typedef __attribute__((address_space(3))) float nlm_lds_f32;

// OY = output rows per thread (4: 256 threads per 128 x 32 tile; 2: 512 threads, half the running
// sums per thread and twice the waves per CU), WPE = waves per SIMD the register budget is set for.
template <int R, int OY, int WPE>
__global__ void __launch_bounds__(32 * (kWinTY / OY), WPE) nlmeans_window_stream3_kernel(const NlmTiledArgs a)
{
    constexpr int NT = 32 * (kWinTY / OY), NWAVES = NT / 64;
    constexpr int NW = ((2 * R + 4) + 3) / 4 * 4;      // floats a thread reads from one staged row
    constexpr int COLSP = kWinTX - 4 + NW;             // row pitch (multiple of 4)
    constexpr int ROWS = kWinTY + 2 * R;
    constexpr int PSZ = ROWS * COLSP;
    constexpr int NPRE = (ROWS + NWAVES - 1) / NWAVES; // staged rows per wave
    extern __shared__ __align__(16) unsigned char nd_smem_s3[];
    float *slots = reinterpret_cast<float *>(nd_smem_s3);         // [2][PSZ]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int64_t b = blockIdx.x;
    const int tx = (int)(b % a.tiles_x);
    b /= a.tiles_x;
    const int ty = (int)(b % a.tiles_y);
    const int v = (int)(b / a.tiles_y);
    const int64_t y0 = a.clo0 + (int64_t)ty * kWinTY, x0 = a.clo1 + (int64_t)tx * kWinTX;
    const int lx = (tid % 32) * 4, ly = (tid / 32) * OY;

    const double nq = (double)(3 * (2 * R + 1) * (2 * R + 1) - 1);
    bool fail;
    const double wself = nlm_self_weight(nq, nq, 1.0, a.n_eff, a.neff_policy, a.status, &fail);
    const double total = nq + wself;
    const double inv_total = 1.0 / total;

    // A descriptor per plane, a wave-uniform row offset (scalar) and one 32-bit byte offset per
    // lane and staged column.  (Row offsets inside a plane fit 31 bits: checked on the host.)
    int xoff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int m = nlm_reflect_i(a.off1 + x0 - R + lane + 64 * j, a.G1) - (int)a.off1;
        xoff[j] = 4 * (m < 0 ? 0 : (m >= (int)a.N1 ? (int)a.N1 - 1 : m));
    }
    int roff[NPRE];
    {
        const int64_t gy0 = a.off0 + y0 - R;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            int m = nlm_reflect_i(gy0 + wave + NWAVES * u, a.G0) - (int)a.off0;
            m = m < 0 ? 0 : (m >= (int)a.N0 ? (int)a.N0 - 1 : m);
            roff[u] = __builtin_amdgcn_readfirstlane(m * (int)a.si0 * 4);
        }
    }
    const float *base = a.arr + (int64_t)v * a.si3;

    auto stage = [&](int64_t p, int slot) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + p * a.si2), 0,
                                                            0x7fffffff, 0x00020000);
        nlm_lds_f32 *dst = (nlm_lds_f32 *)(slots + slot * PSZ + wave * COLSP);
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            if (wave + NWAVES * u < ROWS) {
                nlm_lds_f32 *rp = dst + NWAVES * u * COLSP;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, rp, 4, xoff[0], roff[u], 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, rp + 64, 4, xoff[1], roff[u], 0, 0);
                if (lane < 2 * R) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, rp + 128, 4, xoff[2], roff[u], 0, 0);
            }
        }
    };
    auto zmap = [&](int64_t q) {
        int64_t zz = (int64_t)nlm_reflect_i(a.offz + q, a.Gz) - a.offz;
        return zz < 0 ? (int64_t)0 : (zz >= a.N2 ? a.N2 - 1 : zz);
    };

    const int64_t first = a.clo2, last = a.chi2 - 1;
    const int nsteps = (int)(last - first) + 3;
    stage(zmap(first - 1), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const bool vec_ok = ((((uintptr_t)a.out) & 15) == 0) && ((a.so0 | a.so2 | a.so3 | x0) & 3) == 0;

    nlm_f32x2 ab[OY][4], c[OY / 2][4];
    float cen[OY][4];              // window-centre values of slice q - 1 (read while its plane was staged)
#pragma unroll
    for (int py = 0; py < OY; ++py)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ab[py][i] = (nlm_f32x2){0.f, 0.f};
            cen[py][i] = 0.f;
        }

    // (round 5) Two things the first and last steps do not need.  (i) At the ends of the axis the staged
    // sequence repeats a plane (whole-sample reflection: -1 -> 0, N -> N - 1): the plane is already in the
    // current slot, so nothing is transferred and the slots do not swap -- 2 of k + 2 planes of reads.
    // (ii) Step 0 only starts slice `first` (the sums `c`): the slices q - 1 and q of `ab` do not exist, and
    // the last two steps start no slice: their `c` sums would be thrown away -- 784 of the 16 016 vector
    // additions a thread issues per tile at 24 dates.
    int slot = 0;
    int64_t pcur = zmap(first - 1);
    for (int s = 0; s < nsteps; ++s) {
        const int64_t q = first - 1 + s;
        const bool has_next = s + 1 < nsteps;
        const int64_t pnext = has_next ? zmap(q + 1) : pcur;
        const bool same = pnext == pcur;
        if (has_next && !same) stage(pnext, slot ^ 1);
        const bool do_ab = s >= 1, do_c = s + 2 < nsteps;
        const float *P = slots + slot * PSZ + ly * COLSP + lx;
#pragma unroll
        for (int pr = 0; pr < OY / 2; ++pr)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[pr][i] = (nlm_f32x2){0.f, 0.f};

#pragma unroll
        for (int ry = 0; ry < OY + 2 * R; ++ry) {
            float w[NW];
            const float4 *rp = reinterpret_cast<const float4 *>(P + ry * COLSP);
#pragma unroll
            for (int cc = 0; cc < NW / 4; ++cc) {
                const float4 t = rp[cc];
                w[4 * cc + 0] = t.x;
                w[4 * cc + 1] = t.y;
                w[4 * cc + 2] = t.z;
                w[4 * cc + 3] = t.w;
            }
            // slices q - 1 and q: this row is window row d of output row py
            if (do_ab) {
#pragma unroll
            for (int py = 0; py < OY; ++py) {
                const int d = ry - py;
                if (d >= 0 && d <= 2 * R) {
#pragma unroll
                    for (int dx = 0; dx < 2 * R + 1; ++dx)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float x = w[dx + i];
                            const nlm_f32x2 t = {x, (d == R && dx == R) ? 0.f : x};
                            ab[py][i] = ab[py][i] + t;
                        }
                }
            }
            }
            // slice q + 1 (no centre in its first plane): vertical pairs
            if (do_c) {
#pragma unroll
            for (int pr = 0; pr < OY / 2; ++pr) {
                const int du = ry - 2 * pr, dl = du - 1;
                const bool vu = du >= 0 && du <= 2 * R, vl = dl >= 0 && dl <= 2 * R;
                if (vu && vl) {
#pragma unroll
                    for (int dx = 0; dx < 2 * R + 1; ++dx)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float x = w[dx + i];
                            c[pr][i] = c[pr][i] + (nlm_f32x2){x, x};
                        }
                } else if (vu) {
#pragma unroll
                    for (int dx = 0; dx < 2 * R + 1; ++dx)
#pragma unroll
                        for (int i = 0; i < 4; ++i) c[pr][i].x = c[pr][i].x + w[dx + i];
                } else if (vl) {
#pragma unroll
                    for (int dx = 0; dx < 2 * R + 1; ++dx)
#pragma unroll
                        for (int i = 0; i < 4; ++i) c[pr][i].y = c[pr][i].y + w[dx + i];
                }
            }
            }
            // Keep the rows apart.  The sums of slice q + 1 are only needed when the roles move on,
            // and left alone the optimiser sinks all their additions below the output code -- with
            // the values of every staged row alive until then (254 VGPRs at R = 3).  The empty asm
            // statements pin the running sums to this point of the program.
#pragma unroll
            for (int pr = 0; pr < OY / 2; ++pr)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(c[pr][i]));
#pragma unroll
            for (int py = 0; py < OY; ++py)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(ab[py][i]));
            __builtin_amdgcn_sched_barrier(0);
        }

        // slice q - 1 is complete: self term last (nd/_filters.pyx:417-420), then the mean
        if (s >= 2 && !fail) {
            const int64_t i2 = q - 1;
#pragma unroll
            for (int py = 0; py < OY; ++py) {
                const int64_t y = y0 + ly + py;
                if (y < a.chi0) {
                    float res[4], sv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        sv[i] = (float)((double)ab[py][i].x + (wself * (double)cen[py][i]));
                    nlm_div_rounded4(sv, total, inv_total, res);
                    float *o = a.out + i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x0 + lx;
                    if (vec_ok && x0 + lx + 3 < a.chi1) {
                        *reinterpret_cast<float4 *>(o) = make_float4(res[0], res[1], res[2], res[3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (x0 + lx + i < a.chi1) o[i] = res[i];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the centre values of slice q for its last step, then the roles move on (the compiler
        // barrier keeps the values of the row reads above from being carried down here instead of
        // being read again: 4 OY registers for most of the step)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int py = 0; py < OY; ++py) {
            const float *cr = P + (py + R) * COLSP + R;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                cen[py][i] = cr[i];
                ab[py][i].x = ab[py][i].y;
                ab[py][i].y = (py & 1) ? c[py >> 1][i].y : c[py >> 1][i].x;
            }
        }
        // the next plane has landed (this wave's transfers) and every wave is done with this one
        if (!same) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            slot ^= 1;
        }
        pcur = pnext;
    }
}

static bool launch_stream3(const NlmTiledArgs &a, int64_t nb, hipStream_t stream)
{
    auto lds = [](int R) {
        const size_t nw = ((2 * (size_t)R + 4) + 3) / 4 * 4;
        return 2 * (size_t)(kWinTY + 2 * R) * (kWinTX - 4 + nw) * sizeof(float);
    };
    // 512 threads x (2 x 4) outputs by default; the 256-thread form (4 x 4 outputs per thread, half
    // the waves) measures the same to 1 % -- the kernel is bound by VALU issue, not by latency
    // (profiles/r02_nlmeans_window_pmc.txt)
    static const bool oy4 = getenv("ND_AMD_NLM_S3_OY4") != nullptr;
    const dim3 grid((unsigned)nb);
#define ND_S3(RR)                                                                                          \
    if (oy4)                                                                                               \
        hipLaunchKernelGGL((nlmeans_window_stream3_kernel<RR, 4, 3>), grid, dim3(256), lds(RR), stream, a); \
    else                                                                                                   \
        hipLaunchKernelGGL((nlmeans_window_stream3_kernel<RR, 2, 4>), grid, dim3(512), lds(RR), stream, a); \
    return true
    switch (a.r1) {
    case 1: ND_S3(1);
    case 2: ND_S3(2);
    case 3: ND_S3(3);
    case 4: ND_S3(4);
    case 5: ND_S3(5);
    default: return false;
    }
#undef ND_S3
}

// Dynamic LDS beyond the default 64 KB limit (up to kBigLds of the CU's 160 KB: one block per CU
// then -- still tens of times faster than the generic kernel the request would otherwise fall to)
constexpr size_t kBigLds = 150 * 1024;
#define ND_LAUNCH_LDS(KERNEL, GRID, BLOCK, LDS, STREAM, ARGS)                                          \
    do {                                                                                              \
        if ((LDS) > 64 * 1024)                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL),                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS));        \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, ARGS);                                   \
    } while (0)

template <int R1>
static void launch_roll(const NlmTiledArgs &a, int64_t nb, size_t lds, hipStream_t stream)
{
    // square windows up to 11 x 11 get the fully unrolled row loop
    if (R1 >= 1 && R1 <= 5 && a.r0 == R1) {
        ND_LAUNCH_LDS((nlmeans_window_roll_kernel<R1, (R1 >= 1 && R1 <= 5) ? R1 : -1>),
                      dim3((unsigned)nb), dim3(256), lds, stream, a);
    } else {
        ND_LAUNCH_LDS((nlmeans_window_roll_kernel<R1, -1>), dim3((unsigned)nb), dim3(256), lds, stream, a);
    }
}

static bool launch_roll_r1(const NlmTiledArgs &a, int64_t nb, size_t lds, hipStream_t stream)
{
    switch (a.r1) {
    case 0: launch_roll<0>(a, nb, lds, stream); return true;
    case 1: launch_roll<1>(a, nb, lds, stream); return true;
    case 2: launch_roll<2>(a, nb, lds, stream); return true;
    case 3: launch_roll<3>(a, nb, lds, stream); return true;
    case 4: launch_roll<4>(a, nb, lds, stream); return true;
    case 5: launch_roll<5>(a, nb, lds, stream); return true;
    case 6: launch_roll<6>(a, nb, lds, stream); return true;
    case 7: launch_roll<7>(a, nb, lds, stream); return true;
    case 8: launch_roll<8>(a, nb, lds, stream); return true;
    case 9: launch_roll<9>(a, nb, lds, stream); return true;
    case 10: launch_roll<10>(a, nb, lds, stream); return true;
    default: return false;
    }
}

// ---- patch_mode 1: sliding patch-row sums ---------------------------------------------------
// A NaN inside the pixel's OWN patch makes every patch distance NaN (it enters each of them), hence
// every weight, both weight sums and the weighted sums: the reference's result is NaN for every
// variable (nd/_filters.pyx:386-420; the self weight is 1, NaN or -- when n_eff - 1 == 0 -- the error
// case, which is left to the full evaluation).  Nodata regions therefore cost one look at the patch
// instead of the whole search window in double precision.
template <int F, int V>
__device__ __forceinline__ bool nlm_own_patch_nan(const float *lds, int plane, int cols, int py, int px)
{
    bool nan = false;
#ifdef ND_NO_OWN_NAN
    return false;
#endif
    for (int i = -F; i <= F; ++i)
        for (int j = -F; j <= F; ++j)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float c = lds[v * plane + (py + i) * cols + px + j];
                nan = nan || (c != c);
            }
    return nan;
}

template <int F, int V, int TYW, bool NEFF>
__global__ void __launch_bounds__(256) nlmeans_patch_kernel(const NlmTiledArgs a)
{
    extern __shared__ __align__(16) unsigned char nd_smem_n[];
    constexpr int TX = 64, TY = 4 * TYW;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r0 = a.r0, r1 = a.r1;
    const int halo0 = r0 + F, halo1 = r1 + F;
    const int cols = TX + 2 * halo1, rows = TY + 2 * halo0;
    float *lds = reinterpret_cast<float *>(nd_smem_n);               // [V][rows][cols]
    int *ymap = reinterpret_cast<int *>(lds + V * rows * cols);
    int *xmap = ymap + rows;

    int64_t b = blockIdx.x;
    const int tx = (int)(b % a.tiles_x);
    b /= a.tiles_x;
    const int ty = (int)(b % a.tiles_y);
    const int64_t i2 = a.clo2 + b / a.tiles_y;
    const int64_t y0 = a.clo0 + (int64_t)ty * TY, x0 = a.clo1 + (int64_t)tx * TX;

#pragma unroll
    for (int v = 0; v < V; ++v) {
        __syncthreads();
        nlm_stage(a, lds + v * rows * cols, ymap, xmap, rows, cols, a.off0 + y0 - halo0,
                  a.off1 + x0 - halo1, i2, v, tid);
    }

    // this thread: column `lane`, rows wave*TYW .. wave*TYW + TYW - 1 of the tile
    const int cx = halo1 + lane;                 // LDS column of the pixel
    const int cy0 = halo0 + wave * TYW;          // LDS row of the first pixel
    double tw[TYW], tsq[NEFF ? TYW : 1];
    float wmax[TYW];
    float ws[TYW][V];
#pragma unroll
    for (int p = 0; p < TYW; ++p) {
        tw[p] = 0.0;
        if (NEFF) tsq[p] = 0.0;
        wmax[p] = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) ws[p][v] = 0.f;
    }
    // the reference divides in double; reciprocals in float32 differ by ~1e-7 relative
    const float inv_norm = (float)(1.0 / (double)a.dsq_norm);
    const float neg_inv_h2 = (float)(-1.0 / a.h2);
    const float two_sigma2 = (float)a.two_sigma2;

    // Non-finite row sums (NaN / inf data) must not enter the sliding sum -- NaN - NaN and inf - inf
    // would stay in it for every row further down: they count as 0 there, and the pixels whose
    // 2F + 1 rows hold one are evaluated by the exact path below
    unsigned nonfinite_mask = 0, nan_mask = 0;
    for (int dy = -r0; dy <= r0; ++dy) {
        for (int dx = -r1; dx <= r1; ++dx) {
            if (dy == 0 && dx == 0) continue;
            double S = 0.0;
            double H[TYW + 2 * F];
            int bad_age = 1 << 20, nan_age = 1 << 20;      // rows since the last non-finite / NaN row sum
#pragma unroll
            for (int s = 0; s < TYW + 2 * F; ++s) {
                // patch-row sum of squared differences at image row (cy0 - F + s)
                const int rr = cy0 - F + s;
                float hs = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float *pa = lds + v * rows * cols + rr * cols + cx - F;
                    const float *qa = pa + dy * cols + dx;
#pragma unroll
                    for (int j = 0; j < 2 * F + 1; ++j) {
                        const float df = pa[j] - qa[j];
                        hs = hs + df * df;
                    }
                }
                const bool hbad = !(hs < INFINITY);
                H[s] = hbad ? 0.0 : (double)hs;
                bad_age = hbad ? 0 : bad_age + 1;
                nan_age = (hs != hs) ? 0 : nan_age + 1;
                S = S + H[s];
                if (s >= 2 * F + 1) S = S - H[s - 2 * F - 1];
                if (s >= 2 * F) {
                    const int p = s - 2 * F;
                    if (bad_age <= 2 * F) nonfinite_mask |= 1u << p;
                    if (nan_age <= 2 * F) nan_mask |= 1u << p;
                    // weight argument in float32 from here on (the patch-row sums already are):
                    // relative error ~1e-7 of d2, far inside the budget
                    const float d2 = (float)S * inv_norm;
                    const float t = d2 - two_sigma2;
                    const float m = (0.f > t) ? 0.f : t;
                    const float w = __expf(m * neg_inv_h2);
                    tw[p] = tw[p] + (double)w;
                    if (NEFF) tsq[p] = tsq[p] + (double)w * (double)w;
                    wmax[p] = w > wmax[p] ? w : wmax[p];
                    const int qr = cy0 + p + dy, qc = cx + dx;
                    // (float)((double)ws + (double)w * (double)a) -- the product of two floats is
                    // exact in double, so the single-rounding fused multiply-add gives the same
                    // value.  Measured (A/B on one device): the fma form is 17 % faster for F <= 1
                    // and 27 % slower for F = 3 (scheduling), so each uses its faster spelling.
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const float av = lds[v * rows * cols + qr * cols + qc];
                        if (F <= 1)
                            ws[p][v] = __fmaf_rn(w, av, ws[p][v]);
                        else
                            ws[p][v] = (float)((double)ws[p][v] + ((double)w * (double)av));
                    }
                }
            }
        }
    }

    const int64_t x = x0 + lane;
    // Pixels whose largest float32 weight is (nearly) zero: the reference's double weights are
    // still non-zero there and decide the self weight and the (denormal) sums, so those pixels
    // are recomputed below exactly as the reference does.  Elsewhere weights under 1e-30 of the
    // maximum cannot move the result by 1e-8.
    unsigned exact_mask = 0;
#pragma unroll
    for (int p = 0; p < TYW; ++p) {
        const int64_t y = y0 + wave * TYW + p;
        if (y < a.chi0 && x < a.chi1) {
            if (((nan_mask >> p) & 1u) && !(a.n_eff >= 0 && (a.n_eff - 1.0) == 0)) {
                // a NaN patch distance: NaN weight, NaN sums, NaN result (nd/_filters.pyx:386-420)
#pragma unroll
                for (int v = 0; v < V; ++v)
                    a.out[i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x] = __builtin_nanf("");
                continue;
            }
            if (!(wmax[p] >= 1e-30f) || ((nonfinite_mask >> p) & 1u) ||
                (NEFF && nlm_neff_ill(tw[p], NEFF ? tsq[p] : 0.0, a.n_eff))) {
                exact_mask |= 1u << p;
                continue;
            }
            bool fail;
            const double wself = nlm_self_weight(tw[p], NEFF ? tsq[p] : 0.0, (double)wmax[p],
                                                 a.n_eff, a.neff_policy, a.status, &fail);
            if (!fail) {
                const double total = tw[p] + wself;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float c = lds[v * rows * cols + (cy0 + p) * cols + cx];
                    const float sfin = (float)((double)ws[p][v] + (wself * (double)c));
                    a.out[i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x] =
                        (float)((double)sfin / total);
                }
            }
        }
    }
    if (__any(exact_mask != 0u)) {
        for (int p = 0; p < TYW; ++p) {
            if (!((exact_mask >> p) & 1u)) continue;
            // nd/_filters.pyx:363-420 for this one pixel, from the staged tile
            const int py = cy0 + p;
            if (!(a.n_eff >= 0 && (a.n_eff - 1.0) == 0) && nlm_own_patch_nan<F, V>(lds, rows * cols, cols, py, cx)) {
                const int64_t y = y0 + wave * TYW + p;
#pragma unroll
                for (int v = 0; v < V; ++v)
                    a.out[i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x] = __builtin_nanf("");
                continue;
            }
            double t_w = 0.0, t_sq = 0.0, m_w = 0.0;
            float wsum[V];
#pragma unroll
            for (int v = 0; v < V; ++v) wsum[v] = 0.f;
            for (int dy = -r0; dy <= r0; ++dy)
                for (int dx = -r1; dx <= r1; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    double dsq = 0.0;
                    for (int i = -F; i <= F; ++i)
                        for (int j = -F; j <= F; ++j)
#pragma unroll
                            for (int v = 0; v < V; ++v) {
                                const float *base = lds + v * rows * cols;
                                const float df = base[(py + i) * cols + cx + j] -
                                                 base[(py + dy + i) * cols + cx + dx + j];
                                const float sq = df * df;
                                dsq = dsq + (double)sq;
                            }
                    dsq = dsq / (double)a.dsq_norm;
                    const double t = dsq - a.two_sigma2;
                    const double m = (0.0 > t) ? 0.0 : t;
                    const double w = exp((-m) / a.h2);
                    t_w = t_w + w;
                    t_sq = t_sq + (w * w);
                    if (w > m_w) m_w = w;
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        wsum[v] = (float)((double)wsum[v] +
                                          (w * (double)lds[v * rows * cols + (py + dy) * cols + cx + dx]));
                }
            bool fail;
            const double wself = nlm_self_weight(t_w, t_sq, m_w, a.n_eff, a.neff_policy, a.status, &fail);
            if (!fail) {
                const double total = t_w + wself;
                const int64_t y = y0 + wave * TYW + p;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float c = lds[v * rows * cols + py * cols + cx];
                    const float sfin = (float)((double)wsum[v] + (wself * (double)c));
                    a.out[i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x] =
                        (float)((double)sfin / total);
                }
            }
        }
    }
}

// ---- patch_mode 1, cross-lane form: two columns per lane ------------------------------------
// nlmeans_patch_kernel gives every lane its own copy of each patch-row sum: 2 (2F+1) single-word
// LDS reads and as many subtract / multiply-add per row, although seven adjacent lanes share six
// of the seven squared differences.  Here a lane owns TWO adjacent columns and computes the
// squared difference of each exactly once (packed float32 arithmetic: one instruction for both
// columns); the 2F+1-wide row sums are then assembled across lanes with wave shifts (DPP
// wave_shr / wave_shl, no LDS):  with e(2l), e(2l+1) the lane's squares and P(l) their sum,
//   F = 1:  hs(2l) = e(2l-1) + P(l)                  hs(2l+1) = P(l) + e(2l+2)
//   F = 2:  hs(2l) = P(l-1) + P(l) + e(2l+2)         hs(2l+1) = e(2l-1) + P(l) + P(l+1)
//   F = 3:  hs(2l) = e(2l-3) + P(l-1) + P(l) + P(l+1)   hs(2l+1) = P(l-1) + P(l) + P(l+1) + e(2l+4)
// i.e. 2 to 6 shifted additions per row and lane-pair instead of 2 x 14 LDS reads + 2 x 14
// operations.  The first / last HL lanes of a wave only feed their neighbours (a wave yields
// 2 (64 - 2 HL) columns).  Down the column the patch sum is the sum of the last 2F+1 row sums
// (re-added every step from the ring: no drift), the weight and the weighted sums are packed
// float32 as well; total weights are summed in float32 per search row and in double across rows.
// Differences from the reference's double arithmetic stay <= ~1e-6 relative (tests: 1e-5).
// Pixels whose weights all vanish in float32 are recomputed exactly as in nlmeans_patch_kernel.
typedef float f2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dpp_from_prev(float x)      // lane i <- lane i - 1 (lane 0 <- 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_from_next(float x)      // lane i <- lane i + 1 (lane 63 <- 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xF, 0xF, true));
}

#ifndef ND_PATCH2_TYW1
#define ND_PATCH2_TYW1 8
#endif
template <int V>
struct Patch2Rows {
    static constexpr int TYW = (V == 1) ? ND_PATCH2_TYW1 : 4;      // rows per thread
};
template <int F>
struct Patch2Geom {
    static constexpr int HL = (F == 3) ? 2 : (F >= 1 ? 1 : 0);     // feeder lanes on each side
    static constexpr int TX = 2 * (64 - 2 * HL);                   // columns a wave yields
};

// max of two float pairs, one v_max_f32 per component (gfx950 has no packed max; fmaxf() costs
// three: it canonicalises both operands first).  A NaN operand is dropped (IEEE maxNum), which is
// what the callers want: NaNs are tracked separately (`ssum`).
__device__ __forceinline__ f2_t pk_max(const f2_t a, const f2_t b)
{
    f2_t d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d.x) : "v"(a.x), "v"(b.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(d.y) : "v"(a.y), "v"(b.y));
    return d;
}

template <int F>
__device__ __forceinline__ f2_t patch2_row_sums(const f2_t e)
{
    if (F == 0) return e;
    const float P = e.x + e.y;
    f2_t hs;
    if (F == 1) {
        hs.x = dpp_from_prev(e.y) + P;
        hs.y = P + dpp_from_next(e.x);
    } else if (F == 2) {
        hs.x = (dpp_from_prev(P) + P) + dpp_from_next(e.x);
        hs.y = (dpp_from_prev(e.y) + P) + dpp_from_next(P);
    } else {
        // columns 2l-3 .. 2l+3 and 2l-2 .. 2l+4 out of three-column pieces, every step one addition
        // with a wave-shifted operand (v_add_f32_dpp; fetching e(2l-3) and e(2l+4) by themselves
        // takes two shifts each):  T(l) = e(2l-1) + P(l),  U(l) = P(l) + e(2l+2)
        const float T = dpp_from_prev(e.y) + P;
        const float U = P + dpp_from_next(e.x);
        hs.x = (dpp_from_prev(T) + P) + dpp_from_next(P);
        hs.y = (dpp_from_prev(P) + P) + dpp_from_next(U);
    }
    return hs;
}

// MC: the margin as a compile-time constant (0: from r1).  With the row pitch of the tile known at
// compile time the rows of a column become immediate offsets of the LDS instructions instead of
// one address register per row, each advanced by a vector addition at every search offset.
constexpr int kPatch2Margin = 16;
// Waves per SIMD the register allocation is held to.  One variable: 3 (168 registers; the few
// values that no longer fit are spilled outside the offset loop) -- the cross-lane row sums are
// chains of dependent additions with DPP wait states, and a third wave fills them: config 3
// 53 -> 46 ms; 4 waves gain nothing more.  Several variables: 2 (3 spills inside the loop:
// 4 variables 34 -> 59 ms); unbounded, the n_eff forms pass 256 registers and run one wave per
// SIMD (58 -> 93 ms).  Measured with tools/exp_nlm_cc.py.
#ifndef ND_P2_W1
#define ND_P2_W1 3
#endif
#ifndef ND_P2_WV
#define ND_P2_WV 2
#endif
template <int F, int V, int TYW, bool NEFF, int MC>
__global__ void __launch_bounds__(256, V == 1 ? ND_P2_W1 : ND_P2_WV) nlmeans_patch2_kernel(const NlmTiledArgs a)
{
    extern __shared__ __align__(16) unsigned char nd_smem_n[];
    constexpr int HL = Patch2Geom<F>::HL, TX = Patch2Geom<F>::TX, TY = 4 * TYW;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r0 = a.r0, r1 = a.r1;
    const int halo0 = r0 + F;
    // left / right margin: even, >= r1 + F and >= r1 + 2 HL
    const int M = MC ? MC : r1 + 2 * HL + (r1 & 1);
    const int cols = TX + 2 * M, rows = TY + 2 * halo0;
    float *lds = reinterpret_cast<float *>(nd_smem_n);               // [V][rows][cols]
    int *ymap = reinterpret_cast<int *>(lds + V * rows * cols);
    int *xmap = ymap + rows;

    int64_t b = blockIdx.x;
    const int tx = (int)(b % a.tiles_x);
    b /= a.tiles_x;
    const int ty = (int)(b % a.tiles_y);
    const int64_t i2 = a.clo2 + b / a.tiles_y;
    // Lane pairs sit on GLOBAL (even, odd) columns whatever the tile or the written range: the two
    // columns of a pair add their row sums in different groupings, and a pixel's value must not
    // depend on how the raster was cut (multi-GPU row blocks, halo tiles).
    const int64_t xs = a.clo1 - ((a.off1 + a.clo1) & 1);
    const int64_t y0 = a.clo0 + (int64_t)ty * TY, x0 = xs + (int64_t)tx * TX;

    bool nonfinite = false;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        __syncthreads();
        nlm_stage(a, lds + v * rows * cols, ymap, xmap, rows, cols, a.off0 + y0 - halo0,
                  a.off1 + x0 - M, i2, v, tid, &nonfinite);
    }
    // A patch sum can only be NaN when the tile holds a NaN or an infinity (squares of finite
    // differences are finite or +inf, and +inf gives the weight 0 like the reference's exp(-inf)):
    // tiles of finite values -- all but nodata areas -- skip the NaN bookkeeping of the loop below
    const bool tile_nonfinite = __syncthreads_or(nonfinite ? 1 : 0) != 0;

    // this lane: tile columns 2 (lane - HL), + 1 (feeder lanes: columns outside the tile), rows
    // wave*TYW .. wave*TYW + TYW - 1; LDS column of the pair's first element is even
    const int cx = M + 2 * (lane - HL);
    const int cy0 = halo0 + wave * TYW;
    double tw[TYW][2], tsq[NEFF ? TYW : 1][2];
    f2_t wmax[TYW], ws[TYW][V];
#pragma unroll
    for (int p = 0; p < TYW; ++p) {
        tw[p][0] = tw[p][1] = 0.0;
        if (NEFF) tsq[p][0] = tsq[p][1] = 0.0;
        wmax[p] = (f2_t){0.f, 0.f};
#pragma unroll
        for (int v = 0; v < V; ++v) ws[p][v] = (f2_t){0.f, 0.f};
    }
    // the reference divides in double; reciprocals in float32 differ by ~1e-7 relative
    const float exp2_scale = (float)(-1.4426950408889634 / a.h2);      // w = 2^(m * exp2_scale)
    const float c1 = (float)((double)exp2_scale / (double)a.dsq_norm);
    const float c0 = (float)(-(double)a.two_sigma2 * (double)exp2_scale);
    const f2_t exp_c1 = (f2_t){c1, c1}, exp_c0 = (f2_t){c0, c0};
    const f2_t zero2 = (f2_t){0.f, 0.f};
    f2_t ssum[TYW];
#pragma unroll
    for (int p = 0; p < TYW; ++p) ssum[p] = zero2;

    auto search = [&](auto track_nan) {
    for (int dy = -r0; dy <= r0; ++dy) {
        f2_t twf[TYW];                    // NEFF = false: float32 partial sums of this search row
#pragma unroll
        for (int p = 0; p < TYW; ++p) twf[p] = (f2_t){0.f, 0.f};
        // three search offsets per trip: their LDS reads then share one address register per row
        // (the offsets along x are immediates); trips beyond r1 and the centre are skipped
        for (int dx0 = -r1; dx0 <= r1; dx0 += 3)
#pragma unroll
        for (int du = 0; du < 3; ++du) {
            const int dx = dx0 + du;
            if (dx > r1 || (dy == 0 && dx == 0)) continue;
            f2_t H[2 * F + 1], Q[2 * F + 1];      // rings: row sums, and sums of two adjacent rows
#pragma unroll
            for (int s = 0; s < TYW + 2 * F; ++s) {
                // squared differences of this lane's two columns at image row (cy0 - F + s)
                const int rr = cy0 - F + s;
                f2_t e = (f2_t){0.f, 0.f};
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float *pa = lds + v * rows * cols + rr * cols + cx;
                    const float *qa = pa + dy * cols + dx;
                    const f2_t pv = *reinterpret_cast<const f2_t *>(pa);      // 8-byte aligned
                    const f2_t qv = (f2_t){qa[0], qa[1]};
                    const f2_t df = pv - qv;
                    e = __builtin_elementwise_fma(df, df, e);
                }
                constexpr int NR = 2 * F + 1;
                H[s % NR] = patch2_row_sums<F>(e);
                if (F >= 2 && s >= 1) Q[s % NR] = H[(s - 1) % NR] + H[s % NR];      // rows s-1, s
                if (s >= 2 * F) {
                    const int p = s - 2 * F;
                    // the patch sum: rows p .. p + 2F, always grouped the same way relative to the
                    // pixel's own row (pairs from the top, last row alone) -- the ring position of a
                    // row depends on where the tile starts, the order of the additions must not
                    f2_t S;
                    if (F == 0) {
                        S = H[s % NR];
                    } else if (F == 1) {
                        S = (H[(s - 2) % NR] + H[(s - 1) % NR]) + H[s % NR];
                    } else if (F == 2) {
                        S = (Q[(s - 3) % NR] + Q[(s - 1) % NR]) + H[s % NR];
                    } else {
                        S = ((Q[(s - 5) % NR] + Q[(s - 3) % NR]) + Q[(s - 1) % NR]) + H[s % NR];
                    }
                    if (decltype(track_nan)::value)
                        ssum[p] = ssum[p] + S;      // NaN anywhere in the pixel's patch sums sticks here
                    // w = exp(-max(d2 / norm - 2 sigma^2, 0) / h^2) = min(2^(S c1 + c0), 1): one fused
                    // multiply-add for the exponent, and the hardware exponential with the clamp
                    // modifier (result clamped to [0, 1]; a NaN would be dropped here, `ssum`
                    // remembers it for the exact path below).  The s_nop covers the wait state a
                    // transcendental result needs before an ordinary vector instruction reads it,
                    // which the compiler does not insert around inline assembly.
                    const f2_t ma = __builtin_elementwise_fma(S, exp_c1, exp_c0);
                    f2_t w;
                    asm("v_exp_f32_e64 %0, %2 clamp\n\tv_exp_f32_e64 %1, %3 clamp\n\ts_nop 0"
                        : "=&v"(w.x), "=v"(w.y) : "v"(ma.x), "v"(ma.y));
                    if (NEFF) {
                        // the self weight solves a quadratic whose discriminant cancels: the two
                        // weight sums it is made of are kept in double, like the reference's
                        tw[p][0] = tw[p][0] + (double)w.x;
                        tw[p][1] = tw[p][1] + (double)w.y;
                        tsq[p][0] = tsq[p][0] + (double)w.x * (double)w.x;
                        tsq[p][1] = tsq[p][1] + (double)w.y * (double)w.y;
                    } else {
                        twf[p] = twf[p] + w;
                    }
                    wmax[p] = pk_max(wmax[p], w);
                    const int qr = cy0 + p + dy, qc = cx + dx;
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const float *ap = lds + v * rows * cols + qr * cols + qc;
                        const f2_t av = (f2_t){ap[0], ap[1]};
                        // (float)((double)ws + (double)w * (double)a): the product of two floats is
                        // exact in double, so the fused multiply-add rounds to the same value
                        ws[p][v] = __builtin_elementwise_fma(w, av, ws[p][v]);
                    }
                }
            }
        }
        if (!NEFF) {
#pragma unroll
            for (int p = 0; p < TYW; ++p) {
                tw[p][0] = tw[p][0] + (double)twf[p].x;
                tw[p][1] = tw[p][1] + (double)twf[p].y;
            }
        }
    }
    };
    if (tile_nonfinite)
        search(std::true_type{});
    else
        search(std::false_type{});

    const bool feeder = (lane < HL) || (lane >= 64 - HL);
    // (with n_eff - 1 == 0 the reference takes its error branch instead: left to the exact path)
    const bool nan_is_final = !(a.n_eff >= 0 && (a.n_eff - 1.0) == 0);
    // Pixels whose largest float32 weight is (nearly) zero: the reference's double weights are
    // still non-zero there and decide the self weight and the (denormal) sums, so those pixels
    // are recomputed below exactly as the reference does.
    unsigned exact_mask = 0;              // bit 2p + c
#pragma unroll
    for (int p = 0; p < TYW; ++p) {
        const int64_t y = y0 + wave * TYW + p;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int64_t x = x0 + 2 * (lane - HL) + c;
            if (feeder || !(y < a.chi0 && x < a.chi1 && x >= a.clo1)) continue;
            const float wm = c ? wmax[p].y : wmax[p].x;
            const float sp = c ? ssum[p].y : ssum[p].x;
            if (!(sp == sp) && nan_is_final) {
                // some patch distance of this pixel is NaN: so are that weight, both weight sums and
                // every weighted sum in the reference (nd/_filters.pyx:386-420) -- the result is NaN
#pragma unroll
                for (int v = 0; v < V; ++v)
                    a.out[i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x] = __builtin_nanf("");
                continue;
            }
            if (!(wm >= 1e-30f) || !(sp == sp) || (NEFF && nlm_neff_ill(tw[p][c], NEFF ? tsq[p][c] : 0.0, a.n_eff))) {
                exact_mask |= 1u << (2 * p + c);
                continue;
            }
            bool fail;
            const double wself = nlm_self_weight(tw[p][c], NEFF ? tsq[p][c] : 0.0, (double)wm,
                                                 a.n_eff, a.neff_policy, a.status, &fail);
            if (!fail) {
                const double total = tw[p][c] + wself;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float cv = lds[v * rows * cols + (cy0 + p) * cols + cx + c];
                    const float wsv = c ? ws[p][v].y : ws[p][v].x;
                    const float sfin = (float)((double)wsv + (wself * (double)cv));
                    a.out[i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x] =
                        (float)((double)sfin / total);
                }
            }
        }
    }
    if (__any(exact_mask != 0u)) {
        for (int pc = 0; pc < 2 * TYW; ++pc) {
            if (!((exact_mask >> pc) & 1u)) continue;
            // nd/_filters.pyx:363-420 for this one pixel, from the staged tile
            const int p = pc >> 1, c = pc & 1;
            const int py = cy0 + p, pxc = cx + c;
            if (!(a.n_eff >= 0 && (a.n_eff - 1.0) == 0) && nlm_own_patch_nan<F, V>(lds, rows * cols, cols, py, pxc)) {
                const int64_t y = y0 + wave * TYW + p;
                const int64_t x = x0 + 2 * (lane - HL) + c;
#pragma unroll
                for (int v = 0; v < V; ++v)
                    a.out[i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x] = __builtin_nanf("");
                continue;
            }
            double t_w = 0.0, t_sq = 0.0, m_w = 0.0;
            float wsum[V];
#pragma unroll
            for (int v = 0; v < V; ++v) wsum[v] = 0.f;
            for (int dy = -r0; dy <= r0; ++dy)
                for (int dx = -r1; dx <= r1; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    double dsq = 0.0;
                    for (int i = -F; i <= F; ++i)
                        for (int j = -F; j <= F; ++j)
#pragma unroll
                            for (int v = 0; v < V; ++v) {
                                const float *base = lds + v * rows * cols;
                                const float df = base[(py + i) * cols + pxc + j] -
                                                 base[(py + dy + i) * cols + pxc + dx + j];
                                const float sq = df * df;
                                dsq = dsq + (double)sq;
                            }
                    dsq = dsq / (double)a.dsq_norm;
                    const double t = dsq - a.two_sigma2;
                    const double m = (0.0 > t) ? 0.0 : t;
                    const double w = exp((-m) / a.h2);
                    t_w = t_w + w;
                    t_sq = t_sq + (w * w);
                    if (w > m_w) m_w = w;
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        wsum[v] = (float)((double)wsum[v] +
                                          (w * (double)lds[v * rows * cols + (py + dy) * cols + pxc + dx]));
                }
            bool fail;
            const double wself = nlm_self_weight(t_w, t_sq, m_w, a.n_eff, a.neff_policy, a.status, &fail);
            if (!fail) {
                const double total = t_w + wself;
                const int64_t y = y0 + wave * TYW + p;
                const int64_t x = x0 + 2 * (lane - HL) + c;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float cv = lds[v * rows * cols + py * cols + pxc];
                    const float sfin = (float)((double)wsum[v] + (wself * (double)cv));
                    a.out[i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x] =
                        (float)((double)sfin / total);
                }
            }
        }
    }
}

template <int F, int V>
static void launch_patch2(const NlmTiledArgs &a, int64_t nslices, size_t lds, hipStream_t stream)
{
    constexpr int TYW = Patch2Rows<V>::TYW;
    const int64_t nb = (int64_t)a.tiles_x * a.tiles_y * nslices;
    const bool mc = a.r1 + 2 * Patch2Geom<F>::HL + (a.r1 & 1) <= kPatch2Margin;    // as the host sized the tile
    if (a.n_eff >= 0) {
        if (mc) {
            ND_LAUNCH_LDS((nlmeans_patch2_kernel<F, V, TYW, true, kPatch2Margin>), dim3((unsigned)nb), dim3(256), lds, stream, a);
        } else {
            ND_LAUNCH_LDS((nlmeans_patch2_kernel<F, V, TYW, true, 0>), dim3((unsigned)nb), dim3(256), lds, stream, a);
        }
    } else {
        if (mc) {
            ND_LAUNCH_LDS((nlmeans_patch2_kernel<F, V, TYW, false, kPatch2Margin>), dim3((unsigned)nb), dim3(256), lds, stream, a);
        } else {
            ND_LAUNCH_LDS((nlmeans_patch2_kernel<F, V, TYW, false, 0>), dim3((unsigned)nb), dim3(256), lds, stream, a);
        }
    }
}

template <int F>
static bool launch_patch2_v(const NlmTiledArgs &a, int64_t nslices, size_t lds, hipStream_t stream)
{
    switch (a.nvars) {
    case 1: launch_patch2<F, 1>(a, nslices, lds, stream); return true;
    case 2: launch_patch2<F, 2>(a, nslices, lds, stream); return true;
    case 3: launch_patch2<F, 3>(a, nslices, lds, stream); return true;
    case 4: launch_patch2<F, 4>(a, nslices, lds, stream); return true;
    }
    return false;
}

template <int F, int V>
static void launch_patch(const NlmTiledArgs &a, int64_t nslices, size_t lds, hipStream_t stream)
{
    constexpr int TYW = (V == 1) ? 16 : 8;
    const int64_t nb = (int64_t)a.tiles_x * a.tiles_y * nslices;
    if (a.n_eff >= 0) {
        ND_LAUNCH_LDS((nlmeans_patch_kernel<F, V, TYW, true>), dim3((unsigned)nb), dim3(256), lds, stream, a);
    } else {
        ND_LAUNCH_LDS((nlmeans_patch_kernel<F, V, TYW, false>), dim3((unsigned)nb), dim3(256), lds, stream, a);
    }
}

template <int F>
static bool launch_patch_v(const NlmTiledArgs &a, int64_t nslices, size_t lds, hipStream_t stream)
{
    switch (a.nvars) {
    case 1: launch_patch<F, 1>(a, nslices, lds, stream); return true;
    case 2: launch_patch<F, 2>(a, nslices, lds, stream); return true;
    case 3: launch_patch<F, 3>(a, nslices, lds, stream); return true;
    case 4: launch_patch<F, 4>(a, nslices, lds, stream); return true;
    }
    return false;
}

// Try a tiled form; 1 = launched, 0 = not applicable.
// ---- patch_mode 1 with a window along the third axis (round 6) ---------------------------------
// The signed patch distances of a search with an extent along time -- NLMeansFilter(dims=('time', 'y',
// 'x'), r=(1, 3, 3), f=1), the tutorial's filter (examples/tutorial_s1.ipynb cell 11), in the mode the
// source text intends (nd/_filters.pyx:363-403) -- ran only in the per-pixel kernel: 146 neighbours x
// 108 squared differences per pixel, every operand fetched from memory (4 variables, 6 x 1024 x 2048:
// 152 ms).  Here a block owns a 16-row x (64 - 2 F)-column tile of ONE plane z of the (z, y, x) array:
//   * the 2 (rz + FZ) + 1 planes around z are staged into LDS for every variable, whole-sample
//     reflection applied in GLOBAL coordinates on all three axes;
//   * a thread owns one column and TYW = 4 rows (+ 2 F rows of patch halo) and keeps ITS OWN values of
//     the 2 FZ + 1 patch planes in registers for the whole search ((TYW + 2 F) (2 FZ + 1) V values);
//   * per search offset (dz, dy, dx), visited in the reference's order (z outermost): for every row the
//     squared difference summed over the patch planes and the variables (one LDS read, one subtraction,
//     one multiply-add per value), the 2 F + 1-wide row sum across lanes (wave shifts, no LDS), the
//     2 F + 1 row sums of a pixel's patch added down the column: O((2 FZ + 1) V) work per pixel and
//     offset instead of (2 FZ + 1) (2 F + 1)^2 V;
//   * weights and weighted sums as in nlmeans_patch_kernel (float32 weight from a float32 distance:
//     <= ~1e-6 relative, inside the 1e-5 budget of this mode; the weighted sums keep the reference's
//     order and per-step rounding to float32); pixels with a non-finite value within reach, with all
//     float32 weights vanishing, or with an ill-conditioned n_eff equation are recomputed one by one in
//     double, in the reference's order, from the staged planes.
template <int F, int FZ, int V>
__device__ __forceinline__ bool nlm_own_patch_nan3(const float *lds, int pz0, int pstride, int vstride, int cols,
                                                   int py, int px)
{
    bool nan = false;
    for (int z = 0; z < 2 * FZ + 1; ++z)
        for (int i = -F; i <= F; ++i)
            for (int j = -F; j <= F; ++j)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float c = lds[(pz0 + z) * pstride + v * vstride + (py + i) * cols + px + j];
                    nan = nan || (c != c);
                }
    return nan;
}

template <int F>
__device__ __forceinline__ float nlm_row_sum_lanes(const float e)
{
    if (F == 0) return e;
    float acc = e, up = e, dn = e;
#pragma unroll
    for (int j = 0; j < F; ++j) {
        up = dpp_from_prev(up);
        dn = dpp_from_next(dn);
        acc = (acc + up) + dn;
    }
    return acc;
}

constexpr int kPatch3TYW = 4;
constexpr float kPatch3Redo = 1e-2f;      // a pixel whose largest float32 weight is below this: the wave searches again in double

// RMAX: the search radius along rows and columns the tile is laid out for (r0, r1 <= RMAX): the pitches of
// the staged planes are compile-time constants then, and the (2 FZ + 1) V (TYW + 2 F) reads of a search
// offset are immediate offsets of three address registers instead of one address computation each.
// TYW: rows per thread (4: a 16-row tile; 2: an 8-row tile for windows whose planes would not fit otherwise --
// seven planes of four variables)
template <int F, int FZ, int V, bool NEFF, int RMAX, int TYW>
__global__ void __launch_bounds__(256) nlmeans_patch3_kernel(const NlmTiledArgs a)
{
    extern __shared__ __align__(16) unsigned char nd_smem_p3[];
    constexpr int TY = 4 * TYW, TXO = 64 - 2 * F, NPP = 2 * FZ + 1, NS = TYW + 2 * F;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r0 = a.r0, r1 = a.r1, rz = a.rz;
    constexpr int cols = 64 + 2 * RMAX, rows = TY + 2 * (RMAX + F);
    const int np = 2 * (rz + FZ) + 1;                       // staged planes
    constexpr int vstride = rows * cols, pstride = V * vstride;
    float *lds = reinterpret_cast<float *>(nd_smem_p3);              // [plane][V][rows][cols]
    int *ymap = reinterpret_cast<int *>(lds + np * pstride);
    int *xmap = ymap + rows;

    int64_t b = blockIdx.x;
    const int tx = (int)(b % a.tiles_x);
    b /= a.tiles_x;
    const int ty = (int)(b % a.tiles_y);
    const int64_t i2 = a.clo2 + b / a.tiles_y;
    const int64_t y0 = a.clo0 + (int64_t)ty * TY, x0 = a.clo1 + (int64_t)tx * TXO;

    bool nonfinite_here = false;
    for (int pz = 0; pz < np; ++pz) {
        int zi = nlm_reflect_i(a.offz + i2 - (rz + FZ) + pz, a.Gz) - (int)a.offz;
        zi = zi < 0 ? 0 : (zi >= (int)a.N2 ? (int)a.N2 - 1 : zi);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            __syncthreads();
            nlm_stage(a, lds + pz * pstride + v * vstride, ymap, xmap, rows, cols, a.off0 + y0 - (RMAX + F),
                      a.off1 + x0 - F - RMAX, zi, v, tid, &nonfinite_here);
        }
    }
    // (any non-finite value in the block's planes: every pixel takes the exact path -- rare, and it spares
    //  the fast path the bookkeeping of which row sums a NaN or an infinity has entered)
    const bool block_nonfinite = __syncthreads_or(nonfinite_here ? 1 : 0) != 0;

    // this thread: LDS column `cx`, tile rows wave * TYW - F .. wave * TYW + TYW - 1 + F
    const int cx = RMAX + lane;
    const int cy0 = RMAX + F + wave * TYW;           // LDS row of the thread's first OUTPUT row
    const int pzc = rz + FZ;                         // LDS plane of z itself
    float P[NS][NPP][V];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
        for (int z = 0; z < NPP; ++z)
#pragma unroll
            for (int v = 0; v < V; ++v)
                P[s_][z][v] = lds[(pzc - FZ + z) * pstride + v * vstride + (cy0 - F + s_) * cols + cx];

    double tw[TYW], tsq[NEFF ? TYW : 1];
    float wmax[TYW];
    float ws[TYW][V];
#pragma unroll
    for (int p_ = 0; p_ < TYW; ++p_) {
        tw[p_] = 0.0;
        if (NEFF) tsq[p_] = 0.0;
        wmax[p_] = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) ws[p_][v] = 0.f;
    }
    const float inv_norm = (float)(1.0 / (double)a.dsq_norm);
    const float neg_inv_h2 = (float)(-1.0 / a.h2);
    const float two_sigma2 = (float)a.two_sigma2;

    // The search, in two precisions.  DACC = false: the squared differences summed in float32 (a tree of depth
    // ~16: <= ~1e-6 relative on the distance, i.e. E x 1e-6 on a weight e^-E -- plus the float32 exponent of
    // __expf -- fine while a pixel's LARGEST weight is not small).  DACC = true: the reference's float32 squares
    // accumulated in double and the weight's exponent formed in double (split into integer and fraction for the
    // hardware exp2): ~1e-7 on every weight whatever its size.  A wave in which some pixel's largest float32
    // weight is below kPatch3Redo searches again in the second form (wave-uniform; about the cost of ONE per-pixel
    // evaluation in double): pixels next to a x4 step have weights of e^-20 and less, and at E = 69 the float32
    // form was off by 6e-5 (config 5's share, 5 values of 196 608).
    auto search = [&](auto dacc_c) {
        constexpr bool DACC = decltype(dacc_c)::value;
        const double inv_norm_d = 1.0 / (double)a.dsq_norm;
        const double neg_inv_h2_log2e = (-1.0 / a.h2) * 1.4426950408889634;
        for (int dz = -rz; dz <= rz; ++dz)
            for (int dy = -r0; dy <= r0; ++dy) {
                const float *qrow = lds + (pzc - FZ + dz) * pstride + (cy0 - F + dy) * cols + cx;
                for (int dx = -r1; dx <= r1; ++dx) {
                    if (dz == 0 && dy == 0 && dx == 0) continue;
                    typename std::conditional<DACC, double, float>::type rs[NS];
#pragma unroll
                    for (int s_ = 0; s_ < NS; ++s_) {
                        if (DACC) {
                            double e = 0.0;
#pragma unroll
                            for (int z = 0; z < NPP; ++z)
#pragma unroll
                                for (int v = 0; v < V; ++v) {
                                    const float df = P[s_][z][v] - qrow[z * pstride + v * vstride + s_ * cols + dx];
                                    const float sq = df * df;
                                    e = e + (double)sq;
                                }
                            // the row sum across lanes, the two words of the double shifted separately
                            double acc = e, up = e, dn = e;
#pragma unroll
                            for (int j = 0; j < F; ++j) {
                                const unsigned long long ub = __builtin_bit_cast(unsigned long long, up);
                                const unsigned long long db = __builtin_bit_cast(unsigned long long, dn);
                                const unsigned ulo = __builtin_bit_cast(unsigned, dpp_from_prev(__builtin_bit_cast(float, (unsigned)ub)));
                                const unsigned uhi = __builtin_bit_cast(unsigned, dpp_from_prev(__builtin_bit_cast(float, (unsigned)(ub >> 32))));
                                const unsigned dlo = __builtin_bit_cast(unsigned, dpp_from_next(__builtin_bit_cast(float, (unsigned)db)));
                                const unsigned dhi = __builtin_bit_cast(unsigned, dpp_from_next(__builtin_bit_cast(float, (unsigned)(db >> 32))));
                                up = __builtin_bit_cast(double, ((unsigned long long)uhi << 32) | ulo);
                                dn = __builtin_bit_cast(double, ((unsigned long long)dhi << 32) | dlo);
                                acc = (acc + up) + dn;
                            }
                            rs[s_] = acc;
                        } else {
                            float e = 0.f;
#pragma unroll
                            for (int z = 0; z < NPP; ++z)
#pragma unroll
                                for (int v = 0; v < V; ++v) {
                                    const float df = P[s_][z][v] - qrow[z * pstride + v * vstride + s_ * cols + dx];
                                    e = e + df * df;
                                }
                            rs[s_] = nlm_row_sum_lanes<F>(e);
                        }
                    }
#pragma unroll
                    for (int p_ = 0; p_ < TYW; ++p_) {
                        auto D = rs[p_];
#pragma unroll
                        for (int j = 1; j <= 2 * F; ++j) D = D + rs[p_ + j];
                        float w;
                        if (DACC) {
                            const double t = (D * inv_norm_d) - a.two_sigma2;
                            const double m = (0.0 > t) ? 0.0 : t;
                            const double xe = m * neg_inv_h2_log2e;               // log2 of the weight, <= 0
                            const double xn = rint(xe);
                            const float fr = (float)(xe - xn);                    // in [-0.5, 0.5], exact enough
                            const float wn = __builtin_amdgcn_exp2f(fr);
                            w = xn < -200.0 ? 0.f : ldexpf(wn, (int)xn);
                        } else {
                            const float d2 = D * inv_norm;
                            const float t = d2 - two_sigma2;
                            const float m = (0.f > t) ? 0.f : t;
                            w = __expf(m * neg_inv_h2);
                        }
                        tw[p_] = tw[p_] + (double)w;
                        if (NEFF) tsq[p_] = tsq[p_] + (double)w * (double)w;
                        wmax[p_] = w > wmax[p_] ? w : wmax[p_];
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            // (float)((double)ws + (double)w * (double)a): the product of two floats is exact
                            // in double, so the single-rounding fused multiply-add gives the same value
                            const float av = qrow[FZ * pstride + v * vstride + (F + p_) * cols + dx];
                            ws[p_][v] = __fmaf_rn(w, av, ws[p_][v]);
                        }
                    }
                }
            }
    };
    if (!block_nonfinite) {
        search(std::integral_constant<bool, false>());
        bool small = false;
#pragma unroll
        for (int p_ = 0; p_ < TYW; ++p_) small = small || (wmax[p_] < kPatch3Redo);
        if (__any(small)) {
#pragma unroll
            for (int p_ = 0; p_ < TYW; ++p_) {
                tw[p_] = 0.0;
                if (NEFF) tsq[p_] = 0.0;
                wmax[p_] = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) ws[p_][v] = 0.f;
            }
            search(std::integral_constant<bool, true>());
        }
    }

    const int64_t x = x0 - F + lane;
    const bool col_out = lane >= F && lane < 64 - F && x < a.chi1;
    unsigned exact_mask = 0;
#pragma unroll
    for (int p_ = 0; p_ < TYW; ++p_) {
        const int64_t y = y0 + wave * TYW + p_;
        if (y < a.chi0 && col_out) {
            if (block_nonfinite || !(wmax[p_] >= 1e-30f) ||
                (NEFF && nlm_neff_ill(tw[p_], NEFF ? tsq[p_] : 0.0, a.n_eff))) {
                exact_mask |= 1u << p_;
                continue;
            }
            bool fail;
            const double wself = nlm_self_weight(tw[p_], NEFF ? tsq[p_] : 0.0, (double)wmax[p_], a.n_eff,
                                                 a.neff_policy, a.status, &fail);
            if (!fail) {
                const double total = tw[p_] + wself;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float c = P[F + p_][FZ][v];
                    const float sfin = (float)((double)ws[p_][v] + (wself * (double)c));
                    a.out[i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x] = (float)((double)sfin / total);
                }
            }
        }
    }
    if (__any(exact_mask != 0u)) {
        for (int p_ = 0; p_ < TYW; ++p_) {
            if (!((exact_mask >> p_) & 1u)) continue;
            // nd/_filters.pyx:363-420 for this one pixel, from the staged planes, in double
            const int py = cy0 + p_;
            const int64_t y = y0 + wave * TYW + p_;
            if (!(a.n_eff >= 0 && (a.n_eff - 1.0) == 0) &&
                nlm_own_patch_nan3<F, FZ, V>(lds, pzc - FZ, pstride, vstride, cols, py, cx)) {
                // a NaN in the pixel's own patch: every distance, weight and sum is NaN (nd/_filters.pyx:386-420)
#pragma unroll
                for (int v = 0; v < V; ++v)
                    a.out[i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x] = __builtin_nanf("");
                continue;
            }
            double t_w = 0.0, t_sq = 0.0, m_w = 0.0;
            float wsum[V];
#pragma unroll
            for (int v = 0; v < V; ++v) wsum[v] = 0.f;
            for (int dz = -rz; dz <= rz; ++dz)
                for (int dy = -r0; dy <= r0; ++dy)
                    for (int dx = -r1; dx <= r1; ++dx) {
                        if (dz == 0 && dy == 0 && dx == 0) continue;
                        double dsq = 0.0;
                        for (int z = -FZ; z <= FZ; ++z)
                            for (int i = -F; i <= F; ++i)
                                for (int j = -F; j <= F; ++j)
#pragma unroll
                                    for (int v = 0; v < V; ++v) {
                                        const float *pb = lds + (pzc + z) * pstride + v * vstride;
                                        const float *qb = lds + (pzc + z + dz) * pstride + v * vstride;
                                        const float df = pb[(py + i) * cols + cx + j] - qb[(py + dy + i) * cols + cx + dx + j];
                                        const float sq = df * df;
                                        dsq = dsq + (double)sq;
                                    }
                        dsq = dsq / (double)a.dsq_norm;
                        const double t = dsq - a.two_sigma2;
                        const double m = (0.0 > t) ? 0.0 : t;
                        const double w = exp((-m) / a.h2);
                        t_w = t_w + w;
                        t_sq = t_sq + (w * w);
                        if (w > m_w) m_w = w;
#pragma unroll
                        for (int v = 0; v < V; ++v)
                            wsum[v] = (float)((double)wsum[v] +
                                              (w * (double)lds[(pzc + dz) * pstride + v * vstride + (py + dy) * cols + cx + dx]));
                    }
            bool fail;
            const double wself = nlm_self_weight(t_w, t_sq, m_w, a.n_eff, a.neff_policy, a.status, &fail);
            if (!fail) {
                const double total = t_w + wself;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float c = lds[pzc * pstride + v * vstride + py * cols + cx];
                    const float sfin = (float)((double)wsum[v] + (wself * (double)c));
                    a.out[i2 * a.so2 + (int64_t)v * a.so3 + y * a.so0 + x] = (float)((double)sfin / total);
                }
            }
        }
    }
}

constexpr int kPatch3RMax = 4;

template <int F, int FZ, int V, int TYW>
static int launch_patch3(const NlmTiledArgs &a, int64_t nb, size_t lds, hipStream_t stream)
{
    if (a.n_eff >= 0) {
        if (lds > 64 * 1024)
            ND_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&nlmeans_patch3_kernel<F, FZ, V, true, kPatch3RMax, TYW>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((nlmeans_patch3_kernel<F, FZ, V, true, kPatch3RMax, TYW>), dim3((unsigned)nb), dim3(256), lds, stream, a);
    } else {
        if (lds > 64 * 1024)
            ND_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&nlmeans_patch3_kernel<F, FZ, V, false, kPatch3RMax, TYW>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((nlmeans_patch3_kernel<F, FZ, V, false, kPatch3RMax, TYW>), dim3((unsigned)nb), dim3(256), lds, stream, a);
    }
    return ND_AMD_OK;
}

template <int F, int FZ, int TYW>
static int launch_patch3_v(const NlmTiledArgs &a, int64_t nb, size_t lds, hipStream_t stream)
{
    // (the 8-row tiles exist for the plane counts that need them: three and four variables)
    if constexpr (TYW == 4) {
        switch (a.nvars) {
        case 1: return launch_patch3<F, FZ, 1, TYW>(a, nb, lds, stream);
        case 2: return launch_patch3<F, FZ, 2, TYW>(a, nb, lds, stream);
        case 3: return launch_patch3<F, FZ, 3, TYW>(a, nb, lds, stream);
        default: return launch_patch3<F, FZ, 4, TYW>(a, nb, lds, stream);
        }
    } else {
        if (a.nvars == 3) return launch_patch3<F, FZ, 3, TYW>(a, nb, lds, stream);
        if (a.nvars == 4) return launch_patch3<F, FZ, 4, TYW>(a, nb, lds, stream);
        return ND_AMD_EUNSUPPORTED;
    }
}

static int nlm_try_tiled(const void *arr, void *out, int dtype, const int64_t N[3], int64_t nvars,
                         const int64_t si[4], const int64_t so[4], const uint32_t r[3],
                         const uint32_t f[3], double sigma, double h, double n_eff, int patch_mode,
                         int neff_policy, int32_t *status_dev, const int64_t G[3],
                         const int64_t toff[3], const int64_t clo[3], const int64_t chi[3],
                         hipStream_t stream)
{
    static const bool disabled = getenv("ND_AMD_NO_TILED") != nullptr;
    if (disabled) return 0;
    // float64 arrays: the uniform-weight window kernel only (the reference-compatible mode with
    // f > 0); everything else of float64 stays in the per-pixel kernel
    const bool f64 = dtype != ND_AMD_F32;
    if (f64 && !((patch_mode == 0) && (f[0] > 0 || f[1] > 0 || f[2] > 0))) return 0;
    if (nvars < 1) return 0;
    // Canonical axes of the tiled kernels: rows, contiguous columns, and a third axis that is
    // either a plain slice axis or (window kernel only) a window axis visited OUTERMOST.
    //   layout A: user axes (rows, cols, slices) -- e.g. (y, x, time): cols contiguous, r[2] = f[2] = 0
    //   layout B: user axes (z, rows, cols)      -- e.g. (time, y, x): cols contiguous, z = first
    //             filter dimension, so the reference visits it outermost (nd/_filters.pyx:363-370)
    int A0, A1, A2;
    if (si[1] == 1 && so[1] == 1 && r[2] == 0 && f[2] == 0) {
        A0 = 0; A1 = 1; A2 = 2;
    } else if (si[2] == 1 && so[2] == 1) {
        A0 = 1; A1 = 2; A2 = 0;
    } else {
        return 0;
    }
    const uint32_t rz = r[A2], fz = f[A2];
    if (r[A0] == 0 && r[A1] == 0 && rz == 0) return 0;
    for (int d = 0; d < 3; ++d)
        if (G[d] > 0x3fffffff || N[d] > 0x3fffffff) return 0;
    NlmTiledArgs a;
    a.arr = static_cast<const float *>(arr);
    a.out = static_cast<float *>(out);
    a.N0 = N[A0]; a.N1 = N[A1]; a.N2 = N[A2];
    a.G0 = G[A0]; a.G1 = G[A1]; a.Gz = G[A2];
    a.off0 = toff[A0]; a.off1 = toff[A1]; a.offz = toff[A2];
    a.clo0 = clo[A0]; a.chi0 = chi[A0]; a.clo1 = clo[A1]; a.chi1 = chi[A1];
    a.clo2 = clo[A2]; a.chi2 = chi[A2];
    a.si0 = si[A0]; a.si2 = si[A2]; a.si3 = si[3];
    a.so0 = so[A0]; a.so2 = so[A2]; a.so3 = so[3];
    a.r0 = (int)r[A0]; a.r1 = (int)r[A1]; a.f0 = (int)f[A0]; a.f1 = (int)f[A1];
    a.rz = (int)rz;
    a.nvars = (int)nvars;
    a.dsq_norm = (float)((((uint32_t)nvars * (2u * f[0] + 1u)) * (2u * f[1] + 1u)) * (2u * f[2] + 1u));
    a.two_sigma2 = 2.0 * (sigma * sigma);
    a.h2 = h * h;
    a.n_eff = n_eff;
    a.neff_policy = neff_policy;
    a.status = status_dev;
    const int64_t ey = a.chi0 - a.clo0, ex = a.chi1 - a.clo1, nsl = a.chi2 - a.clo2;
    if (ey < 1 || ex < 1 || nsl < 1) return 0;

    const bool uniform = (patch_mode == 0) && (f[0] > 0 || f[1] > 0 || f[2] > 0);
    if (uniform) {
        // weight exp(-max(0 - 2 sigma^2, 0) / h^2) must be exactly 1
        if (!(sigma == sigma) || !(h == h) || h == 0.0 || !(sigma * sigma < INFINITY)) return 0;
        if (a.r1 > 16) return 0;
        a.tiles_x = (int)ceil_div(ex, kWinTX);
        a.tiles_y = (int)ceil_div(ey, kWinTY);
        static const bool no_roll = getenv("ND_AMD_NLM_NOROLL") != nullptr;
        static const bool no_stream3 = getenv("ND_AMD_NLM_NOSTREAM3") != nullptr;
        if (!f64 && !no_roll && !no_stream3 && rz == 1 && a.r0 == a.r1 && a.r1 >= 1 && a.r1 <= 5 &&
            a.si0 >= 0 && (a.N0 * a.si0 + a.N1) * 4 < 0x7fffffffLL) {
            const int64_t nbr = (int64_t)a.tiles_x * a.tiles_y * nvars;
            if (nbr <= 0x7fffffffLL) {
                KernelTimer timer(ND_AMD_KERNEL_NLMEANS_TILED, stream);
                if (launch_stream3(a, nbr, stream)) return 1;
            }
        }
        if (!f64 && !no_roll && a.r0 <= kRollR0Max && a.r1 <= 10) {
            const size_t nw = ((2 * (size_t)a.r1 + 4) + 3) / 4 * 4;
            const size_t lds_r = (size_t)(2 * rz + 1) * (kWinTY + 2 * a.r0) * (kWinTX - 4 + nw) * sizeof(float);
            const int64_t nbr = (int64_t)a.tiles_x * a.tiles_y * nvars;
            if (lds_r <= kBigLds && nbr <= 0x7fffffffLL) {
                KernelTimer timer(ND_AMD_KERNEL_NLMEANS_TILED, stream);
                if (launch_roll_r1(a, nbr, lds_r, stream)) return 1;
            }
        }
        const int64_t nb = (int64_t)a.tiles_x * a.tiles_y * nsl;
        const size_t rows = kWinTY + 2 * a.r0, cols = kWinTX + 2 * a.r1;
        const size_t lds = (size_t)(2 * rz + 1) * rows * cols * (f64 ? sizeof(double) : sizeof(float)) +
                           (rows + cols) * sizeof(int);
        if (nb > 0x7fffffffLL) return 0;
        if (f64) {
            if (lds > kBigLds) return 0;
            KernelTimer timer(ND_AMD_KERNEL_NLMEANS_TILED, stream);
#define ND_LAUNCH_WIN64(R)                                                                                     \
    do {                                                                                                      \
        if (lds > 64 * 1024)                                                                                  \
            ND_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&nlmeans_window_kernel<double, R>), \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
        hipLaunchKernelGGL((nlmeans_window_kernel<double, R>), dim3((unsigned)nb), dim3(256), lds, stream, a); \
    } while (0)
            if (a.r1 <= 4)
                ND_LAUNCH_WIN64(4);
            else if (a.r1 <= 10)
                ND_LAUNCH_WIN64(10);
            else
                ND_LAUNCH_WIN64(16);
#undef ND_LAUNCH_WIN64
            return 1;
        }
        if (lds > 64 * 1024) return 0;
        KernelTimer timer(ND_AMD_KERNEL_NLMEANS_TILED, stream);
        if (a.r1 <= 4)
            hipLaunchKernelGGL((nlmeans_window_kernel<float, 4>), dim3((unsigned)nb), dim3(256), lds, stream, a);
        else if (a.r1 <= 10)
            hipLaunchKernelGGL((nlmeans_window_kernel<float, 10>), dim3((unsigned)nb), dim3(256), lds, stream, a);
        else
            hipLaunchKernelGGL((nlmeans_window_kernel<float, 16>), dim3((unsigned)nb), dim3(256), lds, stream, a);
        return 1;
    }
    // true patch distances (patch_mode 1, or f = 0 in either mode: the loops run once)
    if (rz != 0 || fz != 0) {
        // a window along the third axis (round 6): nlmeans_patch3_kernel -- the visiting order is the
        // reference's only where that axis is the outermost filter dimension (layout B)
        static const bool no_patch3 = getenv("ND_AMD_NLM_NOPATCH3") != nullptr;
        const uint32_t F0z = (patch_mode == 1) ? f[A0] : 0u, F1z = (patch_mode == 1) ? f[A1] : 0u;
        const uint32_t FZz = (patch_mode == 1) ? fz : 0u;
        if (no_patch3 || A2 != 0 || f64 || nvars > 4) return 0;
        if (patch_mode == 0 && (f[A0] != 0 || f[A1] != 0 || fz != 0)) return 0;
        // (3 x 3 patches, or none, in the plane; 5 x 5 and larger: the per-pixel kernel -- every instantiation costs
        //  build time, and no caller of the reference uses them with a window along time)
        if (F0z != F1z || F0z > 1 || FZz > 1 || rz > 2 || a.r0 > kPatch3RMax || a.r1 > kPatch3RMax) return 0;
        const int Fp = (int)F0z;
        const size_t cols3 = 64 + 2 * (size_t)kPatch3RMax;
        const size_t np3 = 2 * (size_t)(rz + FZz) + 1;
        // 16-row tiles where their planes fit the LDS, 8-row tiles otherwise (seven planes of four variables)
        int tyw = kPatch3TYW;
        size_t rows3 = 4 * (size_t)tyw + 2 * (size_t)(kPatch3RMax + Fp);
        size_t lds3 = np3 * (size_t)nvars * rows3 * cols3 * sizeof(float) + (rows3 + cols3) * sizeof(int);
        if (lds3 > kBigLds) {
            if (nvars < 3) return 0;
            tyw = 2;
            rows3 = 4 * (size_t)tyw + 2 * (size_t)(kPatch3RMax + Fp);
            lds3 = np3 * (size_t)nvars * rows3 * cols3 * sizeof(float) + (rows3 + cols3) * sizeof(int);
        }
        a.tiles_x = (int)ceil_div(ex, 64 - 2 * Fp);
        a.tiles_y = (int)ceil_div(ey, 4 * tyw);
        const int64_t nb3 = (int64_t)a.tiles_x * a.tiles_y * nsl;
        if (lds3 > kBigLds || nb3 > 0x7fffffffLL) return 0;
        KernelTimer timer(ND_AMD_KERNEL_NLMEANS_TILED, stream);
        int rc3 = ND_AMD_OK;
#define ND_P3(F_, FZ_)                                                                     \
    (tyw == 4 ? launch_patch3_v<F_, FZ_, 4>(a, nb3, lds3, stream) : launch_patch3_v<F_, FZ_, 2>(a, nb3, lds3, stream))
        if (Fp == 0 && FZz == 0) rc3 = ND_P3(0, 0);
        else if (Fp == 1 && FZz == 0) rc3 = ND_P3(1, 0);
        else if (Fp == 1 && FZz == 1) rc3 = ND_P3(1, 1);
        else if (Fp == 0 && FZz == 1) rc3 = ND_P3(0, 1);
        else return 0;
#undef ND_P3
        return rc3 == ND_AMD_OK ? 1 : 0;
    }
    const uint32_t F0 = (patch_mode == 1) ? f[A0] : 0u, F1 = (patch_mode == 1) ? f[A1] : 0u;
    if (patch_mode == 0 && (f[A0] != 0 || f[A1] != 0)) return 0;
    // (round 6: patches of 9 x 9 and 11 x 11 -- f = 4, 5 -- in the one-column-per-lane kernel, whose sliding sums are
    //  generic in F; before, they took the per-pixel kernel: r = 5, f = 4 on 4 x 1024 x 2048 177 ms)
    if (F0 != F1 || F0 > 5 || nvars > 4) return 0;
    static const bool no_patch2 = getenv("ND_AMD_NLM_PATCH1") != nullptr;
    if (!no_patch2 && F0 >= 1 && F0 <= 3) {
        // cross-lane form: two columns per lane
        const int hl = (F0 == 3) ? 2 : 1, tx2 = 2 * (64 - 2 * hl), tyw2 = (nvars == 1) ? Patch2Rows<1>::TYW : Patch2Rows<2>::TYW;
        int m = a.r1 + 2 * hl + (a.r1 & 1);
        if (m <= kPatch2Margin) m = kPatch2Margin;     // the constant-pitch instantiation (launch_patch2)
        const size_t cols2 = (size_t)tx2 + 2 * (size_t)m, rows2 = 4 * (size_t)tyw2 + 2 * (size_t)(a.r0 + F0);
        const size_t lds2 = (size_t)nvars * rows2 * cols2 * sizeof(float) + (rows2 + cols2) * sizeof(int);
        a.tiles_x = (int)ceil_div(ex + 1, tx2);        // + 1: the tiles start on an even global column
        a.tiles_y = (int)ceil_div(ey, 4 * tyw2);
        if (lds2 <= kBigLds && (int64_t)a.tiles_x * a.tiles_y * nsl <= 0x7fffffffLL) {
            KernelTimer timer(ND_AMD_KERNEL_NLMEANS_TILED, stream);
            bool ok2 = false;
            switch (F0) {
            case 1: ok2 = launch_patch2_v<1>(a, nsl, lds2, stream); break;
            case 2: ok2 = launch_patch2_v<2>(a, nsl, lds2, stream); break;
            default: ok2 = launch_patch2_v<3>(a, nsl, lds2, stream); break;
            }
            if (ok2) return 1;
        }
    }
    const int tyw = (nvars == 1) ? 16 : 8;
    a.tiles_x = (int)ceil_div(ex, 64);
    a.tiles_y = (int)ceil_div(ey, 4 * tyw);
    const size_t cols = 64 + 2 * (a.r1 + F0);
    const size_t rows = 4 * tyw + 2 * (a.r0 + F0);
    const size_t lds = (size_t)nvars * rows * cols * sizeof(float) + (rows + cols) * sizeof(int);
    if (lds > kBigLds) return 0;
    if ((int64_t)a.tiles_x * a.tiles_y * nsl > 0x7fffffffLL) return 0;
    KernelTimer timer(ND_AMD_KERNEL_NLMEANS_TILED, stream);
    bool ok = false;
    switch (F0) {
    case 0: ok = launch_patch_v<0>(a, nsl, lds, stream); break;
    case 1: ok = launch_patch_v<1>(a, nsl, lds, stream); break;
    case 2: ok = launch_patch_v<2>(a, nsl, lds, stream); break;
    case 3: ok = launch_patch_v<3>(a, nsl, lds, stream); break;
    case 4: ok = launch_patch_v<4>(a, nsl, lds, stream); break;
    default: ok = launch_patch_v<5>(a, nsl, lds, stream); break;
    }
    return ok ? 1 : 0;
}

template <typename T>
static int nlmeans_impl(const void *arr, void *out, const int64_t N[3], int64_t nvars,
                        const int64_t si[4], const int64_t so[4], const uint32_t r[3],
                        const uint32_t f[3], double sigma, double h, double n_eff,
                        int patch_mode, int neff_policy, int32_t *status_dev,
                        const int64_t G[3], const int64_t toff[3], const int64_t clo[3],
                        const int64_t chi[3], hipStream_t stream)
{
    NlmArgs<T> a;
    a.arr = static_cast<const T *>(arr);
    a.out = static_cast<T *>(out);
    a.total = 1;
    for (int d = 0; d < 3; ++d) {
        a.N[d] = N[d];
        a.G[d] = G[d];
        a.toff[d] = toff[d];
        a.clo[d] = clo[d];
        a.chi[d] = chi[d];
        if (clo[d] < 0 || chi[d] > N[d] || clo[d] > chi[d] || toff[d] < 0 ||
            toff[d] + N[d] > G[d]) {
            set_error("nd_amd_nlmeans3d: inconsistent tile description on axis %d", d);
            return ND_AMD_EINVAL;
        }
        a.total *= (chi[d] - clo[d]);
        a.r[d] = (int64_t)r[d];
        a.dhi[d] = (int64_t)(f[d] + 1u);
        a.dlo[d] = patch_mode ? -(int64_t)f[d] : (int64_t)(uint32_t)(0u - f[d]);
        // a single reflection must land inside the global array (the reference indexes out of
        // bounds otherwise: boundscheck is off, nd/_filters.pyx:317)
        const int64_t reach = (int64_t)r[d] + (patch_mode || f[d] == 0 ? (int64_t)f[d] : 0);
        if (G[d] > 0 && reach > G[d] - 1) {
            set_error("nd_amd_nlmeans3d: r+f = %lld exceeds the array extent %lld on axis %d",
                      (long long)reach, (long long)G[d], d);
            return ND_AMD_EINVAL;
        }
        // the tile must hold every element the written range touches
        const int64_t need_lo = toff[d] + clo[d] - reach, need_hi = toff[d] + chi[d] - 1 + reach;
        const int64_t have_lo = toff[d], have_hi = toff[d] + N[d] - 1;
        if (chi[d] > clo[d]) {
            const int64_t rl = need_lo < 0 ? 0 : need_lo;             // reflected reads stay inside
            const int64_t rh = need_hi > G[d] - 1 ? G[d] - 1 : need_hi;
            int64_t lo_reach = rl, hi_reach = rh;
            // reflection of out-of-range coordinates maps into [0, reach] / [G-1-reach, G-1]
            if (need_lo < 0 && -need_lo > hi_reach) hi_reach = -need_lo;
            if (need_hi > G[d] - 1 && 2 * G[d] - 2 - need_hi < lo_reach) lo_reach = 2 * G[d] - 2 - need_hi;
            if (lo_reach < have_lo || hi_reach > have_hi) {
                set_error("nd_amd_nlmeans3d: tile on axis %d lacks the halo the window needs", d);
                return ND_AMD_EINVAL;
            }
        }
    }
    for (int d = 0; d < 4; ++d) {
        a.si[d] = si[d];
        a.so[d] = so[d];
    }
    a.nvars = (int)nvars;
    // nd/_filters.pyx:337: unsigned-int product assigned to `floating`
    a.dsq_norm = (T)((((uint32_t)nvars * (2u * f[0] + 1u)) * (2u * f[1] + 1u)) * (2u * f[2] + 1u));
    a.two_sigma2 = 2.0 * (sigma * sigma);
    a.h2 = h * h;
    a.n_eff = n_eff;
    a.neff_policy = neff_policy;
    a.status = status_dev;
    if (status_dev) ND_HIP_CHECK(hipMemsetAsync(status_dev, 0, sizeof(int32_t), stream));
    if (a.total == 0) return ND_AMD_OK;

    if (nlm_try_tiled(arr, out, sizeof(T) == 4 ? ND_AMD_F32 : ND_AMD_F64, N, nvars, si, so, r, f,
                      sigma, h, n_eff, patch_mode, neff_policy, status_dev, G, toff, clo, chi,
                      stream)) {
        ND_HIP_CHECK(hipGetLastError());
        return ND_AMD_OK;
    }

    // lanes along the axis with the smallest non-trivial input stride
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j) {
            const int64_t ei = chi[ord[i]] - clo[ord[i]], ej = chi[ord[j]] - clo[ord[j]];
            const int64_t ki = ei > 1 ? llabs(si[ord[i]]) : INT64_MAX;
            const int64_t kj = ej > 1 ? llabs(si[ord[j]]) : INT64_MAX;
            if (kj < ki) {
                int t = ord[i];
                ord[i] = ord[j];
                ord[j] = t;
            }
        }
    a.order[0] = ord[0];
    a.order[1] = ord[1];
    a.order[2] = ord[2];

    const int64_t nblocks = ceil_div(a.total, 256);
    if (nblocks > 0x7fffffffLL) {
        set_error("nd_amd_nlmeans3d: array too large for one launch");
        return ND_AMD_EUNSUPPORTED;
    }
    {
        KernelTimer timer(ND_AMD_KERNEL_NLMEANS, stream);
        if (nvars <= 1)
            hipLaunchKernelGGL((nlmeans_generic_kernel<T, 1>), dim3((unsigned)nblocks), dim3(256),
                               0, stream, a);
        else if (nvars <= 4)
            hipLaunchKernelGGL((nlmeans_generic_kernel<T, 4>), dim3((unsigned)nblocks), dim3(256),
                               0, stream, a);
        else
            hipLaunchKernelGGL((nlmeans_generic_kernel<T, 16>), dim3((unsigned)nblocks),
                               dim3(256), 0, stream, a);
    }
    ND_HIP_CHECK(hipGetLastError());
    return ND_AMD_OK;
}

}  // namespace nd_amd

using namespace nd_amd;

extern "C" int nd_amd_nlmeans3d(const void *arr, void *out, int dtype, const int64_t N[3],
                                int64_t nvars, const int64_t in_strides[4],
                                const int64_t out_strides[4], const uint32_t r[3],
                                const uint32_t f[3], double sigma, double h, double n_eff,
                                int patch_mode, int neff_policy, int32_t *status_dev,
                                const int64_t global_N[3], const int64_t tile_off[3],
                                const int64_t core_lo[3], const int64_t core_hi[3],
                                void *hip_stream)
{
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_nlmeans3d: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (!N || !in_strides || !out_strides || !r || !f) {
        set_error("nd_amd_nlmeans3d: null argument");
        return ND_AMD_EINVAL;
    }
    if (nvars < 0 || nvars > 16) {
        set_error("nd_amd_nlmeans3d: nvars = %lld not supported (0..16)", (long long)nvars);
        return ND_AMD_EUNSUPPORTED;
    }
    for (int d = 0; d < 3; ++d)
        if (N[d] < 0) {
            set_error("nd_amd_nlmeans3d: negative dimension");
            return ND_AMD_EINVAL;
        }
    if (n_eff >= 0 && neff_policy == 1 && !status_dev) {
        set_error("nd_amd_nlmeans3d: neff_policy 1 needs status_dev");
        return ND_AMD_EINVAL;
    }
    if (N[0] * N[1] * N[2] * nvars == 0) return ND_AMD_OK;
    if (!arr || !out) {
        set_error("nd_amd_nlmeans3d: null data pointer");
        return ND_AMD_EINVAL;
    }
    const int64_t zero3[3] = {0, 0, 0};
    const int64_t *G = global_N ? global_N : N;
    const int64_t *toff = tile_off ? tile_off : zero3;
    const int64_t *clo = core_lo ? core_lo : zero3;
    const int64_t *chi = core_hi ? core_hi : N;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (dtype == ND_AMD_F32)
        return nlmeans_impl<float>(arr, out, N, nvars, in_strides, out_strides, r, f, sigma, h,
                                   n_eff, patch_mode, neff_policy, status_dev, G, toff, clo, chi,
                                   stream);
    return nlmeans_impl<double>(arr, out, N, nvars, in_strides, out_strides, r, f, sigma, h, n_eff,
                                patch_mode, neff_policy, status_dev, G, toff, clo, chi, stream);
}
