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
