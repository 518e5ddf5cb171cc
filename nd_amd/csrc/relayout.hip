// nd_amd/csrc/relayout.hip -- the step in front of the hot path: the reference stacks its
// variables as (y, x, time[, variable]) with time fastest (nd/change.py:66-67,
// `to_array().transpose('y', 'x', 'time', 'variable')`); the kernels here read planar stacks
// (time, y, x) with x fastest.  This is the transpose between the two for device-resident data:
//     out[t * out_date_stride + p] = in[p * in_pixel_stride + t * in_date_stride]
// with p the flattened (y, x) pixel index.  in_date_stride 1 = a real (y, x, time) array,
// 2 = the real or imaginary half of an interleaved complex64/128 array (C12).
//
// A 256-thread block moves PB pixels: their source elements form one contiguous span, read with
// fully coalesced loads into LDS (pixel-major, pitch odd so that the column reads below are
// conflict-free), then written date by date, lanes along the pixels (coalesced plane rows).
#include "common.hpp"

namespace nd_amd {

template <typename T>
struct RelayoutArgs {
    const T *in;
    T *out;
    T *out2;                    // second half of an interleaved source (planar direction only), or null
    int64_t npix, out_date_stride;
    int k, ids;                 // dates, element distance between dates in the source (1 or 2)
    int span_per_pixel;         // k * ids: source elements per pixel
    unsigned magic;             // ceil(2^32 / span_per_pixel): e / span_per_pixel = umulhi(e, magic)
    int pb;                     // pixels per block (power of two, <= 256)
};

template <typename T>
__global__ void __launch_bounds__(256) relayout_planar_kernel(const RelayoutArgs<T> a)
{
    extern __shared__ __align__(16) unsigned char nd_smem_rl[];
    T *lds = reinterpret_cast<T *>(nd_smem_rl);
    const int tid = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * a.pb;
    const int64_t left = a.npix - p0;
    const int np = left < a.pb ? (int)left : a.pb;
    const bool both = a.out2 != nullptr;                  // keep both halves of an interleaved source
    const int spp = a.span_per_pixel, pitch = both ? (spp | 1) : (a.k | 1);
    const int total = np * spp;
    const T *src = a.in + p0 * spp;
    for (int e0 = 0; e0 < total; e0 += 256 * 8) {
        T buf[8];
        int dsti[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256 + tid;
            dsti[u] = -1;
            if (e < total) {
                const int q = (int)__umulhi((unsigned)e, a.magic);      // pixel within the block
                const int r = e - q * spp;                              // element within the pixel
                // interleaved source, one half wanted: only that half's elements are touched (the
                // other half's last element may lie beyond the end of the allocation)
                if (a.ids == 1 || both) {
                    dsti[u] = q * pitch + r;
                } else if ((r & 1) == 0) {
                    dsti[u] = q * pitch + (r >> 1);
                }
                if (dsti[u] >= 0) buf[u] = __builtin_nontemporal_load(src + e);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (dsti[u] >= 0) lds[dsti[u]] = buf[u];
    }
    __syncthreads();
    const int p = tid & (a.pb - 1), tg = tid / a.pb, ntg = 256 / a.pb;
    if (p < np) {
        T *dst = a.out + p0 + p;
        if (!both) {
            for (int t = tg; t < a.k; t += ntg)
                __builtin_nontemporal_store(lds[p * pitch + t], dst + (int64_t)t * a.out_date_stride);
        } else {
            T *dst2 = a.out2 + p0 + p;
            for (int t = tg; t < a.k; t += ntg) {
                __builtin_nontemporal_store(lds[p * pitch + 2 * t], dst + (int64_t)t * a.out_date_stride);
                __builtin_nontemporal_store(lds[p * pitch + 2 * t + 1], dst2 + (int64_t)t * a.out_date_stride);
            }
        }
    }
}

// The way back: in[t * in_date_stride + p]  ->  out[p * k * ods + t * ods]  (ods = 1, or 2 for one
// half of an interleaved complex array).  Plane rows are read with lanes along the pixels, the
// pixel-major span is written with consecutive lanes on consecutive elements.
template <typename T>
__global__ void __launch_bounds__(256) relayout_pixel_major_kernel(const RelayoutArgs<T> a)
{
    extern __shared__ __align__(16) unsigned char nd_smem_rl2[];
    T *lds = reinterpret_cast<T *>(nd_smem_rl2);
    const int tid = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * a.pb;
    const int64_t left = a.npix - p0;
    const int np = left < a.pb ? (int)left : a.pb;
    const int spp = a.span_per_pixel, pitch = a.k | 1;
    const int p = tid & (a.pb - 1), tg = tid / a.pb, ntg = 256 / a.pb;
    if (p < np) {
        const T *src = a.in + p0 + p;
        for (int t0 = tg; t0 < a.k; t0 += ntg * 8) {
            T buf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * ntg;
                if (t < a.k) buf[u] = __builtin_nontemporal_load(src + (int64_t)t * a.out_date_stride);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * ntg;
                if (t < a.k) lds[p * pitch + t] = buf[u];
            }
        }
    }
    __syncthreads();
    const int total = np * spp;
    T *dst = a.out + p0 * spp;
    for (int e = tid; e < total; e += 256) {
        const int q = (int)__umulhi((unsigned)e, a.magic);
        const int r = e - q * spp;
        if (a.ids == 1)
            __builtin_nontemporal_store(lds[q * pitch + r], dst + e);
        else if ((r & 1) == 0)
            dst[e] = lds[q * pitch + (r >> 1)];
    }
}

template <typename T>
static int relayout_impl(const void *in, void *out, void *out2, int64_t npix, int64_t k, int64_t ids,
                         int64_t out_date_stride, bool to_planar, hipStream_t stream)
{
    RelayoutArgs<T> a;
    a.in = static_cast<const T *>(in);
    a.out = static_cast<T *>(out);
    a.out2 = static_cast<T *>(out2);
    a.npix = npix;
    a.out_date_stride = out_date_stride;
    a.k = (int)k;
    a.ids = (int)ids;
    a.span_per_pixel = (int)(k * ids);
    a.magic = (unsigned)((0x100000000ULL + (unsigned)a.span_per_pixel - 1) / (unsigned)a.span_per_pixel);
    // LDS image pb x (k | 1) elements, at most 48 KiB
    const size_t row_elems = out2 ? (size_t)((k * ids) | 1) : (size_t)(k | 1);
    int pb = 256;
    while (pb > 1 && (size_t)pb * row_elems * sizeof(T) > 48 * 1024) pb >>= 1;
    if ((size_t)pb * row_elems * sizeof(T) > 48 * 1024) {
        set_error("nd_amd_relayout_planar: %lld dates do not fit the staging buffer", (long long)k);
        return ND_AMD_EUNSUPPORTED;
    }
    // e / span_per_pixel through the 32-bit reciprocal is exact for e < 2^16 * ... : bound it
    if ((int64_t)pb * a.span_per_pixel > (1 << 20)) {
        set_error("nd_amd_relayout_planar: series too long");
        return ND_AMD_EUNSUPPORTED;
    }
    a.pb = pb;
    const int64_t nblocks = ceil_div(npix, (int64_t)pb);
    if (nblocks > 0x7fffffffLL) {
        set_error("nd_amd_relayout_planar: raster too large for one launch");
        return ND_AMD_EUNSUPPORTED;
    }
    const size_t lds = (size_t)pb * row_elems * sizeof(T);
    {
        KernelTimer timer(ND_AMD_KERNEL_RELAYOUT, stream);
        if (to_planar)
            hipLaunchKernelGGL((relayout_planar_kernel<T>), dim3((unsigned)nblocks), dim3(256), lds, stream, a);
        else
            hipLaunchKernelGGL((relayout_pixel_major_kernel<T>), dim3((unsigned)nblocks), dim3(256), lds,
                               stream, a);
    }
    ND_HIP_CHECK(hipGetLastError());
    return ND_AMD_OK;
}

}  // namespace nd_amd

using namespace nd_amd;

extern "C" int nd_amd_relayout_planar(const void *in, void *out, int dtype, int64_t npix, int64_t k,
                                      int64_t in_date_stride, int64_t out_date_stride,
                                      void *hip_stream)
{
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_relayout_planar: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (npix < 0 || k < 0 || (in_date_stride != 1 && in_date_stride != 2) || out_date_stride < npix) {
        set_error("nd_amd_relayout_planar: bad shape or strides");
        return ND_AMD_EINVAL;
    }
    if (npix == 0 || k == 0) return ND_AMD_OK;
    if (!in || !out) {
        set_error("nd_amd_relayout_planar: null data pointer");
        return ND_AMD_EINVAL;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (dtype == ND_AMD_F32)
        return relayout_impl<float>(in, out, nullptr, npix, k, in_date_stride, out_date_stride, true, stream);
    return relayout_impl<double>(in, out, nullptr, npix, k, in_date_stride, out_date_stride, true, stream);
}

extern "C" int nd_amd_relayout_pixel_major(const void *in, void *out, int dtype, int64_t npix,
                                           int64_t k, int64_t in_date_stride,
                                           int64_t out_date_stride, void *hip_stream)
{
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_relayout_pixel_major: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (npix < 0 || k < 0 || (out_date_stride != 1 && out_date_stride != 2) || in_date_stride < npix) {
        set_error("nd_amd_relayout_pixel_major: bad shape or strides");
        return ND_AMD_EINVAL;
    }
    if (npix == 0 || k == 0) return ND_AMD_OK;
    if (!in || !out) {
        set_error("nd_amd_relayout_pixel_major: null data pointer");
        return ND_AMD_EINVAL;
    }
    // RelayoutArgs: `ids` is the element distance between dates on the pixel-major side and
    // `out_date_stride` the plane pitch on the planar side, whichever direction the copy runs
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (dtype == ND_AMD_F32)
        return relayout_impl<float>(in, out, nullptr, npix, k, out_date_stride, in_date_stride, false, stream);
    return relayout_impl<double>(in, out, nullptr, npix, k, out_date_stride, in_date_stride, false, stream);
}

extern "C" int nd_amd_relayout_planar_complex(const void *in, void *out_re, void *out_im, int dtype,
                                              int64_t npix, int64_t k, int64_t out_date_stride,
                                              void *hip_stream)
{
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_relayout_planar_complex: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (npix < 0 || k < 0 || out_date_stride < npix) {
        set_error("nd_amd_relayout_planar_complex: bad shape or strides");
        return ND_AMD_EINVAL;
    }
    if (npix == 0 || k == 0) return ND_AMD_OK;
    if (!in || !out_re || !out_im) {
        set_error("nd_amd_relayout_planar_complex: null data pointer");
        return ND_AMD_EINVAL;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (dtype == ND_AMD_F32)
        return relayout_impl<float>(in, out_re, out_im, npix, k, 2, out_date_stride, true, stream);
    return relayout_impl<double>(in, out_re, out_im, npix, k, 2, out_date_stride, true, stream);
}

// ---- interleaved complex -> two real arrays of the same (contiguous) shape --------------------
// The device side of disassemble_complex (nd/io.py:26-69) for data that is already planar: a
// complex64 / complex128 variable in (time, y, x) order becomes C12__re and C12__im in one pass
// over its memory (two strided torch copies read it twice).  A thread moves four complex values:
// two 16-byte loads, one 16-byte store per half.
namespace nd_amd {
template <typename T>
__global__ void __launch_bounds__(256) split_complex_kernel(const T *__restrict__ in, T *__restrict__ re,
                                                            T *__restrict__ im, int64_t n)
{
    constexpr int V = 16 / (int)sizeof(T);          // elements per 16-byte access
    struct alignas(16) Vec {
        T v[V];
    };
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (i0 + V <= n) {
        const Vec a = *reinterpret_cast<const Vec *>(in + 2 * i0);
        const Vec b = *reinterpret_cast<const Vec *>(in + 2 * i0 + V);
        Vec r, m;
#pragma unroll
        for (int j = 0; j < V / 2; ++j) {
            r.v[j] = a.v[2 * j];
            m.v[j] = a.v[2 * j + 1];
            r.v[V / 2 + j] = b.v[2 * j];
            m.v[V / 2 + j] = b.v[2 * j + 1];
        }
        *reinterpret_cast<Vec *>(re + i0) = r;
        *reinterpret_cast<Vec *>(im + i0) = m;
    } else {
        for (int64_t i = i0; i < n; ++i) {
            re[i] = in[2 * i];
            im[i] = in[2 * i + 1];
        }
    }
}
}  // namespace nd_amd

extern "C" int nd_amd_split_complex(const void *in, void *out_re, void *out_im, int dtype, int64_t n,
                                    void *hip_stream)
{
    using namespace nd_amd;
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_split_complex: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (n < 0) {
        set_error("nd_amd_split_complex: negative length");
        return ND_AMD_EINVAL;
    }
    if (n == 0) return ND_AMD_OK;
    if (!in || !out_re || !out_im) {
        set_error("nd_amd_split_complex: null data pointer");
        return ND_AMD_EINVAL;
    }
    if ((((uintptr_t)in | (uintptr_t)out_re | (uintptr_t)out_im) & 15) != 0) {
        set_error("nd_amd_split_complex: pointers must be 16-byte aligned");
        return ND_AMD_EINVAL;
    }
    const int v = dtype == ND_AMD_F32 ? 4 : 2;
    const int64_t nblocks = ceil_div(ceil_div(n, v), 256);
    if (nblocks > 0x7fffffffLL) {
        set_error("nd_amd_split_complex: array too large for one launch");
        return ND_AMD_EUNSUPPORTED;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    {
        KernelTimer timer(ND_AMD_KERNEL_RELAYOUT, stream);
        if (dtype == ND_AMD_F32)
            hipLaunchKernelGGL((split_complex_kernel<float>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                               static_cast<const float *>(in), static_cast<float *>(out_re),
                               static_cast<float *>(out_im), n);
        else
            hipLaunchKernelGGL((split_complex_kernel<double>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                               static_cast<const double *>(in), static_cast<double *>(out_re),
                               static_cast<double *>(out_im), n);
    }
    ND_HIP_CHECK(hipGetLastError());
    return ND_AMD_OK;
}

// ---- two real arrays -> one interleaved complex array (the inverse of nd_amd_split_complex) -----
namespace nd_amd {
template <typename T>
__global__ void __launch_bounds__(256) merge_complex_kernel(const T *__restrict__ re, const T *__restrict__ im,
                                                            T *__restrict__ out, int64_t n)
{
    constexpr int V = 16 / (int)sizeof(T);
    struct alignas(16) Vec {
        T v[V];
    };
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (i0 + V <= n) {
        const Vec r = *reinterpret_cast<const Vec *>(re + i0);
        const Vec m = *reinterpret_cast<const Vec *>(im + i0);
        Vec a, b;
#pragma unroll
        for (int j = 0; j < V / 2; ++j) {
            a.v[2 * j] = r.v[j];
            a.v[2 * j + 1] = m.v[j];
            b.v[2 * j] = r.v[V / 2 + j];
            b.v[2 * j + 1] = m.v[V / 2 + j];
        }
        *reinterpret_cast<Vec *>(out + 2 * i0) = a;
        *reinterpret_cast<Vec *>(out + 2 * i0 + V) = b;
    } else {
        for (int64_t i = i0; i < n; ++i) {
            out[2 * i] = re[i];
            out[2 * i + 1] = im[i];
        }
    }
}
}  // namespace nd_amd

extern "C" int nd_amd_merge_complex(const void *in_re, const void *in_im, void *out, int dtype, int64_t n,
                                    void *hip_stream)
{
    using namespace nd_amd;
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_merge_complex: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (n < 0) {
        set_error("nd_amd_merge_complex: negative length");
        return ND_AMD_EINVAL;
    }
    if (n == 0) return ND_AMD_OK;
    if (!in_re || !in_im || !out) {
        set_error("nd_amd_merge_complex: null data pointer");
        return ND_AMD_EINVAL;
    }
    if ((((uintptr_t)in_re | (uintptr_t)in_im | (uintptr_t)out) & 15) != 0) {
        set_error("nd_amd_merge_complex: pointers must be 16-byte aligned");
        return ND_AMD_EINVAL;
    }
    const int v = dtype == ND_AMD_F32 ? 4 : 2;
    const int64_t nblocks = ceil_div(ceil_div(n, v), 256);
    if (nblocks > 0x7fffffffLL) {
        set_error("nd_amd_merge_complex: array too large for one launch");
        return ND_AMD_EUNSUPPORTED;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    {
        KernelTimer timer(ND_AMD_KERNEL_RELAYOUT, stream);
        if (dtype == ND_AMD_F32)
            hipLaunchKernelGGL((merge_complex_kernel<float>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                               static_cast<const float *>(in_re), static_cast<const float *>(in_im),
                               static_cast<float *>(out), n);
        else
            hipLaunchKernelGGL((merge_complex_kernel<double>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                               static_cast<const double *>(in_re), static_cast<const double *>(in_im),
                               static_cast<double *>(out), n);
    }
    ND_HIP_CHECK(hipGetLastError());
    return ND_AMD_OK;
}
