// nd_amd/csrc/correlate.hip -- kernel convolution / boxcar for gfx950.
// Replaces scipy.ndimage.convolve as called at nd/filters.py:256-267 (ConvolutionFilter,
// BoxcarFilter).  scipy's NI_Correlate semantics, restated:
//   - the host hands over the footprint: non-zero taps of the flipped kernel in C order, with
//     per-axis input offsets (origin shift for even sizes included);
//   - per output element  double tmp = 0;  tmp += w[t] * (double)in[extend(i + off[t])]  in
//     footprint order (multiply and add rounded separately: this TU is built with
//     -ffp-contract=off);  out = (T)tmp;
//   - border handling per axis: reflect / constant / nearest / mirror / wrap
//     (ni_support.c NI_InitFilterOffsets).
//
// Generic kernel: one thread per output element of a 4-D strided view, last axis fastest across
// lanes.  Interior elements (no tap leaves the array) use precomputed linear tap offsets.
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace nd_amd {

constexpr int kTapsArgs = 128;   // taps that travel as a kernel argument

struct Tap {
    int32_t o[4];   // per-axis input offset
    double w;
};

struct TapsByValue {
    Tap t[kTapsArgs];
};

__device__ __forceinline__ int64_t extend_index(int64_t cc, int64_t len, int mode)
{
    if (cc >= 0 && cc < len) return cc;
    switch (mode) {
    case ND_AMD_MODE_REFLECT: {
        if (len <= 1) return 0;
        const int64_t sz2 = 2 * len;
        if (cc < 0) {
            if (cc < -sz2) cc += sz2 * (-cc / sz2);
            return cc < -len ? cc + sz2 : -cc - 1;
        }
        cc -= sz2 * (cc / sz2);
        if (cc >= len) cc = sz2 - cc - 1;
        return cc;
    }
    case ND_AMD_MODE_CONSTANT:
        return -1;
    case ND_AMD_MODE_NEAREST:
        return cc < 0 ? 0 : len - 1;
    case ND_AMD_MODE_MIRROR: {
        if (len <= 1) return 0;
        const int64_t sz2 = 2 * len - 2;
        if (cc < 0) {
            cc = sz2 * (-cc / sz2) + cc;
            return cc <= 1 - len ? cc + sz2 : -cc;
        }
        cc -= sz2 * (cc / sz2);
        if (cc >= len) cc = sz2 - cc;
        return cc;
    }
    case ND_AMD_MODE_WRAP: {
        if (len <= 1) return 0;
        if (cc < 0) {
            cc += len * (-cc / len);
            if (cc < 0) cc += len;
            return cc;
        }
        cc -= len * (cc / len);
        return cc;
    }
    }
    return -1;
}

template <typename T>
struct CorrArgs {
    const T *in;
    T *out;
    int64_t A[4], si[4], so[4];
    int64_t lo[4], hi[4];   // interior box: lo <= i < hi on every axis -> no tap leaves the array
    int64_t total;
    int ntaps, mode;
    double cval;
    const Tap *taps_dev;
};

template <typename T, bool TAPS_ARGS>
__global__ void __launch_bounds__(256) correlate_kernel(const CorrArgs<T> a, const TapsByValue tv)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.total) return;
    int64_t rem = idx;
    const int64_t i3 = rem % a.A[3];
    rem /= a.A[3];
    const int64_t i2 = rem % a.A[2];
    rem /= a.A[2];
    const int64_t i1 = rem % a.A[1];
    const int64_t i0 = rem / a.A[1];
    const bool interior = i0 >= a.lo[0] && i0 < a.hi[0] && i1 >= a.lo[1] && i1 < a.hi[1] &&
                          i2 >= a.lo[2] && i2 < a.hi[2] && i3 >= a.lo[3] && i3 < a.hi[3];
    const int64_t base = i0 * a.si[0] + i1 * a.si[1] + i2 * a.si[2] + i3 * a.si[3];
    double tmp = 0.0;
    for (int t = 0; t < a.ntaps; ++t) {
        const Tap tp = TAPS_ARGS ? tv.t[t] : a.taps_dev[t];
        double v;
        if (interior) {
            v = (double)a.in[base + tp.o[0] * a.si[0] + tp.o[1] * a.si[1] + tp.o[2] * a.si[2] +
                             tp.o[3] * a.si[3]];
        } else {
            const int64_t j0 = extend_index(i0 + tp.o[0], a.A[0], a.mode);
            const int64_t j1 = extend_index(i1 + tp.o[1], a.A[1], a.mode);
            const int64_t j2 = extend_index(i2 + tp.o[2], a.A[2], a.mode);
            const int64_t j3 = extend_index(i3 + tp.o[3], a.A[3], a.mode);
            if (j0 < 0 || j1 < 0 || j2 < 0 || j3 < 0)
                v = a.cval;
            else
                v = (double)a.in[j0 * a.si[0] + j1 * a.si[1] + j2 * a.si[2] + j3 * a.si[3]];
        }
        tmp = tmp + tp.w * v;
    }
    a.out[i0 * a.so[0] + i1 * a.so[1] + i2 * a.so[2] + i3 * a.so[3]] = (T)tmp;
}

template <typename T>
static int correlate_impl(const void *in, void *out, const int64_t dims[4], const int64_t si[4],
                          const int64_t so[4], int64_t ntaps, const int64_t *offsets,
                          const double *weights, int mode, double cval, void *taps_dev,
                          size_t taps_dev_bytes, hipStream_t stream)
{
    CorrArgs<T> a;
    a.in = static_cast<const T *>(in);
    a.out = static_cast<T *>(out);
    a.total = 1;
    for (int d = 0; d < 4; ++d) {
        a.A[d] = dims[d];
        a.si[d] = si[d];
        a.so[d] = so[d];
        a.total *= dims[d];
        int64_t mn = 0, mx = 0;
        for (int64_t t = 0; t < ntaps; ++t) {
            const int64_t o = offsets[4 * t + d];
            if (o < mn) mn = o;
            if (o > mx) mx = o;
            if (o < -0x3fffffff || o > 0x3fffffff) {
                set_error("nd_amd_correlate: tap offset out of range");
                return ND_AMD_EINVAL;
            }
        }
        a.lo[d] = -mn;
        a.hi[d] = dims[d] - mx;
    }
    a.ntaps = (int)ntaps;
    a.mode = mode;
    a.cval = cval;
    a.taps_dev = nullptr;
    if (a.total == 0) return ND_AMD_OK;

    TapsByValue tv;
    memset(&tv, 0, sizeof(tv));
    const bool by_value = ntaps <= kTapsArgs;
    if (by_value) {
        for (int64_t t = 0; t < ntaps; ++t) {
            for (int d = 0; d < 4; ++d) tv.t[t].o[d] = (int32_t)offsets[4 * t + d];
            tv.t[t].w = weights[t];
        }
    } else {
        const size_t need = (size_t)ntaps * sizeof(Tap);
        if (taps_dev == nullptr || taps_dev_bytes < need) {
            set_error("nd_amd_correlate: %lld taps need a %zu-byte device tap buffer (taps_dev)",
                      (long long)ntaps, need);
            return ND_AMD_EWORKSPACE;
        }
        Tap *h = (Tap *)malloc(need);
        if (!h) {
            set_error("nd_amd_correlate: out of host memory");
            return ND_AMD_EINVAL;
        }
        for (int64_t t = 0; t < ntaps; ++t) {
            for (int d = 0; d < 4; ++d) h[t].o[d] = (int32_t)offsets[4 * t + d];
            h[t].w = weights[t];
        }
        hipError_t e = hipMemcpyAsync(taps_dev, h, need, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        free(h);
        ND_HIP_CHECK(e);
        a.taps_dev = static_cast<const Tap *>(taps_dev);
    }
    const int64_t nblocks = ceil_div(a.total, 256);
    if (nblocks > 0x7fffffffLL) {
        set_error("nd_amd_correlate: array too large for one launch");
        return ND_AMD_EUNSUPPORTED;
    }
    {
        KernelTimer timer(ND_AMD_KERNEL_CORRELATE, stream);
        if (by_value)
            hipLaunchKernelGGL((correlate_kernel<T, true>), dim3((unsigned)nblocks), dim3(256), 0,
                               stream, a, tv);
        else
            hipLaunchKernelGGL((correlate_kernel<T, false>), dim3((unsigned)nblocks), dim3(256), 0,
                               stream, a, tv);
    }
    ND_HIP_CHECK(hipGetLastError());
    return ND_AMD_OK;
}

}  // namespace nd_amd

using namespace nd_amd;

extern "C" int nd_amd_correlate(const void *in, void *out, int dtype, const int64_t dims[4],
                                const int64_t in_strides[4], const int64_t out_strides[4],
                                int64_t ntaps, const int64_t *offsets, const double *weights,
                                int mode, double cval, void *taps_dev, size_t taps_dev_bytes,
                                void *hip_stream)
{
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_correlate: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (!dims || !in_strides || !out_strides || ntaps < 0 || (ntaps > 0 && (!offsets || !weights))) {
        set_error("nd_amd_correlate: null argument");
        return ND_AMD_EINVAL;
    }
    for (int d = 0; d < 4; ++d)
        if (dims[d] < 0) {
            set_error("nd_amd_correlate: negative dimension");
            return ND_AMD_EINVAL;
        }
    if (mode < 0 || mode > 4) {
        set_error("nd_amd_correlate: unknown border mode %d", mode);
        return ND_AMD_EINVAL;
    }
    if (dims[0] * dims[1] * dims[2] * dims[3] == 0) return ND_AMD_OK;
    if (!in || !out) {
        set_error("nd_amd_correlate: null data pointer");
        return ND_AMD_EINVAL;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (dtype == ND_AMD_F32)
        return correlate_impl<float>(in, out, dims, in_strides, out_strides, ntaps, offsets,
                                     weights, mode, cval, taps_dev, taps_dev_bytes, stream);
    return correlate_impl<double>(in, out, dims, in_strides, out_strides, ntaps, offsets, weights,
                                  mode, cval, taps_dev, taps_dev_bytes, stream);
}
