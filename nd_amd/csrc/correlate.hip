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
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

// cache policy of the window kernels' 16-byte output stores: 2 = non-temporal (written once, never read
// back by the kernel; boxcar 3 x 3 / 5 x 5 on 24 x 4096^2: 0.632 / 0.640 -> 0.612 / 0.620 ms, the fused
// Gaussian unchanged).  Non-temporal LOADS of the rows measured slower (0.611 -> 0.623 ms).
#ifndef ND_CORR_ST_AUX
#define ND_CORR_ST_AUX 2
#endif

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

// -----------------------------------------------------------------------------------------
// LDS-tiled form for a dense KH x KW window over the last two axes (y, x) of x-contiguous planes
// [batch][y][x] -- the layout of the device stacks, and the shape of BoxcarFilter / ConvolutionFilter
// with dims ('y', 'x').
//
// A 256-thread block computes a 128 x 32 tile of outputs.  The (32+KH-1) x (128+KW-1) input window
// is staged once into LDS as doubles with the border rule already applied (reflect / nearest /
// mirror / wrap are separable per axis), so halo re-reads never leave the CU.  Each thread owns a
// 4 x 4 patch of outputs and walks the input rows of its patch once: a row's 4+KW-1 values are read
// from LDS into registers and feed every (output row, kernel row) pair they belong to.  For one
// output the contributions still arrive kernel-row by kernel-row, left to right -- scipy's footprint
// order -- and each is `tmp = tmp + w * v` with separate rounding of product and sum.
// BOX: all weights equal (boxcar): the product w * v is formed once per input element at staging
// time and the inner loop is additions only (same values, same order, half the FP64 work).
// -----------------------------------------------------------------------------------------
constexpr int kTileX = 128, kTileY = 32, kOX = 4, kOY = 4;
constexpr int kMaxKH = 31;               // (round 6: windows of 17 .. 31 rows / columns; 15 before)
constexpr int kSmallKH = 15;             // the register-array class of the windows up to 15 x 15

template <typename T>
struct TiledArgs {
    const T *in;
    T *out;
    int64_t ny, nx;             // plane size
    int64_t sin_v, sin_t, sin_y;   // element strides of the two batch axes and y (x stride is 1)
    int64_t sout_v, sout_t, sout_y;
    int64_t nt;                 // planes along the second batch axis (the one a 3-D window runs along)
    int kh, oy0, ox0;           // window: input row = y + oy0 + i, input col = x + ox0 + j
    int kt, ot0;                // 3-D windows: kt planes, input plane = t + ot0 + dt (1, 0: a 2-D window)
    int mode;
    int tiles_x, tiles_y;
    int64_t nbatch;             // planes
    int ppb;                    // planes one block walks through
    double w[kMaxKH * kMaxKH];  // dense KT x KH x KW weights, row-major (0 = tap absent)
    double cval;                // mode `constant`: the value of every sample outside the plane
};

// The block keeps its tile position and walks `ppb` consecutive planes of the batch: the border
// maps and every thread's source offsets are computed once, and the loads of plane b+1 are in
// flight (held in registers) while plane b is computed and stored.  KHB bounds the kernel height
// the instantiation can stage (register array sizes).
// 3-D windows (round 5; the reference hands scipy N-D kernels as they are, nd/filters.py:256-267): a
// window of kt planes along the second batch axis is a walk over kt staged planes PER OUTPUT plane --
// plane extend(t + ot0 + dt) for dt = 0 .. kt - 1, the border rule applied to the plane index -- with
// the running sums kept across them: scipy's footprint order is plane-major, so the terms arrive in
// its order.  Same prefetch (the next staged plane's loads in flight while this one is summed).
template <typename T, int KW, bool BOX, int KHB>
__global__ void __launch_bounds__(256) correlate_tiled_kernel(const TiledArgs<T> a)
{
    extern __shared__ __align__(16) unsigned char nd_smem_c[];
    double *tile = reinterpret_cast<double *>(nd_smem_c);
    const int tid = threadIdx.x;
    const int kh = a.kh;
    const int th = kTileY + kh - 1;              // staged rows
    int64_t b = blockIdx.x;
    const int tx = (int)(b % a.tiles_x);
    b /= a.tiles_x;
    const int ty = (int)(b % a.tiles_y);
    const int64_t b0 = (b / a.tiles_y) * a.ppb;
    const int64_t b1 = b0 + a.ppb < a.nbatch ? b0 + a.ppb : a.nbatch;
    const int kt = a.kt;
    const int64_t x_base = (int64_t)tx * kTileX, y_base = (int64_t)ty * kTileY;

    // ---- once per block: the border rule is separable -- one source row per staged row and one
    // source column per staged column go through small LDS tables into per-thread offsets ----
    const double wbox = a.w[0];
    constexpr int TW = kTileX + KW - 1;
    // LDS image: columns de-interleaved by (c mod 4), element (r, c) at (r*4 + (c&3)) * TWQ + (c>>2).
    // A thread's 4 adjacent outputs start at column 4*lane', so the lanes of one read instruction
    // hit consecutive doubles (conflict-free) instead of every fourth one (4-way conflict).
    constexpr int TWQ = (TW + 3) / 4;
    int *ymap = reinterpret_cast<int *>(tile + th * 4 * TWQ);
    int *xmap = ymap + th;
    for (int i = tid; i < th + TW; i += 256) {
        // (scipy's reflect table yields -1 for offsets that are exact multiples of 2*len beyond
        // -2*len; clamp so that such a degenerate window can never index before the plane)
        // mode `constant`: -1 marks a sample outside the plane; it is staged as cval (below) and its
        // load goes to element 0
        if (i < th) {
            const int64_t m = extend_index(y_base + a.oy0 + i, a.ny, a.mode);
            ymap[i] = m < 0 ? (a.mode == ND_AMD_MODE_CONSTANT ? -1 : 0) : (int)m;
        } else {
            const int64_t m = extend_index(x_base + a.ox0 + (i - th), a.nx, a.mode);
            xmap[i - th] = m < 0 ? (a.mode == ND_AMD_MODE_CONSTANT ? -1 : 0) : (int)m;
        }
    }
    __syncthreads();
    // Staging map.  Main part: thread (row parity p = tid / 128, column c = tid % 128) takes rows
    // p, p + 2, ... of column c: the row offsets are wave-uniform (scalar registers), the column
    // offset is one register, and the LDS destinations differ by compile-time constants.  The
    // KW - 1 columns to the right of the 128th are spread over the threads element by element.
    constexpr int NR = (kTileY + KHB - 1 + 1) / 2;
    constexpr int NE = ((kTileY + KHB - 1) * (KW - 1) + 255) / 256;
    const int wrow = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int c0 = tid & 127;
    const bool colok = xmap[c0] >= 0;
    const int xo = colok ? xmap[c0] : 0;
    const int dlane = (c0 & 3) * TWQ + (c0 >> 2);
    int rowoff[NR];                // source offset of the row inside a plane (fits 31 bits, host)
    unsigned long long rowok = 0ull;   // bit i: staged row wrow + 2 i lies inside the plane (wave-uniform)
    static_assert(NR <= 64, "row validity bits");
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = wrow + 2 * i;
        const int m = __builtin_amdgcn_readfirstlane(r < th ? ymap[r] : 0);
        rowoff[i] = m < 0 ? 0 : m * (int)a.sin_y;
        rowok |= m < 0 ? 0ull : (1ull << i);
    }
    int soff2[NE > 0 ? NE : 1], doff2[NE > 0 ? NE : 1];
    unsigned edgeok = 0u;              // bit j: this thread's j-th edge element lies inside the plane
    static_assert(NE <= 32, "edge validity bits");
    const int n_edge = th * (KW - 1);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int e = tid + 256 * j;
        soff2[j] = 0;
        doff2[j] = 0;
        if (e < n_edge) {
            const int r = e / (KW > 1 ? KW - 1 : 1), c = kTileX + (e - r * (KW - 1));
            const bool inside = ymap[r] >= 0 && xmap[c] >= 0;
            soff2[j] = inside ? ymap[r] * (int)a.sin_y + xmap[c] : 0;
            edgeok |= inside ? (1u << j) : 0u;
            doff2[j] = (r * 4 + (c & 3)) * TWQ + (c >> 2);
        }
    }
    const double cval = a.cval;

    T buf[NR], buf2[NE > 0 ? NE : 1];
    bool plane_ok = true;              // the plane in `buf` lies inside the array (mode `constant`: else cval)
    // the plane staged for output plane bb and window plane dt
    auto load_plane = [&](int64_t bb, int dt) {
        const int64_t v = bb / a.nt, t = bb - v * a.nt;
        // (a single tap plane may still lie off the output's own: a sparse kernel whose other planes are zero)
        const int64_t ts = (kt == 1 && a.ot0 == 0) ? t : extend_index(t + a.ot0 + dt, a.nt, a.mode);
        plane_ok = ts >= 0;
        const T *plane = a.in + v * a.sin_v + (ts < 0 ? 0 : ts) * a.sin_t;
#pragma unroll
        for (int i = 0; i < NR; ++i)
            if (wrow + 2 * i < th) buf[i] = (plane + rowoff[i])[xo];
#pragma unroll
        for (int j = 0; j < NE; ++j)
            if (tid + 256 * j < n_edge) buf2[j] = plane[soff2[j]];
    };
    auto store_plane = [&]() {
#pragma unroll
        for (int i = 0; i < NR; ++i)
            if (wrow + 2 * i < th) {
                const double v = (plane_ok && colok && ((rowok >> i) & 1ull)) ? (double)buf[i] : cval;
                tile[(wrow + 2 * i) * 4 * TWQ + dlane] = BOX ? wbox * v : v;
            }
#pragma unroll
        for (int j = 0; j < NE; ++j)
            if (tid + 256 * j < n_edge) {
                const double v = (plane_ok && ((edgeok >> j) & 1u)) ? (double)buf2[j] : cval;
                tile[doff2[j]] = BOX ? wbox * v : v;
            }
    };
    if (b0 < b1) {
        load_plane(b0, 0);
        store_plane();
    }
    __syncthreads();

    const int lx = (tid % 32) * kOX, ly = (tid / 32) * kOY;
    for (int64_t bb = b0; bb < b1; ++bb) {
        // ---- 4 x 4 outputs per thread ----
        double acc[kOY][kOX];
#pragma unroll
        for (int oy = 0; oy < kOY; ++oy)
#pragma unroll
            for (int ox = 0; ox < kOX; ++ox) acc[oy][ox] = 0.0;

        for (int dt = 0; dt < kt; ++dt) {
        const bool last_dt = dt + 1 == kt;
        const bool has_next = !last_dt || bb + 1 < b1;
        if (has_next) load_plane(last_dt ? bb + 1 : bb, last_dt ? 0 : dt + 1);
        const double *wd = a.w + dt * kh * KW;

        for (int r = 0; r < kOY + kh - 1; ++r) {
            double v[kOX + KW - 1];
            const double *src = tile + (ly + r) * 4 * TWQ + (lx >> 2);      // lx is a multiple of 4
#pragma unroll
            for (int c = 0; c < kOX + KW - 1; ++c) v[c] = src[(c & 3) * TWQ + (c >> 2)];
#pragma unroll
            for (int oy = 0; oy < kOY; ++oy) {
                const int i = r - oy;                       // kernel row feeding output row oy
                if (i >= 0 && i < kh) {
#pragma unroll
                    for (int j = 0; j < KW; ++j) {
                        const double w = wd[i * KW + j];
                        if (BOX || w != 0.0) {
#pragma unroll
                            for (int ox = 0; ox < kOX; ++ox) {
                                if (BOX)
                                    acc[oy][ox] = acc[oy][ox] + v[ox + j];
                                else
                                    acc[oy][ox] = acc[oy][ox] + w * v[ox + j];
                            }
                        }
                    }
                }
            }
        }

        if (has_next && !last_dt) {
            __syncthreads();               // every thread is done reading this plane's image
            store_plane();
            __syncthreads();
        }
        }          // dt

        const bool has_next = bb + 1 < b1;
        T *oplane = a.out + (bb / a.nt) * a.sout_v + (bb % a.nt) * a.sout_t;
#pragma unroll
        for (int oy = 0; oy < kOY; ++oy) {
            const int64_t y = y_base + ly + oy;
            if (y < a.ny) {
                T *orow = oplane + y * a.sout_y;
                const int64_t x = x_base + lx;
                if (x + kOX <= a.nx && ((uintptr_t)(orow + x) % (sizeof(T) * kOX)) == 0) {
                    struct alignas(sizeof(T) * kOX) Out4 {
                        T v[kOX];
                    } o;
#pragma unroll
                    for (int ox = 0; ox < kOX; ++ox) o.v[ox] = (T)acc[oy][ox];
                    *reinterpret_cast<Out4 *>(orow + x) = o;
                } else {
#pragma unroll
                    for (int ox = 0; ox < kOX; ++ox)
                        if (x + ox < a.nx) orow[x + ox] = (T)acc[oy][ox];
                }
            }
        }
        if (has_next) {
            __syncthreads();               // every thread is done reading this plane's image
            store_plane();
            __syncthreads();
        }
    }
}

template <typename T, int KW>
static void launch_tiled(const TiledArgs<T> &a, bool box, int64_t nblocks, size_t lds,
                         hipStream_t stream)
{
    const dim3 grid((unsigned)nblocks), block(256);
    if (a.kh <= 5) {
        if (box)
            hipLaunchKernelGGL((correlate_tiled_kernel<T, KW, true, 5>), grid, block, lds, stream, a);
        else
            hipLaunchKernelGGL((correlate_tiled_kernel<T, KW, false, 5>), grid, block, lds, stream, a);
    } else {
        if (box)
            hipLaunchKernelGGL((correlate_tiled_kernel<T, KW, true, kSmallKH>), grid, block, lds, stream, a);
        else
            hipLaunchKernelGGL((correlate_tiled_kernel<T, KW, false, kSmallKH>), grid, block, lds, stream, a);
    }
}

// Windows of 17 .. 31 columns (any height up to 31), and the taller-than-15 windows of fewer columns run as
// 17 wide with absent columns: one register-array class (31 rows), an LDS image of up to 79 KB (two blocks per
// CU).  Before round 6 every window beyond 15 x 15 took the per-element kernel: BoxcarFilter(w=17) on
// 8 x 2048^2 13.4 ms against 0.43 ms for w = 15.
template <typename T, int KW>
static bool launch_tiled_big(const TiledArgs<T> &a, bool box, int64_t nblocks, size_t lds, hipStream_t stream)
{
    const dim3 grid((unsigned)nblocks), block(256);
    if (box) {
        static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&correlate_tiled_kernel<T, KW, true, kMaxKH>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return false;
        hipLaunchKernelGGL((correlate_tiled_kernel<T, KW, true, kMaxKH>), grid, block, lds, stream, a);
    } else {
        static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&correlate_tiled_kernel<T, KW, false, kMaxKH>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return false;
        hipLaunchKernelGGL((correlate_tiled_kernel<T, KW, false, kMaxKH>), grid, block, lds, stream, a);
    }
    return true;
}

// -----------------------------------------------------------------------------------------
// Register-window form for small square windows (3 x 3, 5 x 5, 7 x 7, centred in x) over float32
// planes: no LDS, no barrier.  A WAVE owns a strip of 256 columns (1 KiB per row, line-aligned) and
// walks down a chunk of rows of one plane.  A lane owns four adjacent columns: one 16-byte load per
// input row, converted to double (boxcar: times the weight, once per input element); the KW - 1
// columns it needs from its neighbours arrive through DPP wave shifts of the converted values; the
// strip's own outer neighbours (K / 2 columns on either side) come from one more, two-lane load per
// row.  The KH x (4 + KW - 1) window lives in registers as a ring over the rows (the row loop is
// unrolled KH times, so every ring index is static), the loads of the next KH rows are in flight
// while a row is computed, and every output still receives its KH * KW terms in scipy's footprint
// order with product and sum rounded separately.  Border rule: rows through a scalar index per
// step, columns outside the row through per-lane mapped single-word loads.
// (Measured on 24 x 4096^2: strips of 248 written columns whose ends share cache lines with the
// neighbour strips cost 8 %; non-temporal loads cost 6-10 %, non-temporal stores, deeper read-ahead
// and the chunk height change nothing; the arithmetic is hidden: the kernel runs at the rate of
// its own loads and stores.)
// -----------------------------------------------------------------------------------------
constexpr int kRollStrip = 256;          // columns per wave: 64 lanes x 4

// extend_index in 32-bit arithmetic (extents below 2^30, checked on the host), clamped into the
// array: the row loop holds 2 K inlined copies and the 64-bit divisions made most of its code
__device__ __forceinline__ int extend_index32(int cc, int len, int mode)
{
    if (cc >= 0 && cc < len) return cc;
    int m = 0;
    if (len > 1) {
        switch (mode) {
        case ND_AMD_MODE_REFLECT: {
            const int sz2 = 2 * len;
            if (cc < 0) {
                if (cc < -sz2) cc += sz2 * (-cc / sz2);
                m = cc < -len ? cc + sz2 : -cc - 1;
            } else {
                cc -= sz2 * (cc / sz2);
                m = cc >= len ? sz2 - cc - 1 : cc;
            }
            break;
        }
        case ND_AMD_MODE_NEAREST:
            m = cc < 0 ? 0 : len - 1;
            break;
        case ND_AMD_MODE_MIRROR: {
            const int sz2 = 2 * len - 2;
            if (cc < 0) {
                cc = sz2 * (-cc / sz2) + cc;
                m = cc <= 1 - len ? cc + sz2 : -cc;
            } else {
                cc -= sz2 * (cc / sz2);
                m = cc >= len ? sz2 - cc : cc;
            }
            break;
        }
        case ND_AMD_MODE_WRAP: {
            if (cc < 0) {
                cc += len * (-cc / len);
                if (cc < 0) cc += len;
                m = cc;
            } else {
                m = cc - len * (cc / len);
            }
            break;
        }
        }
    }
    return m < 0 ? 0 : (m >= len ? len - 1 : m);
}

struct RollArgs {
    const float *in;
    float *out;
    int64_t ny, nx;
    int64_t sin_b, sin_y, sout_b, sout_y;
    int oy0, mode;
    int nstrips, nchunks, rows_per_chunk;
    int64_t nbatch;
    int vec_in, vec_out;        // 16-byte accesses allowed (base and strides aligned)
    double w[49];               // dense KH x KW weights, row-major
};

__device__ __forceinline__ double roll_from_prev(double x)      // lane i <- lane i - 1
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double roll_from_next(double x)      // lane i <- lane i + 1
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int K, bool BOX>
__global__ void __launch_bounds__(256, (K == 3 ? 6 : (K == 5 ? 4 : 2))) correlate_roll_kernel(const RollArgs a)
{
    constexpr int H = K / 2, WC = 4 + K - 1;
    const int lane = threadIdx.x & 63;
    int64_t wid = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int strip = (int)(wid % a.nstrips);
    wid /= a.nstrips;
    const int chunk = (int)(wid % a.nchunks);
    const int64_t plane = wid / a.nchunks;
    if (plane >= a.nbatch) return;
    const int64_t ys = (int64_t)chunk * a.rows_per_chunk;
    const int64_t ye = ys + a.rows_per_chunk < a.ny ? ys + a.rows_per_chunk : a.ny;
    const int xs = strip * kRollStrip, x = xs + 4 * lane;
    const int nx = (int)a.nx;

    // Column offsets in bytes.  Own columns: a wave whose 256 columns all lie inside the row takes
    // one 16-byte load per lane, the last strip of a row takes mapped single-word loads.  Outer
    // columns: the K / 2 columns left of the strip are what lane 0 needs, the K / 2 right of it
    // what lane 63 needs; every lane of the lower half-wave issues lane 0's loads and every lane
    // of the upper half lane 63's (two distinct words per instruction, no divergent branch).
    int xm[4], xe[H];
    const bool first = lane == 0, lastl = lane == 63;
#pragma unroll
    for (int c = 0; c < 4; ++c) xm[c] = 4 * extend_index32(x + c, nx, a.mode);
#pragma unroll
    for (int j = 0; j < H; ++j)
        xe[j] = 4 * extend_index32((lane < 32 ? xs - H : xs + kRollStrip) + j, nx, a.mode);
    const bool wave_vec = a.vec_in && xs + kRollStrip <= nx;
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in + plane * a.sin_b), 0,
                                                       0x7fffffff, 0x00020000);
    const auto rout = __builtin_amdgcn_make_buffer_rsrc(a.out + plane * a.sout_b, 0, 0x7fffffff,
                                                        0x00020000);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    struct RowRegs {
        f32x4 v;
        float e[H];
    };
    auto load_row = [&](int64_t r) -> RowRegs {
        RowRegs rr;
        f32x4 v;
        const int m = extend_index32((int)r, (int)a.ny, a.mode);
        const int soff = __builtin_amdgcn_readfirstlane(m * (int)a.sin_y * 4);
        if (wave_vec) {
            v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, xm[0], soff, 0));
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                v[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, xm[c], soff, 0));
        }
#pragma unroll
        for (int j = 0; j < H; ++j)
            rr.e[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, xe[j], soff, 0));
        rr.v = v;
        return rr;
    };

    const double wbox = a.w[0];
    double win[K][WC];
    f32x4 raw[K];
    float rawe[K][H];
    const int nt = (int)(ye - ys) + K - 1;          // input rows this wave consumes
#pragma unroll
    for (int ph = 0; ph < K; ++ph) {
        if (ph < nt) {
            const RowRegs rr = load_row(ys + a.oy0 + ph);
            raw[ph] = rr.v;
#pragma unroll
            for (int j = 0; j < H; ++j) rawe[ph][j] = rr.e[j];
        }
    }

    const bool writer = x < nx;
    const bool store_vec = a.vec_out && x + 3 < nx;

    for (int t0 = 0; t0 < nt; t0 += K) {
#pragma unroll
        for (int ph = 0; ph < K; ++ph) {
            const int t = t0 + ph;
            if (t < nt) {
                // input row t: own columns, then the neighbours' edge columns
                double d[4], de[H];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double v = (double)raw[ph][c];
                    d[c] = BOX ? wbox * v : v;
                }
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    const double ve = (double)rawe[ph][j];
                    de[j] = BOX ? wbox * ve : ve;
                }
                if (t + K < nt) {
                    const RowRegs rr = load_row(ys + a.oy0 + t + K);
                    raw[ph] = rr.v;
#pragma unroll
                    for (int j = 0; j < H; ++j) rawe[ph][j] = rr.e[j];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) win[ph][H + c] = d[c];
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    const double lp = roll_from_prev(d[4 - H + j]), rn = roll_from_next(d[j]);
                    win[ph][j] = first ? de[j] : lp;
                    win[ph][H + 4 + j] = lastl ? de[j] : rn;
                }
                if (t >= K - 1) {
                    // output row: window row i sits in ring slot (ph + 1 + i) mod K
                    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int i = 0; i < K; ++i) {
                        const int sl = (ph + 1 + i) % K;
#pragma unroll
                        for (int j = 0; j < K; ++j) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (BOX)
                                    acc[c] = acc[c] + win[sl][c + j];
                                else
                                    acc[c] = acc[c] + a.w[i * K + j] * win[sl][c + j];
                            }
                        }
                    }
                    const int64_t y = ys + t - (K - 1);
                    const int ooff = __builtin_amdgcn_readfirstlane((int)(y * a.sout_y) * 4);
                    if (writer) {
                        if (store_vec) {
                            const f32x4 o = {(float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rout, x * 4, ooff, ND_CORR_ST_AUX);
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (x + c < nx)
                                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)acc[c]), rout,
                                                                          (x + c) * 4, ooff, 0);
                        }
                    }
                }
            }
        }
    }
}

// Try the register-window form; returns 1 if it was launched.
template <typename T>
static int try_roll(const void *in, void *out, const int64_t dims[4], const int64_t si[4],
                    const int64_t so[4], int64_t ntaps, const int64_t *offsets,
                    const double *weights, int mode, hipStream_t stream)
{
    return 0;
}

template <>
int try_roll<float>(const void *in, void *out, const int64_t dims[4], const int64_t si[4],
                    const int64_t so[4], int64_t ntaps, const int64_t *offsets,
                    const double *weights, int mode, hipStream_t stream)
{
    static const bool disabled = getenv("ND_AMD_NO_ROLL") != nullptr || getenv("ND_AMD_NO_TILED") != nullptr;
    if (disabled || mode == ND_AMD_MODE_CONSTANT) return 0;
    if (ntaps != 9 && ntaps != 25 && ntaps != 49) return 0;
    const int K = ntaps == 9 ? 3 : (ntaps == 25 ? 5 : 7);
    if (si[3] != 1 || so[3] != 1 || si[2] < 0 || so[2] < 0) return 0;
    // dense K x K window in row-major order, centred in x, every weight non-zero
    const int64_t oy0 = offsets[2], ox0 = offsets[3];
    if (ox0 != -(K / 2)) return 0;
    bool box = true;
    for (int64_t t = 0; t < ntaps; ++t) {
        if (offsets[4 * t + 0] != 0 || offsets[4 * t + 1] != 0) return 0;
        if (offsets[4 * t + 2] != oy0 + t / K || offsets[4 * t + 3] != ox0 + t % K) return 0;
        if (weights[t] == 0.0 || !(weights[t] == weights[t])) return 0;
        if (weights[t] != weights[0]) box = false;
    }
    if (!box && K == 7) return 0;            // 49 scalar weights: leave to the LDS form
    const int64_t ny = dims[2], nx = dims[3];
    if (ny < 1 || nx < 8 || ny > 0x3fffffff || nx > 0x3fffffff) return 0;
    {   // scipy's offset table is not periodic far outside the array: leave that to the generic kernel
        const int64_t ry = -oy0 > oy0 + K - 1 ? -oy0 : oy0 + K - 1;
        if (ry >= 2 * ny || K / 2 >= 2 * nx) return 0;
    }
    int64_t nb = dims[0] * dims[1], sbi, sbo;
    if (dims[0] == 1) {
        sbi = si[1];
        sbo = so[1];
    } else if (dims[1] == 1) {
        sbi = si[0];
        sbo = so[0];
    } else if (si[0] == si[1] * dims[1] && so[0] == so[1] * dims[1]) {
        sbi = si[1];
        sbo = so[1];
    } else {
        return 0;
    }
    // row and column offsets inside a plane travel as 32-bit byte offsets
    if ((ny * si[2] + nx) * 4 >= 0x7fffffffLL || (ny * so[2] + nx) * 4 >= 0x7fffffffLL) return 0;
    RollArgs a;
    a.in = static_cast<const float *>(in);
    a.out = static_cast<float *>(out);
    a.ny = ny;
    a.nx = nx;
    a.sin_b = sbi;
    a.sin_y = si[2];
    a.sout_b = sbo;
    a.sout_y = so[2];
    a.oy0 = (int)oy0;
    a.mode = mode;
    a.nbatch = nb;
    a.nstrips = (int)ceil_div(nx, kRollStrip);
    // rows per wave: long enough to amortise the K - 1 rows read ahead of the first output, short
    // enough for >= ~16 k waves
    int64_t rpc = 64;
    while (rpc > 16 && (int64_t)a.nstrips * ceil_div(ny, rpc) * nb < 16384) rpc /= 2;
    a.rows_per_chunk = (int)rpc;
    a.nchunks = (int)ceil_div(ny, rpc);
    a.vec_in = (((uintptr_t)in & 15) == 0 && (sbi & 3) == 0 && (si[2] & 3) == 0) ? 1 : 0;
    a.vec_out = (((uintptr_t)out & 15) == 0 && (sbo & 3) == 0 && (so[2] & 3) == 0) ? 1 : 0;
    for (int i = 0; i < 49; ++i) a.w[i] = i < ntaps ? weights[i] : 0.0;
    const int64_t nwaves = (int64_t)a.nstrips * a.nchunks * nb;
    const int64_t nblocks = ceil_div(nwaves, 4);
    if (nblocks > 0x7fffffffLL) return 0;
    KernelTimer timer(ND_AMD_KERNEL_BOXCAR_TILED, stream);
    const dim3 grid((unsigned)nblocks), block(256);
    if (K == 3) {
        if (box) hipLaunchKernelGGL((correlate_roll_kernel<3, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((correlate_roll_kernel<3, false>), grid, block, 0, stream, a);
    } else if (K == 5) {
        if (box) hipLaunchKernelGGL((correlate_roll_kernel<5, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((correlate_roll_kernel<5, false>), grid, block, 0, stream, a);
    } else {
        hipLaunchKernelGGL((correlate_roll_kernel<7, true>), grid, block, 0, stream, a);
    }
    return 1;
}

// Try the tiled form; returns 1 if it was launched, 0 if the request does not fit it.
template <typename T>
static int try_tiled(const void *in, void *out, const int64_t dims[4], const int64_t si[4],
                     const int64_t so[4], int64_t ntaps, const int64_t *offsets,
                     const double *weights, int mode, double cval, hipStream_t stream)
{
    static const bool disabled = getenv("ND_AMD_NO_TILED") != nullptr;
    if (disabled || ntaps < 1) return 0;
    if (si[3] != 1 || so[3] != 1) return 0;
    int64_t ymin = 0, ymax = 0, xmin = 0, xmax = 0, tmin = 0, tmax = 0;
    for (int64_t t = 0; t < ntaps; ++t) {
        if (offsets[4 * t + 0] != 0) return 0;
        const int64_t ot = offsets[4 * t + 1], oy = offsets[4 * t + 2], ox = offsets[4 * t + 3];
        if (t == 0) {
            tmin = tmax = ot;
            ymin = ymax = oy;
            xmin = xmax = ox;
        }
        tmin = ot < tmin ? ot : tmin;
        tmax = ot > tmax ? ot : tmax;
        ymin = oy < ymin ? oy : ymin;
        ymax = oy > ymax ? oy : ymax;
        xmin = ox < xmin ? ox : xmin;
        xmax = ox > xmax ? ox : xmax;
        // scipy's order is plane-major, then row-major over the window: the dense walk below must visit
        // the taps in the order they were given
        if (t > 0) {
            const int64_t pt = offsets[4 * (t - 1) + 1], py = offsets[4 * (t - 1) + 2], px = offsets[4 * (t - 1) + 3];
            if (ot < pt || (ot == pt && (oy < py || (oy == py && ox <= px)))) return 0;
        }
    }
    // An even-width window (scipy shifts its origin, nd_amd.kernels.footprint) runs as the next odd
    // width with one absent column on the right: absent taps are skipped like scipy's dropped
    // zero weights, so the sums are the same, term for term.
    const int64_t kh = ymax - ymin + 1, kw_taps = xmax - xmin + 1;
    int64_t kw = (kw_taps & 1) ? kw_taps : kw_taps + 1;
    const int64_t kt = tmax - tmin + 1;
    // (taller than 15 rows but narrower than 17 columns: as 17 wide, the columns beyond the window absent)
    const bool big = kh > kSmallKH || kw > kSmallKH;
    if (big && kw < 17) kw = 17;
    if (kh > kMaxKH || kw > kMaxKH || kt * kh * kw > kMaxKH * kMaxKH) return 0;
    if (dims[2] < 1 || dims[3] < 1 || dims[2] > 0x7fffffffLL || dims[3] > 0x7fffffffLL) return 0;
    // windows reaching farther than twice the plane from it hit the non-periodic corner of
    // scipy's offset table: leave those to the generic kernel, which restates the table as is
    {
        const int64_t ry = (-ymin > ymax ? -ymin : ymax), rx = (-xmin > xmax ? -xmin : xmax);
        const int64_t rt = (-tmin > tmax ? -tmin : tmax);
        if (ry >= 2 * dims[2] || rx >= 2 * dims[3] || (rt > 0 && rt >= 2 * dims[1])) return 0;
    }
    // batch = dims[0] x dims[1], addressed as (v, t): a 3-D window runs along the second of them
    const int64_t nb = dims[0] * dims[1];
    if (dims[1] < 1) return 0;
    TiledArgs<T> a;
    a.in = static_cast<const T *>(in);
    a.out = static_cast<T *>(out);
    a.ny = dims[2];
    a.nx = dims[3];
    a.sin_v = si[0];
    a.sin_t = si[1];
    a.sin_y = si[2];
    a.sout_v = so[0];
    a.sout_t = so[1];
    a.sout_y = so[2];
    a.nt = dims[1];
    a.kt = (int)kt;
    a.ot0 = (int)tmin;
    a.kh = (int)kh;
    a.oy0 = (int)ymin;
    a.ox0 = (int)xmin;
    a.mode = mode;
    a.cval = cval;
    a.tiles_x = (int)ceil_div(dims[3], kTileX);
    a.tiles_y = (int)ceil_div(dims[2], kTileY);
    for (int i = 0; i < kMaxKH * kMaxKH; ++i) a.w[i] = 0.0;
    bool box = (ntaps == kt * kh * kw);      // (a window padded to 17 columns is not one: its absent taps are skipped)
    for (int64_t t = 0; t < ntaps; ++t) {
        a.w[((offsets[4 * t + 1] - tmin) * kh + (offsets[4 * t + 2] - ymin)) * kw + (offsets[4 * t + 3] - xmin)] = weights[t];
        if (weights[t] != weights[0]) box = false;
    }
    // source offsets inside one plane are kept as 32-bit integers
    if ((dims[2] - 1) * (si[2] < 0 ? -si[2] : si[2]) + dims[3] > 0x7fffffffLL || si[2] < 0) return 0;
    // each block walks `ppb` planes; keep at least ~2048 blocks in flight for small rasters
    a.nbatch = nb;
    {
        const int64_t tiles = (int64_t)a.tiles_x * a.tiles_y;
        int64_t groups = ceil_div(2048, tiles);
        if (groups > nb) groups = nb;
        if (groups < 1) groups = 1;
        a.ppb = (int)ceil_div(nb, groups);
    }
    const int64_t nblocks = (int64_t)a.tiles_x * a.tiles_y * ceil_div(nb, (int64_t)a.ppb);
    if (nblocks > 0x7fffffffLL || nblocks < 1) return 0;
    const size_t twq = (size_t)(kTileX + kw - 1 + 3) / 4;
    const size_t lds = (size_t)(kTileY + kh - 1) * 4 * twq * sizeof(double) +
                       (size_t)((kTileY + kh - 1) + (kTileX + kw - 1)) * sizeof(int);
    if (lds > (big ? 96 : 64) * 1024) return 0;
    if (big) {
        KernelTimer timer(ND_AMD_KERNEL_BOXCAR_TILED, stream);
        bool ok = false;
        switch (kw) {
        case 17: ok = launch_tiled_big<T, 17>(a, box, nblocks, lds, stream); break;
        case 19: ok = launch_tiled_big<T, 19>(a, box, nblocks, lds, stream); break;
        case 21: ok = launch_tiled_big<T, 21>(a, box, nblocks, lds, stream); break;
        case 23: ok = launch_tiled_big<T, 23>(a, box, nblocks, lds, stream); break;
        case 25: ok = launch_tiled_big<T, 25>(a, box, nblocks, lds, stream); break;
        case 27: ok = launch_tiled_big<T, 27>(a, box, nblocks, lds, stream); break;
        case 29: ok = launch_tiled_big<T, 29>(a, box, nblocks, lds, stream); break;
        default: ok = launch_tiled_big<T, 31>(a, box, nblocks, lds, stream); break;
        }
        return ok ? 1 : 0;
    }
    {
        KernelTimer timer(ND_AMD_KERNEL_BOXCAR_TILED, stream);
        switch (kw) {
        case 1: launch_tiled<T, 1>(a, box, nblocks, lds, stream); break;
        case 3: launch_tiled<T, 3>(a, box, nblocks, lds, stream); break;
        case 5: launch_tiled<T, 5>(a, box, nblocks, lds, stream); break;
        case 7: launch_tiled<T, 7>(a, box, nblocks, lds, stream); break;
        case 9: launch_tiled<T, 9>(a, box, nblocks, lds, stream); break;
        case 11: launch_tiled<T, 11>(a, box, nblocks, lds, stream); break;
        case 13: launch_tiled<T, 13>(a, box, nblocks, lds, stream); break;
        default: launch_tiled<T, 15>(a, box, nblocks, lds, stream); break;
        }
    }
    return 1;
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
    if (try_roll<T>(in, out, dims, si, so, ntaps, offsets, weights, mode, stream) ||
        try_tiled<T>(in, out, dims, si, so, ntaps, offsets, weights, mode, cval, stream)) {
        ND_HIP_CHECK(hipGetLastError());
        return ND_AMD_OK;
    }

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

// -----------------------------------------------------------------------------------------
// NI_Correlate1D along one axis (GaussianFilter).  One thread per output element of the 4-D
// view, last axis fastest across lanes whatever the filtered axis is.
// -----------------------------------------------------------------------------------------
constexpr int kMaxW1D = 255;

// Line extension of NI_ExtendLine (ni_support.c), which the 1-D filters use: exactly periodic
// for any distance from the array (unlike the offset table of the N-D correlate).
//   reflect  d c b a | a b c d | d c b a      (period 2n)
//   mirror     d c b | a b c d | c b a        (period 2n - 2)
//   wrap     a b c d | a b c d | a b c d      nearest: edge value      constant: cval (-1)
__device__ __forceinline__ int64_t extend_line(int64_t cc, int64_t len, int mode)
{
    if (cc >= 0 && cc < len) return cc;
    switch (mode) {
    case ND_AMD_MODE_REFLECT: {
        const int64_t p = 2 * len;
        int64_t m = cc % p;
        if (m < 0) m += p;
        return m < len ? m : p - 1 - m;
    }
    case ND_AMD_MODE_CONSTANT:
        return -1;
    case ND_AMD_MODE_NEAREST:
        return cc < 0 ? 0 : len - 1;
    case ND_AMD_MODE_MIRROR: {
        if (len <= 1) return 0;
        const int64_t p = 2 * len - 2;
        int64_t m = cc % p;
        if (m < 0) m += p;
        return m < len ? m : p - m;
    }
    case ND_AMD_MODE_WRAP: {
        int64_t m = cc % len;
        if (m < 0) m += len;
        return m;
    }
    }
    return -1;
}

template <typename T>
struct Corr1dArgs {
    const T *in;
    T *out;
    int64_t A[4], si[4], so[4];
    int64_t total;
    int axis, n, size1, size2, symmetric, mode;
    double cval;
    double w[kMaxW1D];
};

template <typename T>
__global__ void __launch_bounds__(256) correlate1d_kernel(const Corr1dArgs<T> a)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.total) return;
    int64_t i[4];
    int64_t rem = idx;
    i[3] = rem % a.A[3];
    rem /= a.A[3];
    i[2] = rem % a.A[2];
    rem /= a.A[2];
    i[1] = rem % a.A[1];
    i[0] = rem / a.A[1];
    const int64_t len = a.A[a.axis], sax = a.si[a.axis], pos = i[a.axis];
    const int64_t base = i[0] * a.si[0] + i[1] * a.si[1] + i[2] * a.si[2] + i[3] * a.si[3] - pos * sax;
    auto x = [&](int64_t j) -> double {
        const int64_t q = extend_line(pos + j, len, a.mode);
        return q < 0 ? a.cval : (double)a.in[base + q * sax];
    };
    const double *fw = a.w + a.size1;
    double o;
    if (a.symmetric > 0) {
        o = x(0) * fw[0];
        for (int j = -a.size1; j < 0; ++j) o = o + (x(j) + x(-j)) * fw[j];
    } else if (a.symmetric < 0) {
        o = x(0) * fw[0];
        for (int j = -a.size1; j < 0; ++j) o = o + (x(j) - x(-j)) * fw[j];
    } else {
        o = x(a.size2) * fw[a.size2];
        for (int j = -a.size1; j < a.size2; ++j) o = o + x(j) * fw[j];
    }
    a.out[i[0] * a.so[0] + i[1] * a.so[1] + i[2] * a.so[2] + i[3] * a.so[3]] = (T)o;
}

// LDS-tiled form for x-contiguous arrays: a block computes 32 rows x 128 columns of outputs, where
// "rows" run along the filtered axis (ALONG_X = false) or along any other axis (ALONG_X = true, the
// filter then runs along the contiguous axis).  The tile plus the kernel's reach is staged as
// doubles with the line extension applied (NI_ExtendLine keeps the line in a double buffer, cval
// included), so every input element is read from memory once per tile instead of once per tap.
// Like the 2-D tiled kernel the block keeps its tile position and walks `ppb` steps along one of
// the two remaining axes: the extension and every thread's source offsets are computed once, and
// the loads of the next step are in flight while the current one is computed.  A thread owns one
// column and 16 rows; lanes read consecutive doubles (conflict-free).  Taps outermost, the 16
// outputs innermost: each output still receives its terms in the order of the generic kernel.
// (round 6: kernels of up to 97 taps -- Gaussian sigma up to 12 at truncate = 4; 31 taps before, beyond which the
//  per-element kernel took over: sigma = 5 on 8 x 2048^2 1.95 ms against 0.31 ms for sigma = 3)
constexpr int kT1X = 128, kT1R = 32, kT1MaxW = 97;

template <typename T>
struct Corr1dTiledArgs {
    const T *in;
    T *out;
    int64_t nrows, nx;          // extent of the row axis and of the contiguous axis
    int64_t sr_in, sr_out;      // strides of the row axis
    int64_t nb1;                // the two remaining axes; axis b1 is the one a block walks
    int64_t sb0_in, sb1_in, sb0_out, sb1_out;
    int tiles_x, tiles_r;
    int ppb;                    // steps along b1 per block
    int n, size1, size2, symmetric, mode;
    double cval;
    double w[kT1MaxW];
};

// NW: the largest number of weights the instantiation can stage (array sizes, LDS pitch).
template <typename T, bool ALONG_X, int NW>
__global__ void __launch_bounds__(256) correlate1d_tiled_kernel(const Corr1dTiledArgs<T> a)
{
    extern __shared__ __align__(16) unsigned char nd_smem_1[];
    double *tile = reinterpret_cast<double *>(nd_smem_1);
    const int tid = threadIdx.x;
    const int reach = a.n - 1;                                   // size1 + size2
    const int trows = ALONG_X ? kT1R : kT1R + reach;             // staged rows
    const int ucols = ALONG_X ? kT1X + reach : kT1X;             // staged columns in use
    constexpr int tcols = ALONG_X ? kT1X + NW - 1 : kT1X;        // row pitch of the image
    constexpr int kMaxRows = ALONG_X ? kT1R : kT1R + NW - 1;
    int *map = reinterpret_cast<int *>(tile + kMaxRows * tcols); // extended index per staged position
    int64_t b = blockIdx.x;
    const int tx = (int)(b % a.tiles_x);
    b /= a.tiles_x;
    const int tr = (int)(b % a.tiles_r);
    b /= a.tiles_r;
    const int64_t ngroups = (a.nb1 + a.ppb - 1) / a.ppb;
    const int64_t s0 = (b % ngroups) * a.ppb, b0 = b / ngroups;
    const int64_t s1 = s0 + a.ppb < a.nb1 ? s0 + a.ppb : a.nb1;
    const int64_t x0 = (int64_t)tx * kT1X, r0 = (int64_t)tr * kT1R;

    // extended index of every staged position along the filtered axis (-1: cval); positions of
    // the other axis beyond the array are clamped (their outputs are never stored)
    const int nmap = ALONG_X ? ucols : trows;
    const int64_t flen = ALONG_X ? a.nx : a.nrows, f0 = (ALONG_X ? x0 : r0) - a.size1;
    for (int i = tid; i < nmap; i += 256) map[i] = (int)extend_line(f0 + i, flen, a.mode);
    __syncthreads();

    // Staging map: thread (row parity, column) takes every second row of its column.  ALONG_X: the
    // `reach` extra columns right of the 128th go to (row = tid / 32 + 8 i, column = 128 + tid % 32).
    // Offsets are relative to the step's base and fit 31 bits (host); -1 = constant extension.
    constexpr int NR = ALONG_X ? kT1R / 2 : (kT1R + NW - 1 + 1) / 2;
    const int sc = tid & (kT1X - 1), sp = tid >> 7;
    int off[NR];
    {
        const int64_t col = ALONG_X ? (int64_t)map[sc] : (x0 + sc < a.nx ? x0 + sc : a.nx - 1);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int rr = sp + 2 * i;
            off[i] = -1;
            if (rr < trows) {
                const int64_t row = ALONG_X ? (r0 + rr < a.nrows ? r0 + rr : a.nrows - 1) : (int64_t)map[rr];
                if (row >= 0 && col >= 0) off[i] = (int)(row * a.sr_in + col);
            }
        }
    }
    // (the extra columns in groups of 32: NG groups, four rows of each per thread)
    constexpr int NG = ALONG_X ? (NW - 1 + 31) / 32 : 1;
    int off2[4 * NG];
    const int ec = kT1X + (tid & 31), er = tid >> 5;
    if (ALONG_X) {
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            const int ecg = ec + 32 * gq;
            const int64_t ecol = ecg < ucols ? (int64_t)map[ecg] : -1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = er + 8 * i;
                const int64_t row = r0 + rr < a.nrows ? r0 + rr : a.nrows - 1;
                off2[4 * gq + i] = ecol < 0 ? -1 : (int)(row * a.sr_in + ecol);
            }
        }
    }

    T buf[NR], buf2[4 * NG];
    auto load_step = [&](int64_t st) {
        const T *src = a.in + b0 * a.sb0_in + st * a.sb1_in;
#pragma unroll
        for (int i = 0; i < NR; ++i)
            if (sp + 2 * i < trows && off[i] >= 0) buf[i] = src[off[i]];
        if (ALONG_X) {
#pragma unroll
            for (int gq = 0; gq < NG; ++gq)
                if (ec + 32 * gq < ucols) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (off2[4 * gq + i] >= 0) buf2[4 * gq + i] = src[off2[4 * gq + i]];
                }
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int rr = sp + 2 * i;
            if (rr < trows) tile[rr * tcols + sc] = off[i] >= 0 ? (double)buf[i] : a.cval;
        }
        if (ALONG_X) {
#pragma unroll
            for (int gq = 0; gq < NG; ++gq)
                if (ec + 32 * gq < ucols) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        tile[(er + 8 * i) * tcols + ec + 32 * gq] = off2[4 * gq + i] >= 0 ? (double)buf2[4 * gq + i] : a.cval;
                }
        }
    };
    if (s0 < s1) {
        load_step(s0);
        store_step();
    }
    __syncthreads();

    const int c = sc, rpar = sp;
    const int64_t x = x0 + c;
    const double *fw = a.w + a.size1;
    constexpr int step = ALONG_X ? 1 : tcols;             // distance between taps in the image
    const double *p0 = ALONG_X ? tile + rpar * tcols + c + a.size1 : tile + (rpar + a.size1) * tcols + c;
    constexpr int rstep = 2 * tcols;                      // distance between this thread's rows
    constexpr int NO = kT1R / 2;
    for (int64_t st = s0; st < s1; ++st) {
        const bool has_next = st + 1 < s1;
        if (has_next) load_step(st + 1);
        // The thread's 16 rows go in two halves of 8, one after the other (a scheduling barrier
        // keeps the halves apart: 16 outputs with their 32 operands in flight next to the prefetch
        // buffer cost a wave of occupancy).
        T *dst = a.out + b0 * a.sb0_out + st * a.sb1_out + x;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            constexpr int NH = NO / 2;
            const double *ph = p0 + h * NH * rstep;
            double o[NH];
            if (a.symmetric != 0) {
                const double w0 = fw[0];
#pragma unroll
                for (int i = 0; i < NH; ++i) o[i] = ph[i * rstep] * w0;
                if (a.symmetric > 0) {
#pragma unroll 1
                    for (int j = -a.size1; j < 0; ++j) {
                        const double wj = fw[j];
                        const double *pa = ph + j * step, *pb = ph - j * step;
#pragma unroll
                        for (int i = 0; i < NH; ++i) o[i] = o[i] + (pa[i * rstep] + pb[i * rstep]) * wj;
                    }
                } else {
#pragma unroll 1
                    for (int j = -a.size1; j < 0; ++j) {
                        const double wj = fw[j];
                        const double *pa = ph + j * step, *pb = ph - j * step;
#pragma unroll
                        for (int i = 0; i < NH; ++i) o[i] = o[i] + (pa[i * rstep] - pb[i * rstep]) * wj;
                    }
                }
            } else {
                const double w0 = fw[a.size2];
                const double *pz = ph + a.size2 * step;
#pragma unroll
                for (int i = 0; i < NH; ++i) o[i] = pz[i * rstep] * w0;
#pragma unroll 1
                for (int j = -a.size1; j < a.size2; ++j) {
                    const double wj = fw[j];
                    const double *pa = ph + j * step;
#pragma unroll
                    for (int i = 0; i < NH; ++i) o[i] = o[i] + pa[i * rstep] * wj;
                }
            }
            if (x < a.nx) {
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    const int64_t r = r0 + rpar + 2 * (h * NH + i);
                    if (r < a.nrows) dst[r * a.sr_out] = (T)o[i];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (has_next) {
            __syncthreads();
            store_step();
            __syncthreads();
        }
    }
}

template <typename T>
static int correlate1d_impl(const void *in, void *out, const int64_t dims[4], const int64_t si[4],
                            const int64_t so[4], int axis, int n, const double *weights, int mode,
                            double cval, hipStream_t stream)
{
    Corr1dArgs<T> a;
    a.in = static_cast<const T *>(in);
    a.out = static_cast<T *>(out);
    a.total = 1;
    for (int d = 0; d < 4; ++d) {
        a.A[d] = dims[d];
        a.si[d] = si[d];
        a.so[d] = so[d];
        a.total *= dims[d];
    }
    a.axis = axis;
    a.n = n;
    a.size1 = n / 2;
    a.size2 = n - a.size1 - 1;
    a.mode = mode;
    a.cval = cval;
    for (int j = 0; j < kMaxW1D; ++j) a.w[j] = j < n ? weights[j] : 0.0;
    // symmetry test of NI_Correlate1D
    a.symmetric = 0;
    if (n & 1) {
        a.symmetric = 1;
        for (int ii = 1; ii <= n / 2; ++ii)
            if (fabs(weights[ii + a.size1] - weights[a.size1 - ii]) > 2.220446049250313e-16) {
                a.symmetric = 0;
                break;
            }
        if (a.symmetric == 0) {
            a.symmetric = -1;
            for (int ii = 1; ii <= n / 2; ++ii)
                if (fabs(weights[a.size1 + ii] + weights[a.size1 - ii]) > 2.220446049250313e-16) {
                    a.symmetric = 0;
                    break;
                }
        }
    }
    if (a.total == 0) return ND_AMD_OK;
    static const bool no_tiled = getenv("ND_AMD_NO_TILED") != nullptr;
    if (!no_tiled && n <= kT1MaxW && si[3] == 1 && so[3] == 1) {
        // rows = the filtered axis, or (filter along the contiguous axis) axis 2; the two
        // remaining axes form the batch
        Corr1dTiledArgs<T> t;
        const bool along_x = (axis == 3);
        const int rax = along_x ? 2 : axis;
        int bax[2], nbx = 0;
        for (int d = 0; d < 3; ++d)
            if (d != rax) bax[nbx++] = d;
        t.in = a.in;
        t.out = a.out;
        t.nrows = dims[rax];
        t.nx = dims[3];
        t.sr_in = si[rax];
        t.sr_out = so[rax];
        t.nb1 = dims[bax[1]];
        t.sb0_in = si[bax[0]];
        t.sb1_in = si[bax[1]];
        t.sb0_out = so[bax[0]];
        t.sb1_out = so[bax[1]];
        t.tiles_x = (int)ceil_div(dims[3], kT1X);
        t.tiles_r = (int)ceil_div(dims[rax], kT1R);
        t.n = n;
        t.size1 = a.size1;
        t.size2 = a.size2;
        t.symmetric = a.symmetric;
        t.mode = mode;
        t.cval = cval;
        for (int j = 0; j < kT1MaxW; ++j) t.w[j] = j < n ? weights[j] : 0.0;
        // walk the longer of the two remaining axes; keep >= ~2048 blocks in flight
        if (dims[bax[0]] > dims[bax[1]]) {
            const int tmp = bax[0];
            bax[0] = bax[1];
            bax[1] = tmp;
            t.nb1 = dims[bax[1]];
            t.sb0_in = si[bax[0]];
            t.sb1_in = si[bax[1]];
            t.sb0_out = so[bax[0]];
            t.sb1_out = so[bax[1]];
        }
        int64_t groups = 1;
        {
            const int64_t base_blocks = (int64_t)t.tiles_x * t.tiles_r * dims[bax[0]];
            groups = ceil_div(2048, base_blocks < 1 ? 1 : base_blocks);
            if (groups > t.nb1) groups = t.nb1;
            if (groups < 1) groups = 1;
            t.ppb = (int)ceil_div(t.nb1, groups);
            if (t.ppb < 1) t.ppb = 1;
            groups = ceil_div(t.nb1, (int64_t)t.ppb);
        }
        const int64_t nb = (int64_t)t.tiles_x * t.tiles_r * dims[bax[0]] * groups;
        // size class of the instantiation
        const int nwc = n <= 9 ? 9 : (n <= 17 ? 17 : (n <= 25 ? 25 : (n <= 31 ? 31 : (n <= 65 ? 65 : kT1MaxW))));
        const size_t trows = along_x ? kT1R : kT1R + nwc - 1, tcols = along_x ? kT1X + nwc - 1 : kT1X;
        const size_t lds = trows * tcols * sizeof(double) + (along_x ? tcols : trows) * sizeof(int);
        bool fits = nb <= 0x7fffffffLL && lds <= 150 * 1024 && si[rax] >= 0 && so[rax] >= 0;
        for (int d = 0; d < 4; ++d)
            if (dims[d] > 0x3fffffffLL) fits = false;
        // offsets inside one step (row axis x contiguous axis) are kept as 32-bit integers
        if ((dims[rax] - 1) * si[rax] + dims[3] > 0x7fffffffLL) fits = false;
        if (fits) {
            KernelTimer timer(ND_AMD_KERNEL_CORRELATE1D, stream);
            const dim3 grid((unsigned)nb), block(256);
#define ND_LAUNCH_1D(AX, NWC)                                                                                       \
    do {                                                                                                            \
        if (lds > 64 * 1024) {                                                                                      \
            static const hipError_t e_ = hipFuncSetAttribute(                                                       \
                reinterpret_cast<const void *>(&correlate1d_tiled_kernel<T, AX, NWC>),                              \
                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);                                            \
            ND_HIP_CHECK(e_);                                                                                       \
        }                                                                                                           \
        hipLaunchKernelGGL((correlate1d_tiled_kernel<T, AX, NWC>), grid, block, lds, stream, t);                    \
    } while (0)
            if (along_x) {
                if (nwc == 9) ND_LAUNCH_1D(true, 9);
                else if (nwc == 17) ND_LAUNCH_1D(true, 17);
                else if (nwc == 25) ND_LAUNCH_1D(true, 25);
                else if (nwc == 31) ND_LAUNCH_1D(true, 31);
                else if (nwc == 65) ND_LAUNCH_1D(true, 65);
                else ND_LAUNCH_1D(true, kT1MaxW);
            } else {
                if (nwc == 9) ND_LAUNCH_1D(false, 9);
                else if (nwc == 17) ND_LAUNCH_1D(false, 17);
                else if (nwc == 25) ND_LAUNCH_1D(false, 25);
                else if (nwc == 31) ND_LAUNCH_1D(false, 31);
                else if (nwc == 65) ND_LAUNCH_1D(false, 65);
                else ND_LAUNCH_1D(false, kT1MaxW);
            }
#undef ND_LAUNCH_1D
            ND_HIP_CHECK(hipGetLastError());
            return ND_AMD_OK;
        }
    }
    const int64_t nblocks = ceil_div(a.total, 256);
    if (nblocks > 0x7fffffffLL) {
        set_error("nd_amd_correlate1d: array too large for one launch");
        return ND_AMD_EUNSUPPORTED;
    }
    {
        KernelTimer timer(ND_AMD_KERNEL_CORRELATE1D, stream);
        hipLaunchKernelGGL((correlate1d_kernel<T>), dim3((unsigned)nblocks), dim3(256), 0, stream, a);
    }
    ND_HIP_CHECK(hipGetLastError());
    return ND_AMD_OK;
}

// -----------------------------------------------------------------------------------------
// Two correlate1d passes in one kernel: along y, then along x, over x-contiguous float32 planes --
// scipy.ndimage.gaussian_filter on dims ('y', 'x') (nd/filters.py:365-378), which filters axis
// after axis and rounds the intermediate array to the array dtype.  Symmetric kernels of the same
// radius R in both directions.  Same register-window design as correlate_roll_kernel: a wave owns
// a strip of columns and walks down a chunk of rows; a lane owns four columns.  The last 2 R + 1
// input rows (and the rows read ahead) sit in a ring of float32 registers; per output row the lane
// runs NI_Correlate1D's symmetric form down its four columns in double (outermost pair first),
// rounds to float32 -- the intermediate array -- and the pass along x takes its R neighbours on
// either side from the adjacent lanes through DPP wave shifts of those float32 values.  Lanes at
// the ends of a wave (FEED on either side) only feed their neighbours.  Columns and rows outside
// the plane are scipy's periodic line extension as index maps: the first pass evaluated at a mapped
// column IS the extended intermediate line.  One read and one write of the array instead of two.
// -----------------------------------------------------------------------------------------
__device__ __forceinline__ int extend_line32(int cc, int len, int mode)      // clamped into the line
{
    if (cc >= 0 && cc < len) return cc;
    int m = 0;
    if (len > 1) {
        switch (mode) {
        case ND_AMD_MODE_REFLECT: {
            const int p = 2 * len;
            m = cc % p;
            if (m < 0) m += p;
            m = m < len ? m : p - 1 - m;
            break;
        }
        case ND_AMD_MODE_NEAREST:
            m = cc < 0 ? 0 : len - 1;
            break;
        case ND_AMD_MODE_MIRROR: {
            const int p = 2 * len - 2;
            m = cc % p;
            if (m < 0) m += p;
            m = m < len ? m : p - m;
            break;
        }
        case ND_AMD_MODE_WRAP: {
            m = cc % len;
            if (m < 0) m += len;
            break;
        }
        }
    }
    return m < 0 ? 0 : (m >= len ? len - 1 : m);
}

template <typename T>
struct Corr2PassArgs {
    const T *in;
    T *out;
    int64_t ny, nx;
    int64_t sin_b, sin_y, sout_b, sout_y;
    int mode;
    int nstrips, nchunks, rows_per_chunk;
    int64_t nbatch;
    int vec_in, vec_out;
    double wy[9], wx[9];        // w[d] = weight at distance d from the centre
};

__device__ __forceinline__ float f32_from_prev(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float f32_from_next(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xF, 0xF, true));
}

// float64 planes (round 6): the same kernel on doubles -- a lane's four columns are two 16-byte loads, the
// intermediate array is float64 itself (scipy rounds it to the ARRAY dtype), a value crosses lanes as two DPP moves.
__device__ __forceinline__ double f32_from_prev(double x)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x138, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x138, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double f32_from_next(double x)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x130, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x130, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <typename T, int R, bool DWIN>
__global__ void __launch_bounds__(256) correlate1d_yx_kernel(const Corr2PassArgs<T> a)
{
    static_assert(sizeof(T) == 4 || !DWIN, "float64 rows are their own window");
    constexpr int ES = (int)sizeof(T);
    constexpr int FEED = (R + 3) / 4;               // feeder lanes at either end of the wave
    constexpr int NY = 2 * R + 1, PD = 3;
    // DWIN: the window of 2 R + 1 rows is kept as float64 (a row is converted once, when it enters
    // the window, instead of once per output row it contributes to); only the PD rows in flight
    // stay float32.  WN window slots: the next multiple of PD, so that both slot indices are
    // compile-time constants under one unrolling
    constexpr int WN = DWIN ? ((NY + PD - 1) / PD) * PD : 1;
    constexpr int RING = DWIN ? WN : NY + PD;
    constexpr int SW = (64 - 2 * FEED) * 4;         // columns written per wave
    const int lane = threadIdx.x & 63;
    int64_t wid = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int strip = (int)(wid % a.nstrips);
    wid /= a.nstrips;
    const int chunk = (int)(wid % a.nchunks);
    const int64_t plane = wid / a.nchunks;
    if (plane >= a.nbatch) return;
    const int ys = chunk * a.rows_per_chunk, ny = (int)a.ny, nx = (int)a.nx;
    const int ye = ys + a.rows_per_chunk < ny ? ys + a.rows_per_chunk : ny;
    const int x = strip * SW - 4 * FEED + 4 * lane;

    int xm[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xm[c] = ES * extend_line32(x + c, nx, a.mode);
    const bool inside = x >= 0 && x + 3 < nx;
    const bool wave_vec = a.vec_in && __builtin_amdgcn_ballot_w64(!inside) == 0ull;
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(a.in + plane * a.sin_b), 0,
                                                       0x7fffffff, 0x00020000);
    const auto rout = __builtin_amdgcn_make_buffer_rsrc(a.out + plane * a.sout_b, 0, 0x7fffffff,
                                                        0x00020000);
    typedef T f32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    auto load_row = [&](int t) -> f32x4 {            // input row of step t: ys - R + t
        const int m = extend_line32(ys - R + t, ny, a.mode);
        const int soff = __builtin_amdgcn_readfirstlane(m * (int)a.sin_y * ES);
        f32x4 v;
        if constexpr (sizeof(T) == 4) {
            if (wave_vec) {
                v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, xm[0], soff, 0));
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    v[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, xm[c], soff, 0));
            }
        } else {
            if (wave_vec) {
                const f64x2 lo = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rin, xm[0], soff, 0));
                const f64x2 hi = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rin, xm[0] + 16, soff, 0));
                v[0] = lo[0];
                v[1] = lo[1];
                v[2] = hi[0];
                v[3] = hi[1];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    v[c] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rin, xm[c], soff, 0));
            }
        }
        return v;
    };

    f32x4 ring[DWIN ? PD : RING];
    double dwin[WN][4];
    const int nt = (ye - ys) + 2 * R;               // input rows this wave consumes
#pragma unroll
    for (int t = 0; t < PD; ++t)
        if (t < nt) ring[t] = load_row(t);

    const bool writer = lane >= FEED && lane < 64 - FEED && x < nx;
    const bool store_vec = a.vec_out && x + 3 < nx;

    for (int t0 = 0; t0 < nt; t0 += RING) {
#pragma unroll
        for (int pu = 0; pu < RING; ++pu) {
            const int t = t0 + pu;
            if (t < nt) {
                if constexpr (DWIN) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) dwin[pu][c] = (double)ring[pu % PD][c];
                    if (t + PD < nt) ring[pu % PD] = load_row(t + PD);
                } else {
                    if (t + PD < nt) ring[(pu + PD) % RING] = load_row(t + PD);
                }
                if (t >= 2 * R) {
                    // ---- pass along y on this lane's four columns: rows t - 2R .. t, centre t - R
                    T ty[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        double o;
                        if constexpr (DWIN) {
                            o = dwin[(pu + RING - R) % RING][c] * a.wy[0];
#pragma unroll
                            for (int d = R; d >= 1; --d)
                                o = o + (dwin[(pu + 2 * RING - R - d) % RING][c] +
                                         dwin[(pu + RING - R + d) % RING][c]) * a.wy[d];
                        } else {
                            o = (double)ring[(pu + RING - R) % RING][c] * a.wy[0];
#pragma unroll
                            for (int d = R; d >= 1; --d)
                                o = o + ((double)ring[(pu + 2 * RING - R - d) % RING][c] +
                                         (double)ring[(pu + RING - R + d) % RING][c]) * a.wy[d];
                        }
                        ty[c] = (T)o;               // the intermediate array, in the array dtype
                    }
                    // ---- pass along x: R values from either side through the neighbouring lanes
                    double ax[4 + 2 * R];
#pragma unroll
                    for (int c = 0; c < 4; ++c) ax[R + c] = (double)ty[c];
                    {
                        T pl[4], nr[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            pl[c] = f32_from_prev(ty[c]);
                            nr[c] = f32_from_next(ty[c]);
                        }
                        constexpr int R1 = R < 4 ? R : 4;
#pragma unroll
                        for (int j = 0; j < R1; ++j) {
                            ax[R - 1 - j] = (double)pl[3 - j];
                            ax[R + 4 + j] = (double)nr[j];
                        }
                        if (R > 4) {
#pragma unroll
                            for (int j = 0; j < R - 4; ++j) {
                                ax[R - 5 - j] = (double)f32_from_prev(pl[3 - j]);
                                ax[R + 8 + j] = (double)f32_from_next(nr[j]);
                            }
                        }
                    }
                    T res[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        double o = ax[R + c] * a.wx[0];
#pragma unroll
                        for (int d = R; d >= 1; --d) o = o + (ax[R + c - d] + ax[R + c + d]) * a.wx[d];
                        res[c] = (T)o;
                    }
                    const int y = ys + t - 2 * R;
                    const int ooff = __builtin_amdgcn_readfirstlane(y * (int)a.sout_y * ES);
                    if (writer) {
                        if constexpr (sizeof(T) == 4) {
                            if (store_vec) {
                                const f32x4 o4 = {res[0], res[1], res[2], res[3]};
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), rout, x * 4, ooff, ND_CORR_ST_AUX);
                            } else {
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    if (x + c < nx)
                                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res[c]), rout,
                                                                              (x + c) * 4, ooff, 0);
                            }
                        } else {
                            if (store_vec) {
                                const f64x2 lo = {res[0], res[1]}, hi = {res[2], res[3]};
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), rout, x * 8, ooff, ND_CORR_ST_AUX);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), rout, x * 8 + 16, ooff, ND_CORR_ST_AUX);
                            } else {
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    if (x + c < nx)
                                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, res[c]), rout,
                                                                              (x + c) * 8, ooff, 0);
                            }
                        }
                    }
                }
            }
        }
    }
}

static bool corr1d_symmetric(const double *w, int n)
{
    if (!(n & 1)) return false;
    for (int ii = 1; ii <= n / 2; ++ii)
        if (fabs(w[ii + n / 2] - w[n / 2 - ii]) > 2.220446049250313e-16) return false;
    return true;
}

}  // namespace nd_amd

using namespace nd_amd;

extern "C" int nd_amd_correlate1d_yx(const void *in, void *out, int dtype, const int64_t dims[4],
                                     const int64_t in_strides[4], const int64_t out_strides[4],
                                     int nweights, const double *weights_y, const double *weights_x,
                                     int mode, void *hip_stream)
{
    if (!dims || !in_strides || !out_strides || !weights_y || !weights_x || nweights < 1) {
        set_error("nd_amd_correlate1d_yx: bad argument");
        return ND_AMD_EINVAL;
    }
    if (mode < 0 || mode > 4) {
        set_error("nd_amd_correlate1d_yx: unknown border mode %d", mode);
        return ND_AMD_EINVAL;
    }
    for (int d = 0; d < 4; ++d)
        if (dims[d] < 0) {
            set_error("nd_amd_correlate1d_yx: negative dimension");
            return ND_AMD_EINVAL;
        }
    if (dims[0] * dims[1] * dims[2] * dims[3] == 0) return ND_AMD_OK;
    if (!in || !out || in == out) {
        set_error("nd_amd_correlate1d_yx: null or aliased data pointers");
        return ND_AMD_EINVAL;
    }
    static const bool disabled = getenv("ND_AMD_NO_ROLL") != nullptr || getenv("ND_AMD_NO_TILED") != nullptr;
    const int R = nweights / 2;
    const int64_t *si = in_strides, *so = out_strides;
    const int64_t ny = dims[2], nx = dims[3];
    const int64_t es = dtype == ND_AMD_F64 ? 8 : 4;
    static const bool no_f64 = getenv("ND_AMD_YX_F64") != nullptr && atoi(getenv("ND_AMD_YX_F64")) == 0;
    bool fits = !disabled && (dtype == ND_AMD_F32 || (dtype == ND_AMD_F64 && !no_f64)) && mode != ND_AMD_MODE_CONSTANT &&
                (nweights & 1) &&
                R >= 1 && R <= 8 && R != 7 && corr1d_symmetric(weights_y, nweights) &&
                corr1d_symmetric(weights_x, nweights) && si[3] == 1 && so[3] == 1 && si[2] >= 0 &&
                so[2] >= 0 && ny <= 0x3fffffff && nx <= 0x3fffffff && nx >= 8 &&
                (ny * si[2] + nx) * es < 0x7fffffffLL && (ny * so[2] + nx) * es < 0x7fffffffLL;
    int64_t nb = dims[0] * dims[1], sbi = 0, sbo = 0;
    if (dims[0] == 1) {
        sbi = si[1];
        sbo = so[1];
    } else if (dims[1] == 1) {
        sbi = si[0];
        sbo = so[0];
    } else if (si[0] == si[1] * dims[1] && so[0] == so[1] * dims[1]) {
        sbi = si[1];
        sbo = so[1];
    } else {
        fits = false;
    }
    if (!fits) {
        set_error("nd_amd_correlate1d_yx: not a case of the fused kernel (float32 / float64, x-contiguous planes, "
                  "symmetric kernels of one radius 1..6 or 8, no constant mode): run two nd_amd_correlate1d passes");
        return ND_AMD_EUNSUPPORTED;
    }
    const int feed = (R + 3) / 4, sw = (64 - 2 * feed) * 4;
    const int nstrips = (int)ceil_div(nx, sw);
    // rows per wave: the 2 R rows read ahead of the first output are re-read by the next chunk
    int64_t rpc = R >= 3 ? 128 : 64;
    while (rpc > 16 && (int64_t)nstrips * ceil_div(ny, rpc) * nb < 12288) rpc /= 2;
    static const int rpc_env = getenv("ND_AMD_YX_RPC") ? atoi(getenv("ND_AMD_YX_RPC")) : 0;
    if (rpc_env > 0) rpc = rpc_env;
    const int nchunks = (int)ceil_div(ny, rpc);
    const int64_t nwaves = (int64_t)nstrips * nchunks * nb;
    const int64_t nblocks = ceil_div(nwaves, 4);
    if (nblocks > 0x7fffffffLL) {
        set_error("nd_amd_correlate1d_yx: array too large for one launch");
        return ND_AMD_EUNSUPPORTED;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    auto fill = [&](auto &a) {
        a.ny = ny;
        a.nx = nx;
        a.sin_b = sbi;
        a.sin_y = si[2];
        a.sout_b = sbo;
        a.sout_y = so[2];
        a.mode = mode;
        a.nbatch = nb;
        a.nstrips = nstrips;
        a.rows_per_chunk = (int)rpc;
        a.nchunks = nchunks;
        a.vec_in = (((uintptr_t)in & 15) == 0 && (sbi & 3) == 0 && (si[2] & 3) == 0) ? 1 : 0;
        a.vec_out = (((uintptr_t)out & 15) == 0 && (sbo & 3) == 0 && (so[2] & 3) == 0) ? 1 : 0;
        for (int d = 0; d < 9; ++d) {
            // NI_Correlate1D's symmetric form reads the weights left of the centre: fw[j], j < 0
            a.wy[d] = d <= R ? weights_y[R - d] : 0.0;
            a.wx[d] = d <= R ? weights_x[R - d] : 0.0;
        }
    };
    {
        KernelTimer timer(ND_AMD_KERNEL_CORRELATE1D, stream);
        const dim3 grid((unsigned)nblocks), block(256);
        if (dtype == ND_AMD_F64) {
            Corr2PassArgs<double> a;
            a.in = static_cast<const double *>(in);
            a.out = static_cast<double *>(out);
            fill(a);
#define ND_YX64(RR)                                                                                  \
    case RR: hipLaunchKernelGGL((correlate1d_yx_kernel<double, RR, false>), grid, block, 0, stream, a); break;
            switch (R) {
                ND_YX64(1) ND_YX64(2) ND_YX64(3) ND_YX64(4) ND_YX64(5) ND_YX64(6)
            default: hipLaunchKernelGGL((correlate1d_yx_kernel<double, 8, false>), grid, block, 0, stream, a); break;
            }
#undef ND_YX64
        } else {
            Corr2PassArgs<float> a;
            a.in = static_cast<const float *>(in);
            a.out = static_cast<float *>(out);
            fill(a);
            static const int dwin_env = getenv("ND_AMD_YX_DWIN") ? atoi(getenv("ND_AMD_YX_DWIN")) : -1;
            // measured on 24 x 4096 x 4096 (tools/exp_gauss_yx.py): the float64 window wins from radius 6
            // (sigma 1.5: 0.90 -> 0.85 ms, sigma 2: 1.14 -> 0.98 ms) and loses below (registers: 118 vs 82
            // at radius 4)
            const bool dw = dwin_env >= 0 ? dwin_env != 0 : R >= 6;
#define ND_YX(RR)                                                                                  \
    case RR:                                                                                       \
        if (dw) hipLaunchKernelGGL((correlate1d_yx_kernel<float, RR, true>), grid, block, 0, stream, a);   \
        else hipLaunchKernelGGL((correlate1d_yx_kernel<float, RR, false>), grid, block, 0, stream, a);     \
        break;
            switch (R) {
                ND_YX(1) ND_YX(2) ND_YX(3) ND_YX(4) ND_YX(5) ND_YX(6)
            default:
                if (dw) hipLaunchKernelGGL((correlate1d_yx_kernel<float, 8, true>), grid, block, 0, stream, a);
                else hipLaunchKernelGGL((correlate1d_yx_kernel<float, 8, false>), grid, block, 0, stream, a);
                break;
            }
#undef ND_YX
        }
    }
    ND_HIP_CHECK(hipGetLastError());
    return ND_AMD_OK;
}

extern "C" int nd_amd_correlate1d(const void *in, void *out, int dtype, const int64_t dims[4],
                                  const int64_t in_strides[4], const int64_t out_strides[4],
                                  int axis, int nweights, const double *weights, int mode,
                                  double cval, void *hip_stream)
{
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_correlate1d: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (!dims || !in_strides || !out_strides || !weights || axis < 0 || axis > 3 || nweights < 1) {
        set_error("nd_amd_correlate1d: bad argument");
        return ND_AMD_EINVAL;
    }
    if (nweights > kMaxW1D) {
        set_error("nd_amd_correlate1d: %d weights exceed the limit of %d", nweights, kMaxW1D);
        return ND_AMD_EUNSUPPORTED;
    }
    if (mode < 0 || mode > 4) {
        set_error("nd_amd_correlate1d: unknown border mode %d", mode);
        return ND_AMD_EINVAL;
    }
    for (int d = 0; d < 4; ++d)
        if (dims[d] < 0) {
            set_error("nd_amd_correlate1d: negative dimension");
            return ND_AMD_EINVAL;
        }
    if (dims[0] * dims[1] * dims[2] * dims[3] == 0) return ND_AMD_OK;
    if (!in || !out || in == out) {
        set_error("nd_amd_correlate1d: null or aliased data pointers");
        return ND_AMD_EINVAL;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (dtype == ND_AMD_F32)
        return correlate1d_impl<float>(in, out, dims, in_strides, out_strides, axis, nweights,
                                       weights, mode, cval, stream);
    return correlate1d_impl<double>(in, out, dims, in_strides, out_strides, axis, nweights, weights,
                                    mode, cval, stream);
}


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
