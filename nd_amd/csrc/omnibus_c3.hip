// nd_amd/csrc/omnibus_c3.hip -- OmnibusTest for full-pol (3 x 3 complex Hermitian) covariance
// stacks on gfx950: BASELINE.json config "OmnibusTest full-pol C3, 48t x 8192 x 8192".
//
// EXTENSION: the reference implements dual pol only (p = 2 is hard-coded, nd/_change.pyx:51, 99,
// 135).  This applies the same algorithm (nd/_change.pyx:46-77, 133-151, 224-257) with p = 3: the
// generic formulas `_f`, `_rho`, `_omega2` (nd/_change.pyx:20-39), the same rounding points, and
// det C = abc - a|z|^2 - b|y|^2 - c|x|^2 + 2 Re(x z conj(y)) evaluated left to right in `floating`.
// There is no reference implementation to compare with: parity is pinned only against this
// repository's own generic-p oracle (oracle/nd_oracle_impl.h) -- DESIGN.md marks it UNPINNED.
//
// Planes (all sharing one set of element strides):
//   [C11, C22, C33, C12re, C12im, C13re, C13im, C23re, C23im]
// 1728 B per pixel at k = 48 float32: no thread retains that.  Sparse regime (alpha >= 0.75), float32 up
// to 64 dates (round 6): FOUR WAVES SHARE A PIXEL'S TIME AXIS (omnibus_c3_retain_kernel), screen the
// re-associated whole-series statistic and dump the candidates' series from their registers; pass B
// (omnibus_c3_search_rounds_kernel) walks the dump, blocked by 64 series, in lockstep rounds and runs the
// one-sweep-per-segment search of the dual-pol kernel.  Elsewhere in the sparse regime pass A streams the planes in chunks of
// dates without retention (omnibus_c3_global_kernel) and pass B gathers a listed pixel's series from the
// planes (LDS-staged when it fits).  Low thresholds: the search fused into a streaming pass
// (omnibus_c3_stream_kernel), between 0.02 and the sparse regime the chain search in two streaming passes
// (omnibus_c3_stream_chain_kernel).  f = 9 (j-1) is odd for even j, so a = f/2 can be a half-integer: the
// chi-square pair handles both (omnibus_common.hpp).
#include <type_traits>

#include "omnibus_common.hpp"

namespace nd_amd {

constexpr int kC3Threads = 256;
constexpr int kC3Shards = 128;
constexpr int kC3CounterStride = 32;
constexpr int kC3Chunk = 4;          // dates per load chunk in pass A
constexpr size_t kC3CounterBytes = (size_t)kC3Shards * kC3CounterStride * sizeof(uint32_t);

template <typename T>
__device__ __forceinline__ T det3(const T (&v)[9])
{
    const T a = v[0], b = v[1], c = v[2];
    const T xr = v[3], xi = v[4], yr = v[5], yi = v[6], zr = v[7], zi = v[8];
    const T re = (((xr * zr) - (xi * zi)) * yr) + (((xr * zi) + (xi * zr)) * yi);
    return (((((a * b) * c) - (a * ((zr * zr) + (zi * zi)))) - (b * ((yr * yr) + (yi * yi)))) -
            (c * ((xr * xr) + (xi * xi)))) +
           ((T)2 * re);
}

template <typename T>
struct Accum3 {
    T s[9];
    double prod;
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int c = 0; c < 9; ++c) s[c] = 0;
        prod = 1.0;
    }
    __device__ __forceinline__ void step(const T (&v)[9])
    {
        prod = prod * (double)det3<T>(v);
#pragma unroll
        for (int c = 0; c < 9; ++c) s[c] = s[c] + v[c];
    }
};

template <typename T>
__device__ __forceinline__ T z_stat3(const Accum3<T> &A, int j, double nlooks, const OmniTabEntry &e)
{
    const T det_of_sum = det3<T>(A.s);
    const double logQ =
        nlooks * ((e.pklogk + log(A.prod)) - ((double)j * log((double)det_of_sum)));
    return (T)(e.m2rho * logQ);
}

template <typename T>
__device__ __forceinline__ double z_approx3(const Accum3<T> &A, int j, double nlooks, double m2rho,
                                            double pklogk)
{
    const T det_of_sum = det3<T>(A.s);
    const double logQ = nlooks * ((pklogk + approx_ln(A.prod)) -
                                  ((double)j * approx_ln((double)det_of_sum)));
    return m2rho * logQ;
}
template <typename T>
__device__ __forceinline__ double z_approx3(const Accum3<T> &A, int j, double nlooks,
                                            const OmniTabEntry &e)
{
    return z_approx3<T>(A, j, nlooks, e.m2rho, e.pklogk);
}

template <typename T>
struct C3Args {
    const T *pl[9];
    int64_t nx, nrows, sy, sx, st, blocks_per_row;
    int64_t nx_orig;          // pixels per row of the raster (list entries are y * nx_orig + x)
    int k, write_tab;
    int off32;                // every element offset of a plane, in bytes, fits 32 bits
    int lane32;               // the byte offsets within a row (x * sx) fit 32 bits, sx >= 0
    double nlooks, alpha;
    OmniTabEntry e;
    uint8_t *change;
    T *z_out, *p_out;
    uint32_t *flag_count, *flag_idx;
    uint32_t seg;
    OmniTabEntry *tab_dev;
    uint32_t starts_max;      // lists of a shard up to this length are searched one lane per segment start
    int pm_vec;               // pixel-major inputs (st = 1, whole 16-byte vectors of dates, aligned runs): pass B
                              // reads the series of a listed pixel with 16-byte loads
    int mult[9];              // element-offset multiplier per plane (pixel-major inputs: 2 for the halves of an
                              // interleaved complex array, else 1): pass B reads plane c at pl[c][o * mult[c]]
    // Series of the listed pixels, written by the time-split pass A (omnibus_c3_retain_kernel) from its
    // registers, BLOCKED for the lockstep sweep of pass B (omnibus_c3_search_rounds_kernel): the 64 series of a block
    // of list entries interleaved by groups of four dates and by component --
    //   value (t, c) of entry i of shard s:  dump[((((s * dump_cap / 64 + i / 64) * G + t / 4) * 9 + c) * 64 + i % 64) * 4 + t % 4],
    //   G = dump_stride / 36 groups per series (dump_stride = 9 x the dates a slot holds, dump_cap a multiple of 64)
    // while i < dump_cap; later entries are gathered from the planes as before.  nullptr: no dump.
    T *dump;
    uint32_t dump_cap, dump_stride;
    float retain_rel;         // 3 (21 k + 40) 2^-24: rounding band of the re-associated sums (see the kernel)
};

// ---- pass A ---------------------------------------------------------------------------------
template <typename T, bool STATS>
__global__ void __launch_bounds__(kC3Threads) omnibus_c3_global_kernel(const C3Args<T> g,
                                                                       const OmniTab tab)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t b = blockIdx.x;
    const int64_t row = b / g.blocks_per_row;
    const int64_t bx = b - row * g.blocks_per_row;
    const int64_t bpx0 = bx * (int64_t)kC3Threads;
    const int64_t x0 = bpx0 + tid;
    const int k = g.k;
    const bool in = x0 < g.nx;

    if (g.write_tab && b == 0)
        for (int j = tid; j <= k; j += kC3Threads) g.tab_dev[j] = tab.e[j];

    Accum3<T> A;
    A.reset();
    {
        const int64_t xc = in ? x0 : g.nx - 1;
        const int64_t off0 = row * g.sy + xc * g.sx;
        for (int t0 = 0; t0 < k; t0 += kC3Chunk) {
            T v[kC3Chunk][9];
#pragma unroll
            for (int tt = 0; tt < kC3Chunk; ++tt)
                if (t0 + tt < k) {
                    const int64_t off = off0 + (int64_t)(t0 + tt) * g.st;
#pragma unroll
                    for (int c = 0; c < 9; ++c) v[tt][c] = __builtin_nontemporal_load(g.pl[c] + off);
                }
#pragma unroll
            for (int tt = 0; tt < kC3Chunk; ++tt)
                if (t0 + tt < k) A.step(v[tt]);
        }
    }

    bool flag;
    if (STATS) {
        const T z = z_stat3<T>(A, k, g.nlooks, g.e);
        double zd[1] = {(double)z}, P1[1], P2[1];
        chisq_pair<1>(zd, 9 * (k - 1), g.e.lgam, P1, P2);
        const T P = combine_P<T>(P1[0], P2[0], g.e.omega2);
        flag = in && ((double)P > g.alpha);
        if (in) {
            const int64_t pix = row * g.nx + x0;
            if (g.z_out) g.z_out[pix] = z;
            if (g.p_out) g.p_out[pix] = P;
        }
    } else {
        flag = in && (z_approx3<T>(A, k, g.nlooks, g.e) >= g.e.zlo_a);
    }

    if (__any(flag)) {
        const unsigned long long m = __ballot(flag);
        const unsigned shard = (unsigned)(b % kC3Shards);
        unsigned base = 0;
        if (lane == 0)
            base = atomicAdd(g.flag_count + shard * kC3CounterStride, (unsigned)__popcll(m));
        base = __shfl(base, 0);
        if (flag)
            g.flag_idx[(size_t)shard * g.seg + base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] =
                (uint32_t)(row * g.nx + x0);
    }

    // zero-fill this block's slice of the change map (issued last, never waited on)
    {
        const int64_t left = g.nx - bpx0;
        const int npx = left > kC3Threads ? kC3Threads : (int)left;
        uint8_t *ob = g.change + (row * g.nx + bpx0) * (int64_t)k;
        const int nb = npx * k;
        int head = (int)((16 - ((uintptr_t)ob & 15)) & 15);
        if (head > nb) head = nb;
        if (tid < head) ob[tid] = 0;
        const int nvec = (nb - head) >> 4;
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        u4 *vz = reinterpret_cast<u4 *>(ob + head);
        const u4 zero = {0u, 0u, 0u, 0u};
        for (int i = tid; i < nvec; i += kC3Threads) __builtin_nontemporal_store(zero, vz + i);
        const int tail0 = head + (nvec << 4);
        if (tail0 + tid < nb) ob[tail0 + tid] = 0;
    }
}

// ---- pass A, time-split and retaining (round 6) ----------------------------------------------------
// The gather of pass B was a third of the full-pol call at the benchmark's threshold: 432 isolated 4-byte
// reads per listed pixel, one 64-byte sector each -- 7.1 GB of sector traffic for 1.9 % of the pixels
// (profiles/r05_traffic.json), all of it data pass A had just read.  One thread cannot retain 432 values;
// FOUR WAVES SHARING A PIXEL'S TIME AXIS can: block = PG groups of 64 pixels x 4 waves, wave w of a group
// loads the KQ dates [w KQ, (w + 1) KQ) of its 64 pixels (9 KQ registers, all loads in flight at once) and
// folds its slice into nine partial sums and a partial product of determinants.  The partials meet in LDS;
// the group's first wave screens the combined value, lists the candidates, and every wave then stores its
// slice of a candidate's series from registers into the dump (16-byte stores, the series of a pixel as one
// contiguous run [t][9]) where pass B reads it with 16-byte loads instead of gathering from the planes.
//
// Pass A is only a screen -- exactness lives in pass B, which folds in the reference's order
// (nd/_change.pyx:64-69) -- but it must not lose a pixel that fires.  What differs from the reference's
// forward fold, and how the screen pays for it:
//   * the nine sums are re-associated ((s0 + s1) + s2) + s3.  With every date positive semi-definite
//     (positive diagonal, non-negative 2 x 2 minors: checked per date, anything else is listed) both
//     summation orders are within (k - 1) u of the exact sums on the diagonal and within
//     (k - 1) u sqrt(s_ii s_jj) on each off-diagonal part, and for D = det3 the bound of the streaming
//     search applies to each: |D' - D| <= (21 k + 40) u abc (evaluation in `floating` included).  The two
//     determinants therefore differ by at most 2 (21 k + 40) u abc; the screen adds
//     |m2rho| n k * 1.02 * rel,  rel = 3 (21 k + 40) u abc / D'  (u = 2^-24), to z_approx and lists the
//     pixel when rel >= 0.01 or the sums leave [2^-26, 2^26] (no underflow inside det3 then).
//   * the product of determinants is p0 p1 p2 p3 of the slices' double products: (k + 3) roundings of 2^-53
//     instead of k -- inside the 10x margin zlo_a already carries -- PROVIDED no prefix product the
//     reference forms leaves the normal range.  Every wave tracks the binary exponents its running product
//     passes through; a pixel whose prefix exponents (slice offsets added) leave +-1000 is listed.
//   * a date whose determinant is NaN or exactly 0 makes the whole-series statistic NaN or infinite: no
//     change anywhere (omnibus_c3_stream_kernel, `dead`) -- not listed, as the plain screen does not.
constexpr int kC3Slices = 4;

// (Measured and not kept: two pixel groups per block -- 2.96 against 2.57 ms; the kernel held to 128 registers
//  for four waves per SIMD, eight values spilled -- 2.83 ms; plain instead of non-temporal loads -- 2.89 ms;
//  blocks renumbered so that an XCD walks a contiguous eighth of the raster -- 2.72 ms; the per-date checks
//  removed -- 2.556 against 2.562 ms: they are free.  DESIGN-EXPERIMENTS.md, round 6.)
template <int KQ>
__global__ void __launch_bounds__(64 * kC3Slices) omnibus_c3_retain_kernel(const C3Args<float> g, const OmniTab tab)
{
    typedef float T;
    constexpr int NW = kC3Slices;
    __shared__ float part_s[NW][9][64];
    __shared__ double part_p[NW][64];
    __shared__ int part_e[NW][3][64];          // lowest / highest exponent the slice's product passes, flags
    __shared__ unsigned long long flag_mask;
    __shared__ unsigned list_base;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b = blockIdx.x;
    const int64_t row = b / g.blocks_per_row;
    const int64_t bx = b - row * g.blocks_per_row;
    const int64_t gpx0 = bx * 64;                                   // first pixel of this group
    const int64_t x0 = gpx0 + lane;
    const int k = g.k;
    const bool in = x0 < g.nx;
    const int64_t xc = in ? x0 : g.nx - 1;
    const int t_lo = w * KQ;

    if (g.write_tab && b == 0)
        for (int j = tid; j <= k; j += 64 * NW) g.tab_dev[j] = tab.e[j];

    // every load of the slice in flight: a uniform 64-bit base per plane and date (row and date are
    // wave-uniform: scalar arithmetic) plus a 32-bit lane offset (a row's extent fits 32 bits of byte
    // offset: g.lane32) -- `global_load_dword v, v_off, s[base]`, whatever the extent of the stack
    T v[KQ][9];
    const unsigned lo = (unsigned)(xc * g.sx) * (unsigned)sizeof(T);
#pragma unroll
    for (int tt = 0; tt < KQ; ++tt) {
        const int t = t_lo + tt < k ? t_lo + tt : k - 1;           // (behind the series: the last date again)
        const int64_t uo = row * g.sy + (int64_t)t * g.st;
#pragma unroll
        for (int c = 0; c < 9; ++c)
            v[tt][c] = __builtin_nontemporal_load(
                reinterpret_cast<const T *>(reinterpret_cast<const char *>(g.pl[c] + uo) + lo));
    }

    // zero-fill a quarter of the group's slice of the change map (issued early, never waited on)
    if (gpx0 < g.nx) {
        const int64_t left = g.nx - gpx0;
        const int npx = left > 64 ? 64 : (int)left;
        uint8_t *ob = g.change + (row * g.nx + gpx0) * (int64_t)k;
        const int nb = npx * k;
        int head = (int)((16 - ((uintptr_t)ob & 15)) & 15);
        if (head > nb) head = nb;
        if (w == 0 && lane < head) ob[lane] = 0;
        const int nvec = (nb - head) >> 4;
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        u4 *vz = reinterpret_cast<u4 *>(ob + head);
        const u4 zero = {0u, 0u, 0u, 0u};
        for (int i = w * 64 + lane; i < nvec; i += 64 * NW) __builtin_nontemporal_store(zero, vz + i);
        const int tail0 = head + (nvec << 4);
        if (w == 0 && tail0 + lane < nb) ob[tail0 + lane] = 0;
    }

    // fold the slice in time order
    T s[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) s[c] = (T)0;
    double prod = 1.0;
    int emin = 1, emax = 1;                        // frexp exponent of 1.0
    bool bad = false, dead = false;
#pragma unroll
    for (int tt = 0; tt < KQ; ++tt) {
        if (t_lo + tt < k) {                       // wave-uniform
            const T(&q)[9] = v[tt];
            const T det = det3<T>(q);
            const T mn12 = (q[0] * q[1]) - ((q[3] * q[3]) + (q[4] * q[4]));
            const T mn13 = (q[0] * q[2]) - ((q[5] * q[5]) + (q[6] * q[6]));
            const T mn23 = (q[1] * q[2]) - ((q[7] * q[7]) + (q[8] * q[8]));
            const T dmin = fminf(fminf(q[0], q[1]), q[2]);
            const T mmin = fminf(fminf(mn12, mn13), mn23);
            bad = bad | !((dmin > (T)0) & (mmin >= (T)0) & (det > (T)0));
            dead = dead | !((det > (T)0) | (det < (T)0));
            prod = prod * (double)det;
            const int e = __builtin_amdgcn_frexp_exp(prod);
            emin = e < emin ? e : emin;
            emax = e > emax ? e : emax;
#pragma unroll
            for (int c = 0; c < 9; ++c) s[c] = s[c] + q[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) part_s[w][c][lane] = s[c];
    part_p[w][lane] = prod;
    part_e[w][0][lane] = emin;
    part_e[w][1][lane] = emax;
    part_e[w][2][lane] = (bad ? 1 : 0) | (dead ? 2 : 0);
    __syncthreads();

    const unsigned shard = (unsigned)(b % kC3Shards);
    if (w == 0) {
        // the combined screen, by the group's first wave
        T S[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) S[c] = part_s[0][c][lane];
        double PP = part_p[0][lane];
        int fl = part_e[0][2][lane];
        int eoff = 0;
        bool range_ok = true;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            if (u * KQ < k) {
                const double pu = part_p[u][lane];
                const int lo = part_e[u][0][lane], hi = part_e[u][1][lane];
                range_ok = range_ok & (lo > -1000) & (hi < 1000) & (eoff + lo > -1000) & (eoff + hi < 1000) &
                           (pu > 0.0) & (pu < INFINITY);
                eoff += __builtin_amdgcn_frexp_exp(pu);
                if (u > 0) {
#pragma unroll
                    for (int c = 0; c < 9; ++c) S[c] = S[c] + part_s[u][c][lane];
                    PP = PP * pu;
                    fl |= part_e[u][2][lane];
                }
            }
        }
        const bool isdead = (fl & 2) != 0;
        bool isbad = ((fl & 1) != 0) | !range_ok;
        const T det_of_sum = det3<T>(S);
        const T abc = (S[0] * S[1]) * S[2];
        const T smin = fminf(fminf(S[0], S[1]), S[2]), smax = fmaxf(fmaxf(S[0], S[1]), S[2]);
        isbad = isbad | !((smin > 1.4901161e-08f) & (smax < 67108864.f));            // 2^-26, 2^26
        const float rel = g.retain_rel * (abc * __builtin_amdgcn_rcpf(det_of_sum));
        isbad = isbad | !((det_of_sum > (T)0) & (rel < 0.01f));
        const double logQ = g.nlooks * ((g.e.pklogk + approx_ln(PP)) - ((double)k * approx_ln((double)det_of_sum)));
        const double za = g.e.m2rho * logQ;
        const double mz = (fabs(g.e.m2rho) * g.nlooks * (double)k * 1.02) * (double)rel;
        const bool flag = in && !isdead && (isbad || (za + mz >= g.e.zlo_a));
        const unsigned long long m = __ballot(flag);
        unsigned base = 0;
        if (m != 0ull) {
            if (lane == 0) base = atomicAdd(g.flag_count + shard * kC3CounterStride, (unsigned)__popcll(m));
            base = __shfl(base, 0);
            if (flag)
                g.flag_idx[(size_t)shard * g.seg + base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] =
                    (uint32_t)(row * g.nx + x0);
        }
        if (lane == 0) {
            flag_mask = m;
            list_base = base;
        }
    }
    __syncthreads();
    // a candidate's slice leaves the registers: per component and group of four dates one 16-byte piece, into the
    // blocked dump (see C3Args::dump)
    const unsigned long long m = flag_mask;
    if (m != 0ull && t_lo < k) {
        const unsigned pos = list_base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (((m >> lane) & 1ull) && pos < g.dump_cap) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const unsigned G = g.dump_stride / 36u;
            // first group of this wave's slice (KQ is a multiple of 4)
            f4 *dst = reinterpret_cast<f4 *>(g.dump) +
                      ((((size_t)shard * (g.dump_cap >> 6) + (pos >> 6)) * G + (unsigned)(t_lo >> 2)) * 9) * 64 + (pos & 63u);
#pragma unroll
            for (int gq = 0; gq < KQ / 4; ++gq)
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    f4 q;
                    q.x = v[4 * gq + 0][c];
                    q.y = v[4 * gq + 1][c];
                    q.z = v[4 * gq + 2][c];
                    q.w = v[4 * gq + 3][c];
                    dst[(size_t)(gq * 9 + c) * 64] = q;
                }
        }
    }
}

// ---- pass A for the reference's layout ((y, x, time) per variable), sparse regime ---------------------------
// The full-pol counterpart of omnibus_c2_pm_long_kernel (omnibus.hip).  In this layout a pixel's series is a
// contiguous run per variable, so (i) the wave's span of every variable goes into wave-private LDS images by
// LDS-DMA (all transfers in flight at once, no barrier) and every lane folds its pixel's series out of them,
// and (ii) pass B reads a listed pixel's 9 k values as 9 (or, with interleaved complex off-diagonals, 6) runs
// of k -- 27 sectors at 48 dates -- instead of gathering 432 isolated values from planes megabytes apart, which is
// a third of the planar call at the benchmark's threshold.  PXW pixels per wave (the upper lanes idle): images
// of about 28 KB, five waves per CU -- 16 pixels at 48 dates (the fold of 16 lanes is what bounds the kernel:
// 3.2 - 3.4 ms on 48 x 1024 x 8192 against 2.36 ms for the planar pass A -- a wave loads, then folds, and five
// waves per CU do not hide the one behind the other; four lanes per pixel sharing a date's determinant through
// quad broadcasts, 33 instead of 70 instructions per date on all 64 lanes, measured the same: 3.3 - 3.55 ms).
// JOINT: C12, C13, C23 are interleaved complex arrays (re at the even places); otherwise nine real ones.
template <typename T, int N>
struct alignas(sizeof(T) * N) C3Pack {
    T v[N];
};
struct C3PmArgs {
    int img_off[9];           // element offset of each plane's image in the wave's LDS region
};

template <typename T, int PXW, bool STATS, bool JOINT>
__global__ void __launch_bounds__(64) omnibus_c3_pm_kernel(const C3Args<T> g, const OmniTab tab, const C3PmArgs pm)
{
    constexpr int VE = 16 / (int)sizeof(T);
    typedef __attribute__((address_space(1))) unsigned char glb_u8;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    extern __shared__ __align__(16) unsigned char nd_smem3pm[];
    T *img = reinterpret_cast<T *>(nd_smem3pm);
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t px0 = b * PXW;
    const int64_t x0 = px0 + lane;
    const int k = g.k;
    const bool in = lane < PXW && x0 < g.nx;
    const int64_t left = g.nx - px0;
    const int np = left > PXW ? PXW : (int)left;

    // ---- every transfer of the wave in flight ----
    auto stage = [&](int c, int mult) {
        const int wpp = k * mult;                            // elements per pixel in memory
        const int bytes = np * wpp * (int)sizeof(T);        // multiple of 16 (host checks k)
        const unsigned char *src = reinterpret_cast<const unsigned char *>(g.pl[c] + px0 * wpp);
        unsigned char *dst = reinterpret_cast<unsigned char *>(img + pm.img_off[c]);
        for (int c0 = 0; c0 < bytes; c0 += 1024) {
            const int eb = c0 + lane * 16;
            if (eb < bytes) __builtin_amdgcn_global_load_lds((glb_u8 *)(src + eb), (lds_u8 *)(dst + c0), 16, 0, 2);
        }
    };
    stage(0, 1);
    stage(1, 1);
    stage(2, 1);
    if (JOINT) {
        stage(3, 2);
        stage(5, 2);
        stage(7, 2);
    } else {
#pragma unroll
        for (int c = 3; c < 9; ++c) stage(c, 1);
    }
    if (g.write_tab && b == 0)
        for (int j = lane; j <= k; j += 64) g.tab_dev[j] = tab.e[j];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    typedef C3Pack<T, VE> PV;
    Accum3<T> A;
    A.reset();
    const int nv = k / VE;
    {
    // ---- one lane per pixel: fold this lane's series in time order (idle lanes fold the last pixel of the span) ----
    const int own = (lane < np) ? lane : np - 1;
    const PV *im[9];
#pragma unroll
    for (int c = 0; c < 9; ++c)
        im[c] = reinterpret_cast<const PV *>(img + pm.img_off[c] + own * k * ((JOINT && c >= 3) ? 2 : 1));
    for (int u = 0; u < nv; ++u) {
        PV q[9];
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = im[c][u];
        if (JOINT) {
#pragma unroll
            for (int c = 3; c < 9; c += 2) {
                const PV q0 = im[c][2 * u], q1 = im[c][2 * u + 1];      // (re, im) pairs of VE dates
#pragma unroll
                for (int j = 0; j < VE; ++j) {
                    q[c].v[j] = (j < VE / 2 ? q0 : q1).v[2 * (j % (VE / 2))];
                    q[c + 1].v[j] = (j < VE / 2 ? q0 : q1).v[2 * (j % (VE / 2)) + 1];
                }
            }
        } else {
#pragma unroll
            for (int c = 3; c < 9; ++c) q[c] = im[c][u];
        }
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            T v[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) v[c] = q[c].v[j];
            A.step(v);
        }
    }

    }

    bool flag;
    if (STATS) {
        const T z = z_stat3<T>(A, k, g.nlooks, g.e);
        double zd[1] = {(double)z}, P1[1], P2[1];
        chisq_pair<1>(zd, 9 * (k - 1), g.e.lgam, P1, P2);
        const T P = combine_P<T>(P1[0], P2[0], g.e.omega2);
        flag = in && ((double)P > g.alpha);
        if (in) {
            if (g.z_out) g.z_out[x0] = z;
            if (g.p_out) g.p_out[x0] = P;
        }
    } else {
        flag = in && (z_approx3<T>(A, k, g.nlooks, g.e) >= g.e.zlo_a);
    }
    if (__any(flag)) {
        const unsigned long long m = __ballot(flag);
        const unsigned shard = (unsigned)(b % kC3Shards);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(g.flag_count + shard * kC3CounterStride, (unsigned)__popcll(m));
        base = __shfl(base, 0);
        if (flag) g.flag_idx[(size_t)shard * g.seg + base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)x0;
    }
    // zero-fill this wave's slice of the change map
    {
        uint8_t *ob = g.change + px0 * (int64_t)k;
        const int nb = np * k;
        int head = (int)((16 - ((uintptr_t)ob & 15)) & 15);
        if (head > nb) head = nb;
        if (lane < head) ob[lane] = 0;
        const int nvec = (nb - head) >> 4;
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        u4 *vz = reinterpret_cast<u4 *>(ob + head);
        const u4 zero = {0u, 0u, 0u, 0u};
        for (int i = lane; i < nvec; i += 64) __builtin_nontemporal_store(zero, vz + i);
        const int tail0 = head + (nvec << 4);
        if (tail0 + lane < nb) ob[tail0 + lane] = 0;
    }
}

// ---- pass A with the search fused in (low thresholds), streaming form ---------------------------
// The full-pol counterpart of omnibus_c2_stream_kernel (omnibus.hip), for up to 128 dates.  At the
// thresholds users pass (the reference's default alpha = 0.01) nearly every pixel changes at
// nearly every date, and listing every pixel for pass B costs ~25 ns per pixel (54 ms per 2 Mpx at
// k = 48).  Here the dates are consumed as they arrive, last date first, three in flight:
//   per date t (branch-free): determinant of the date (the reference's `floating` arithmetic) and
//     its logarithm as (exponent, fixed-point mantissa log); suffix sums of the nine components in
//     double -> the global test G(t) over ts[t:] from det3 of those sums; the 2- and 3-date sums
//     a_t + a_t+1 (+ a_t+2) in the reference's type and order from a rolling window -> the marginal
//     tests M2(t), M3(t), whose determinants are bit-identical to the reference's.  Each test
//     leaves two bits (fires / undecided) at position t of six 64- or 128-bit masks.
//   walk: per segment start a handful of bit operations; marginals over four and more dates
//     (neither M2 nor M3 fires: ~alpha of the rows) re-read the dates from memory.
// Decisions come from the float32 screen of omnibus_common.hpp (p-agnostic: z = z0 + c L2); whatever
// it cannot decide for certain hands the pixel to pass B, which redoes it exactly.
//
// Global tests use suffix sums in double where the reference sums forward in `floating`.  With
// every date's matrix positive semi-definite (checked per date: positive diagonal, non-negative
// 2 x 2 minors, positive determinant -- otherwise the pixel goes to pass B), the forward sums carry
// relative errors gamma = (n - 1) u on the diagonal and absolute errors gamma sqrt(s_ii s_jj) on
// each off-diagonal part (Cauchy-Schwarz over the dates), and for
//   D = abc - a|z|^2 - b|y|^2 - c|x|^2 + 2 Re(x z conj y)
// the partial derivatives are bounded by bc, ac, ab (diagonal) and 4 sqrt(ab) c ... (off-diagonal,
// |dx| <= sqrt2 gamma sqrt(ab)), which gives |D_ref - D| <= (3 + 3 * 4 sqrt2) gamma abc < 21 gamma abc
// plus < 40 u abc for the reference's own evaluation of D in `floating`:
//   |D_ref - D| <= (21 n + 20) u abc      -> band 1.46 j (21 n + 20) u abc / D in log2 units.
// MW = 1: 64-bit masks (k <= 64); 2: two-word masks, second register set of screen entries, 64-bit
// sum of the mantissa logs (k <= 128) -- as in omnibus_c2_stream_kernel
// 3 x 3 determinants of two sets of sums side by side, in the halves of packed float32 instructions
// (same operations in the same order as det3<float>: the halves are the reference's values bit for bit)
__device__ __forceinline__ f2_t det3_pk(const f2_t (&v)[9])
{
    const f2_t a = v[0], b = v[1], c = v[2];
    const f2_t xr = v[3], xi = v[4], yr = v[5], yi = v[6], zr = v[7], zi = v[8];
    const f2_t re = (((xr * zr) - (xi * zi)) * yr) + (((xr * zi) + (xi * zr)) * yi);
    const f2_t two = {2.f, 2.f};
    return (((((a * b) * c) - (a * ((zr * zr) + (zi * zi)))) - (b * ((yr * yr) + (yi * yi)))) -
            (c * ((xr * xr) + (xi * xi)))) +
           (two * re);
}

constexpr int c3_stream_nj(const int MW) { return MW == 2 ? kDenseMax : 64; }

// Round 4: the per-date code rebuilt like the dual-pol streaming search's (omnibus.hip, round 3) --
//   * 2- and 3-date marginal tests from PRODUCTS of determinants against powers of the sum's
//     determinant (StreamScreen::ca / cb; no logarithm), both tests in the halves of packed float32
//     instructions;
//   * ONE logarithm of a running double product of the determinants for the global test (the
//     reference's own product, formed backwards), guard: its exponents within 900 of each other;
//   * the constants of the global test met at a date by a scalar load from the argument segment
//     (wave-uniform index), no v_readlane;
//   * test results pushed into the masks as m = 2 m + bit (add-with-carry);
//   * a ring of PF + 2 dates read in place: no register moves for the window of dates t + 1, t + 2.
// 275 -> ~190 vector instructions per date of nine planes.
template <typename T, int MW>
__global__ void __launch_bounds__(kC3Threads) omnibus_c3_stream_kernel(const C3Args<T> g, const OmniTab tab,
                                                                       const DenseScreen scr_arg,
                                                                       const StreamScreen<c3_stream_nj(MW)> ss,
                                                                       const int dense_min)
{
    constexpr int PF = 3, NS = PF + 2;
    typedef typename std::conditional<MW == 2, Bits128, unsigned long long>::type MT;
    typedef typename std::conditional<MW == 2, long long, int>::type LmT;
    __shared__ DenseScreenEntry scr_lds[kDenseMax + 1];
    constexpr int kMaxDates = MW == 2 ? 128 : 64;
    __shared__ __align__(16) uint32_t out_img[(kC3Threads / 64) * 16 * kMaxDates];   // store_change_rows_wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t b = blockIdx.x;
    const int64_t row = b / g.blocks_per_row;
    const int64_t bx = b - row * g.blocks_per_row;
    const int64_t bpx0 = bx * (int64_t)kC3Threads;
    const int64_t x0 = bpx0 + tid;
    const int k = g.k;
    const bool in = x0 < g.nx;
    const int64_t xc = in ? x0 : g.nx - 1;                  // idle lanes re-read the last pixel
    const int64_t off0 = row * g.sy + xc * g.sx;
    // (a plane's extent fits 32 bits of byte offset -- checked on the host, g.off32 -- so a load is
    //  `global_load_dword v, v_offset, s[plane]`: one 32-bit addition per DATE instead of one 64-bit
    //  address per plane and date)
    auto load = [&](const int t, T (&v)[9]) {
        const unsigned o = (unsigned)(off0 + (int64_t)t * g.st) * (unsigned)sizeof(T);
#pragma unroll
        for (int c = 0; c < 9; ++c)
            v[c] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(g.pl[c]) + o);
    };
    // ring slot u holds date k - 1 - u at the start; slots NS - 2, NS - 1 the unit matrix behind the series
    T ring[NS][9];
#pragma unroll
    for (int u = 0; u < PF; ++u) load(k - 1 - u > 0 ? k - 1 - u : 0, ring[u]);
#pragma unroll
    for (int u = PF; u < NS; ++u)
#pragma unroll
        for (int c = 0; c < 9; ++c) ring[u][c] = (c < 3) ? (T)1 : (T)0;
    if (tid == 0) {
#pragma unroll 1
        for (int j = 0; j <= k; ++j) scr_lds[j] = scr_arg.e[j];       // (the deep searches look their j up here)
    }
    if (g.write_tab && b == 0)
        for (int j = tid; j <= k; j += kC3Threads) g.tab_dev[j] = tab.e[j];
    __syncthreads();

    // ---- phase 1 ----
    MT gF = mask_zero<MT>(), gC = mask_zero<MT>(), m2F = mask_zero<MT>(), m2C = mask_zero<MT>(),
       m3F = mask_zero<MT>(), m3C = mask_zero<MT>();
    bool bad = false;
    bool dead = false;       // a date whose determinant is NaN or exactly 0 (omnibus.hip)
    double S[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) S[c] = 0.0;
    double PP = 1.0;                           // product of the determinants of ts[t:] (omnibus.hip)
    int emin = 1, emax = 1;
    T det1 = (T)1, prod12 = (T)1;              // det(t + 1);  det(t + 1) * det(t + 2)
    const T dlo = (T)ss.dlo, dhi = (T)ss.dhi;
    const T ca2 = (T)ss.ca.x, cb2 = (T)ss.cb.x, ca3 = (T)ss.ca.y, cb3 = (T)ss.cb.y;

    // q: date t;  d1, d2: dates t + 1, t + 2;  e: the constants of the global test over k - t dates
    auto process = [&](const T (&q)[9], const T (&d1)[9], const T (&d2)[9], const int t, const StreamEntry &e) {
        const T det = det3<T>(q);
        const T mn12 = (q[0] * q[1]) - ((q[3] * q[3]) + (q[4] * q[4]));
        const T mn13 = (q[0] * q[2]) - ((q[5] * q[5]) + (q[6] * q[6]));
        const T mn23 = (q[1] * q[2]) - ((q[7] * q[7]) + (q[8] * q[8]));
        // positive semi-definite dates only (the rounding bound of the suffix sums rests on it); the
        // determinant's range is checked with those of the 2- / 3-date sums
        const T dmin = fmin(fmin(q[0], q[1]), q[2]);
        const T mmin = fmin(fmin(mn12, mn13), mn23);
        bad = bad | !((dmin > (T)0) & (mmin >= (T)0));
        dead = dead | !((det > (T)0) | (det < (T)0));
        PP = PP * (double)det;
#pragma unroll
        for (int c = 0; c < 9; ++c) S[c] += (double)q[c];
        {                                                   // global test of ts[t:], j = k - t
            const int jj = k - t;
            const double dets = det3<double>(S);
            const float df = (float)dets;
            bool okd;
            int es, eP;
            float ms, mP;
            if (sizeof(T) == 4) {
                okd = df > 7.888609052210118e-31f;          // (an infinite df is caught through rel)
                log2_parts(df, es, ms);
            } else {
                okd = (dets > 0.0) & (dets < (double)INFINITY);
                log2_parts(dets, es, ms);
            }
            log2_parts(PP, eP, mP);
            emin = eP < emin ? eP : emin;
            emax = eP > emax ? eP : emax;
            const int E = (eP - e.re) - __mul24(jj, es);
            const float x = (float)E + __builtin_fmaf(-e.jf, ms, mP - e.rf);
            const float qq = (float)((S[0] * S[1]) * S[2]) * __builtin_amdgcn_rcpf(df);
            const float rel = e.cj * qq;                    // 1.46 (21 n + 20) u abc / D
            const float m2 = e.mj * rel;
            bad = bad | !(okd & (rel < 0.01f));
            mask_push(gF, x + m2 < e.a);
            mask_push(gC, x - m2 > e.b);
        }
        if constexpr (sizeof(T) == 4) {                     // marginal tests over 2 and 3 dates
            // the reference's sums in its type and order -- (0 + a_t) + a_t+1 (+ a_t+2) -- side by side
            f2_t s[9];                                      // .x: over 2 dates, .y: over 3
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                s[c].x = q[c] + d1[c];
                s[c].y = s[c].x + d2[c];
            }
            const f2_t dp = det3_pk(s);
            const f2_t sq = dp * dp;
            f2_t pw;
            pw.x = sq.x;
            pw.y = sq.y * dp.y;
            const f2_t ta = ss.ca * pw, tb = ss.cb * pw;
            const T prod2 = det * det1, prod3 = det * prod12;
            const bool above = fminf(fminf(det, dp.x), dp.y) > dlo, below = fmaxf(fmaxf(det, dp.x), dp.y) < dhi;
            bad = bad | !(above & below);
            mask_push(m2F, prod2 < ta.x);
            mask_push(m2C, prod2 > tb.x);
            mask_push(m3F, prod3 < ta.y);
            mask_push(m3C, prod3 > tb.y);
            prod12 = prod2;
            det1 = det;
        } else {
            T s[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) s[c] = q[c] + d1[c];
            const T prod2 = det * det1;
            {
                const T dets = det3<T>(s);
                bad = bad | !((dets > dlo) & (dets < dhi) & (det > dlo) & (det < dhi));
                const T r = dets * dets;
                mask_push(m2F, prod2 < ca2 * r);
                mask_push(m2C, prod2 > cb2 * r);
            }
            {
#pragma unroll
                for (int c = 0; c < 9; ++c) s[c] = s[c] + d2[c];
                const T dets = det3<T>(s);
                bad = bad | !((dets > dlo) & (dets < dhi));
                const T r = (dets * dets) * dets;
                const T prod3 = det * prod12;
                mask_push(m3F, prod3 < ca3 * r);
                mask_push(m3C, prod3 > cb3 * r);
            }
            prod12 = prod2;
            det1 = det;
        }
    };
    // whole groups of NS dates first, nothing conditional inside a group; the slot of date t + 2 is
    // re-loaded with date t - PF as soon as date t has been worked on (a date in front of the series:
    // date 0 again, a cache hit)
    int tb = k - 1;
    for (; tb >= NS - 1; tb -= NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int t = tb - u;
            process(ring[u], ring[(u + NS - 1) % NS], ring[(u + NS - 2) % NS], t, ss.e[k - t]);
            load(t >= PF ? t - PF : 0, ring[(u + NS - 2) % NS]);
        }
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int t = tb - u;
        if (t >= 0) {
            process(ring[u], ring[(u + NS - 1) % NS], ring[(u + NS - 2) % NS], t, ss.e[k - t]);
            if (u < 2) load(t >= PF ? t - PF : 0, ring[(u + NS - 2) % NS]);
        }
    }
    // every product the reference forms is PP(l) / PP(l + m): a normal double while the exponents
    // PP passes through stay within 900 of each other (omnibus.hip)
    bad = bad || (emax - emin > 900);
    // tests that do not exist: the global test of the last date alone, marginal tests reaching behind
    // the series
    MT gI = mask_undecided(gF, gC), m2I = mask_undecided(m2F, m2C), m3I = mask_undecided(m3F, m3C);
    mask_keep_low(gF, k - 1);
    mask_keep_low(gI, k - 1);
    mask_keep_low(m2F, k - 2);
    mask_keep_low(m2I, k - 2);
    mask_keep_low(m3F, k >= 3 ? k - 3 : 0);
    mask_keep_low(m3I, k >= 3 ? k - 3 : 0);
    // nodata: the product of determinants is NaN or 0, P of the whole-series test NaN or 0 -- no
    // change anywhere, and no exact pass needed (see omnibus_c2_stream_kernel)
    if (dead) {
        bad = false;
        gF = mask_zero<MT>();
        gI = mask_zero<MT>();
    }

    const unsigned shard = (unsigned)(b % kC3Shards);
    const int64_t wpx0 = bpx0 + (tid & ~63);
    const int64_t wleft = g.nx - wpx0;
    const int wnp = wleft > 64 ? 64 : (wleft > 0 ? (int)wleft : 0);
    uint8_t *wob = g.change + (row * g.nx + wpx0) * (int64_t)k;

    const bool cand = in && (bad || mask_bit(gF, 0) || mask_bit(gI, 0));
    const bool dense = __popcll(__ballot(cand)) >= dense_min;
    bool listed = cand;                                       // a sparse wave lists its candidates
    MT mask = mask_zero<MT>();
    if (dense) {
        bool handoff = in && bad;
        bool done = !in || bad;
        int cur = 0;
        for (int l = 0; l < k - 1; ++l) {
            const bool act = !done && (cur == l);
            if (!__any(act)) continue;
            bool gi = mask_bit(gI, l), gf = mask_bit(gF, l);
            // A global test the suffix sums cannot decide (their rounding band grows with the length of
            // the segment: ~1 % of the pixels of a 48-date stack meet one) is decided from the reference's
            // OWN sums instead of handing the pixel over: the dates of ts[l:] once more, added forward in
            // `floating` -- bit-identical determinant, the tight band of the marginal tests.  (Round 4:
            // pass B behind the streaming search gathered and searched those pixels whole: 0.66 ms.)
            if (__any(act && gi)) {
                if (act && gi) {
                    T sg[9];
#pragma unroll
                    for (int c = 0; c < 9; ++c) sg[c] = (T)0;
                    int Lg = 0;
                    LmT Lmg = 0;
                    for (int t0 = l; t0 < k; t0 += 2) {
                        T qb[2][9];
#pragma unroll
                        for (int u = 0; u < 2; ++u) load(t0 + u < k ? t0 + u : k - 1, qb[u]);
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            if (t0 + u < k) {
#pragma unroll
                                for (int c = 0; c < 9; ++c) sg[c] = sg[c] + qb[u][c];
                                int e0;
                                float mf;
                                log2_parts(det3<T>(qb[u]), e0, mf);          // the pixel is not `bad`: det > 0
                                Lg += e0;
                                Lmg += (int)rintf(mf * kLogFix);
                            }
                        }
                    }
                    const int jj = k - l;
                    const T dets = det3<T>(sg);
                    const bool oks = (dets > (T)0) && (dets < (T)INFINITY);
                    const DenseScreenEntry c = scr_lds[jj];
                    const float x = dense_x<T>(dets, oks, Lg, Lmg, jj, c);
                    gf = oks && (x < c.a);
                    gi = !gf && !(oks && (x > c.b));
                }
            }
            const bool i2 = mask_bit(m2I, l), f2 = mask_bit(m2F, l);
            const bool i3 = mask_bit(m3I, l), f3 = mask_bit(m3F, l);
            int fire = -1;
            bool deep = false;
            if (act) {
                if (gi) {                                     // global test undecided
                    handoff = true;
                    done = true;
                } else if (!gf) {                             // nd/_change.pyx:241-242
                    done = true;
                } else if (l + 1 == k - 1) {
                    fire = l + 1;                             // the 2-date marginal IS the global test
                } else if (i2) {
                    handoff = true;
                    done = true;
                } else if (f2) {
                    fire = l + 1;
                } else if (l + 2 == k - 1) {
                    fire = l + 2;
                } else if (i3) {
                    handoff = true;
                    done = true;
                } else if (f3) {
                    fire = l + 2;
                } else {
                    deep = true;
                }
            }
            if (__any(deep)) {
                if (deep) {
                    // marginal tests over 4 and more dates: the dates of ts[l:] once more
                    T s[9];
#pragma unroll
                    for (int c = 0; c < 9; ++c) s[c] = (T)0;
                    int Ld = 0;
                    LmT Lmd = 0;
                    bool searching = true;
                    for (int t0 = l; t0 < k && searching; t0 += 2) {
                        T qb[2][9];
#pragma unroll
                        for (int u = 0; u < 2; ++u) load(t0 + u < k ? t0 + u : k - 1, qb[u]);
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int t = t0 + u;
                            if (searching && t < k) {
#pragma unroll
                                for (int c = 0; c < 9; ++c) s[c] = s[c] + qb[u][c];
                                const T det = det3<T>(qb[u]);
                                int e0;
                                float mf;
                                log2_parts(det, e0, mf);          // the pixel is not `bad`: det > 0
                                Ld += e0;
                                Lmd += (int)rintf(mf * kLogFix);
                                if (t >= l + 3) {                 // j = 2, 3 are decided: they do not fire
                                    if (t == k - 1) {
                                        fire = t;                 // the marginal over ts[l:] IS the global test
                                        searching = false;
                                    } else {
                                        const int jj = t - l + 1;
                                        const T dets = det3<T>(s);
                                        const bool oks = (dets > (T)0) && (dets < (T)INFINITY);
                                        const DenseScreenEntry c = scr_lds[jj];
                                        const float x = dense_x<T>(dets, oks, Ld, Lmd, jj, c);
                                        if (oks && (x < c.a)) {
                                            fire = t;
                                            searching = false;
                                        } else if (!(oks && (x > c.b))) {      // undecided
                                            handoff = true;
                                            done = true;
                                            searching = false;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
            if (fire >= 0) {
                mask_set(mask, fire, true);                   // nd/_change.pyx:252
                cur = fire;                                   // :255
                if (cur >= k - 1) done = true;                // :256
            }
        }
        if (handoff) mask = mask_zero<MT>();                  // pass B writes that pixel's changes
        if (change_rows_wave_ok(wob, k, wnp)) {
            store_change_rows_wave(wob, out_img + (tid >> 6) * (16 * kMaxDates), k, mask, lane);
        } else if (in) {
            uint8_t *res = wob + (int64_t)lane * k;
            if ((k & 3) == 0 && ((uintptr_t)res & 3) == 0) {
                uint32_t *w = reinterpret_cast<uint32_t *>(res);
                for (int q = 0; q < (k >> 2); ++q)
                    w[q] = (mask_nibble(mask, q) * 0x00204081u) & 0x01010101u;
            } else {
                for (int t = 0; t < k; ++t) res[t] = (uint8_t)(mask_bit(mask, t) ? 1 : 0);
            }
        }
        listed = handoff;
    }
    if (__any(listed)) {
        const unsigned long long lm_ = __ballot(listed);
        unsigned base = 0;
        if (lane == 0)
            base = atomicAdd(g.flag_count + shard * kC3CounterStride, (unsigned)__popcll(lm_));
        base = __shfl(base, 0);
        if (listed)
            g.flag_idx[(size_t)shard * g.seg + base + (unsigned)__popcll(lm_ & ((1ull << lane) - 1ull))] =
                (uint32_t)(row * g.nx + x0);
    }
    // a sparse wave zero-fills its own slice of the change map (np.zeros, nd/_change.pyx:275)
    if (!dense && wnp > 0) {
        const int nb = wnp * k;
        int head = (int)((16 - ((uintptr_t)wob & 15)) & 15);
        if (head > nb) head = nb;
        if (lane < head) wob[lane] = 0;
        const int nvec = (nb - head) >> 4;
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        u4 *vz = reinterpret_cast<u4 *>(wob + head);
        const u4 zero = {0u, 0u, 0u, 0u};
        for (int i = lane; i < nvec; i += 64) vz[i] = zero;
        const int tail0 = head + (nvec << 4);
        if (tail0 + lane < nb) wob[tail0 + lane] = 0;
    }
}

// ---- the chain search in two streaming passes (round 6) --------------------------------------------
// The full-pol counterpart of omnibus_c2_stream_chain_kernel (omnibus.hip), for the thresholds between the
// streaming search's and the sparse regime's (0.02 < alpha < 0.75): there a marginal test over four and more
// dates is the rule, and the streaming search above re-reads the dates of every such segment from memory
// (48 x 512 x 4096 at alpha = 0.5: 3.3 ms against 0.97 ms at 0.01).  single_pixel_change_detection consumes
// every date ONCE (nd/_change.pyx:246-256: the marginal tests of a segment extend one running sum, and the
// next segment starts at the date that fired); only the global test of ts[l:] looks ahead.  So:
//   pass 1, dates last to first: the global test of EVERY segment start from suffix sums in double, with
//           the rounding band of the reference's forward sums (omnibus_c3_stream_kernel's bound), two mask
//           bits per date;
//   pass 2, dates first to last (date index wave-uniform, every lane busy at every date): each lane carries
//           the reference's own running state of its CURRENT segment -- nine sums in `floating`, the double
//           product of the determinants -- and decides the marginal test over its j dates (bit-identical
//           determinants, the tight band; the constants of the lane's own j from an LDS table).  A global
//           test pass 1 could not decide is carried as ONE pending test per pixel and decided exactly at the
//           end from the reference's own sums of that segment; a second one hands the pixel to pass B.
// Twice the traffic of the one-pass form, the same cost at every threshold.
template <typename T, int MW>
__global__ void __launch_bounds__(kC3Threads) omnibus_c3_stream_chain_kernel(const C3Args<T> g, const OmniTab tab,
                                                                             const StreamScreen<c3_stream_nj(MW)> ss,
                                                                             const int dense_min)
{
    constexpr int PF = 3;
    typedef typename std::conditional<MW == 2, Bits128, unsigned long long>::type MT;
    constexpr int kMaxDates = MW == 2 ? 128 : 64;
    __shared__ StreamEntry tab_lds[kMaxDates + 1];
    __shared__ __align__(16) uint32_t out_img[(kC3Threads / 64) * 16 * kMaxDates];   // store_change_rows_wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t b = blockIdx.x;
    const int64_t row = b / g.blocks_per_row;
    const int64_t bx = b - row * g.blocks_per_row;
    const int64_t bpx0 = bx * (int64_t)kC3Threads;
    const int64_t x0 = bpx0 + tid;
    const int k = g.k;
    const bool in = x0 < g.nx;
    const int64_t xc = in ? x0 : g.nx - 1;                  // idle lanes re-read the last pixel
    const int64_t off0 = row * g.sy + xc * g.sx;
    auto load = [&](const int t, T (&v)[9]) {               // (g.off32: 32-bit byte offsets, see the streaming search)
        const unsigned o = (unsigned)(off0 + (int64_t)t * g.st) * (unsigned)sizeof(T);
#pragma unroll
        for (int c = 0; c < 9; ++c)
            v[c] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(g.pl[c]) + o);
    };
    for (int j = tid; j <= kMaxDates; j += kC3Threads) tab_lds[j] = ss.e[j];
    if (g.write_tab && b == 0)
        for (int j = tid; j <= k; j += kC3Threads) g.tab_dev[j] = tab.e[j];
    __syncthreads();

    // ---- pass 1: the global test of every segment start, dates last to first ----
    MT gF = mask_zero<MT>(), gC = mask_zero<MT>();
    bool bad = false, dead = false;
    const T dlo = (T)ss.dlo, dhi = (T)ss.dhi;
    {
        T ring[PF][9];
#pragma unroll
        for (int u = 0; u < PF; ++u) load(k - 1 - u > 0 ? k - 1 - u : 0, ring[u]);
        double S[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) S[c] = 0.0;
        double PP = 1.0;
        int emin = 1, emax = 1;
        auto back = [&](const T (&q)[9], const int t, const StreamEntry &e) {
            const T det = det3<T>(q);
            const T mn12 = (q[0] * q[1]) - ((q[3] * q[3]) + (q[4] * q[4]));
            const T mn13 = (q[0] * q[2]) - ((q[5] * q[5]) + (q[6] * q[6]));
            const T mn23 = (q[1] * q[2]) - ((q[7] * q[7]) + (q[8] * q[8]));
            const T dmin = fmin(fmin(q[0], q[1]), q[2]);
            const T mmin = fmin(fmin(mn12, mn13), mn23);
            // positive semi-definite dates, determinants inside (dlo, dhi): what the rounding bound of the
            // suffix sums and the range of the products in pass 2 rest on
            bad = bad | !((dmin > (T)0) & (mmin >= (T)0) & (det > dlo) & (det < dhi));
            dead = dead | !((det > (T)0) | (det < (T)0));
            PP = PP * (double)det;
#pragma unroll
            for (int c = 0; c < 9; ++c) S[c] += (double)q[c];
            const int jj = k - t;
            const double dets = det3<double>(S);
            const float df = (float)dets;
            bool okd;
            int es, eP;
            float ms, mP;
            if (sizeof(T) == 4) {
                okd = df > 7.888609052210118e-31f;
                log2_parts(df, es, ms);
            } else {
                okd = (dets > 0.0) & (dets < (double)INFINITY);
                log2_parts(dets, es, ms);
            }
            log2_parts(PP, eP, mP);
            emin = eP < emin ? eP : emin;
            emax = eP > emax ? eP : emax;
            const int E = (eP - e.re) - __mul24(jj, es);
            const float x = (float)E + __builtin_fmaf(-e.jf, ms, mP - e.rf);
            const float qq = (float)((S[0] * S[1]) * S[2]) * __builtin_amdgcn_rcpf(df);
            const float rel = e.cj * qq;                    // 1.46 (21 n + 20) u abc / D
            const float m2 = e.mj * rel;
            bad = bad | !(okd & (rel < 0.01f));
            mask_push(gF, x + m2 < e.a);
            mask_push(gC, x - m2 > e.b);
        };
        int tb = k - 1;
        for (; tb >= PF - 1; tb -= PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = tb - u;
                back(ring[u], t, ss.e[k - t]);
                load(t >= PF ? t - PF : 0, ring[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = tb - u;
            if (t >= 0) back(ring[u], t, ss.e[k - t]);
        }
        bad = bad | (emax - emin > 900);
    }
    MT gI = mask_undecided(gF, gC);
    mask_keep_low(gF, k - 1);
    mask_keep_low(gI, k - 1);
    if (dead) {                                  // a NaN or zero determinant: no change anywhere, no exact pass
        bad = false;
        gF = mask_zero<MT>();
        gI = mask_zero<MT>();
    }
    const unsigned shard = (unsigned)(b % kC3Shards);
    const int64_t wpx0 = bpx0 + (tid & ~63);
    const int64_t wleft = g.nx - wpx0;
    const int wnp = wleft > 64 ? 64 : (wleft > 0 ? (int)wleft : 0);
    uint8_t *wob = g.change + (row * g.nx + wpx0) * (int64_t)k;
    const bool cand = in && (bad || mask_bit(gF, 0) || mask_bit(gI, 0));
    const bool dense = __popcll(__ballot(cand)) >= dense_min;
    bool listed = cand;                          // a sparse wave lists its candidates
    if (dense) {
        // ---- pass 2: marginal tests and restarts, dates first to last ----
        bool handoff = in && bad;
        bool done = !in || bad || !(mask_bit(gF, 0) || mask_bit(gI, 0));
        bool pend = !done && mask_bit(gI, 0);
        int lp = 0;
        T ps[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) ps[c] = (T)0;
        double PQ = 1.0;
        MT mask = mask_zero<MT>();
        T ring[PF][9];
#pragma unroll
        for (int u = 0; u < PF; ++u) load(u < k ? u : k - 1, ring[u]);
        T s[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) s[c] = (T)0;
        double PP = 1.0;
        int j = 0;
        auto fwd = [&](const T (&q)[9], const int t) {
            const bool last = (t == k - 1);
            const T det = det3<T>(q);
#pragma unroll
            for (int c = 0; c < 9; ++c) s[c] = s[c] + q[c];
            PP = PP * (double)det;
            j = j + 1;
            const T dets = det3<T>(s);
            const bool oks = (dets > (T)0) & (dets < (T)INFINITY);
            const StreamEntry *ep = tab_lds + j;                 // the lane's own j
            const int re = ep->re;
            const float rf = ep->rf, ca = ep->a, cb = ep->b;
            int es, eP;
            float ms, mP;
            log2_parts(dets, es, ms);
            log2_parts(PP, eP, mP);
            const int E = (eP - re) - __mul24(j, es);
            const float x = (float)E + __builtin_fmaf(-(float)j, ms, mP - rf);
            const bool tested = t > 0;                           // (the first date only starts the state)
            const bool fires = tested & (last | (oks & (x < ca)));
            const bool cant = !tested | (!last & oks & (x > cb));
            const bool act = !done;
            const bool und = act & !(fires | cant);
            const bool f = act & fires;
            handoff = handoff | und;
            mask_set(mask, t, f);                                // nd/_change.pyx:252
            const bool gi = mask_bit(gI, t), gf = mask_bit(gF, t);
            const bool newp = f & !last & gi;                    // an undecided global test starts here
            handoff = handoff | (newp & pend);
            done = done | und | (newp & pend) | (f & (last | !(gf | gi)));   // :256, :241-242
            PQ = PQ * (double)det;
            const bool startp = newp & !pend;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                ps[c] = ps[c] + q[c];
                ps[c] = startp ? q[c] : ps[c];
                s[c] = f ? q[c] : s[c];
            }
            PQ = startp ? (double)det : PQ;
            lp = startp ? t : lp;
            pend = pend | startp;
            PP = f ? (double)det : PP;
            j = f ? 1 : j;
        };
        int tb = 0;
        for (; tb + PF <= k; tb += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = tb + u;
                fwd(ring[u], t);
                load(t + PF < k ? t + PF : k - 1, ring[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = tb + u;
            if (t < k) fwd(ring[u], t);
        }
        if (__any(pend)) {
            // the pending global test over ts[lp:], from the reference's own sums
            const int jp = k - lp;
            const T dets = det3<T>(ps);
            const bool oks = (dets > (T)0) & (dets < (T)INFINITY);
            const StreamEntry *ep = tab_lds + jp;
            int es, eP;
            float ms, mP;
            log2_parts(dets, es, ms);
            log2_parts(PQ, eP, mP);
            const int E = (eP - ep->re) - __mul24(jp, es);
            const float x = (float)E + __builtin_fmaf(-(float)jp, ms, mP - ep->rf);
            const bool fires = oks & (x < ep->a);
            const bool cant = oks & (x > ep->b);
            if (pend) {
                if (!(fires | cant)) handoff = true;
                if (cant) mask_keep_low(mask, lp + 1);       // the search ended at lp (:241-242)
            }
        }
        if (handoff) mask = mask_zero<MT>();                  // pass B writes that pixel's changes
        if (change_rows_wave_ok(wob, k, wnp)) {
            store_change_rows_wave(wob, out_img + (tid >> 6) * (16 * kMaxDates), k, mask, lane);
        } else if (in) {
            uint8_t *res = wob + (int64_t)lane * k;
            for (int t = 0; t < k; ++t) res[t] = (uint8_t)(mask_bit(mask, t) ? 1 : 0);
        }
        listed = handoff;
    }
    if (__any(listed)) {
        const unsigned long long lm_ = __ballot(listed);
        unsigned base = 0;
        if (lane == 0)
            base = atomicAdd(g.flag_count + shard * kC3CounterStride, (unsigned)__popcll(lm_));
        base = __shfl(base, 0);
        if (listed)
            g.flag_idx[(size_t)shard * g.seg + base + (unsigned)__popcll(lm_ & ((1ull << lane) - 1ull))] =
                (uint32_t)(row * g.nx + x0);
    }
    // a sparse wave zero-fills its own slice of the change map (np.zeros, nd/_change.pyx:275)
    if (!dense && wnp > 0) {
        const int nb = wnp * k;
        int head = (int)((16 - ((uintptr_t)wob & 15)) & 15);
        if (head > nb) head = nb;
        if (lane < head) wob[lane] = 0;
        const int nvec = (nb - head) >> 4;
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        u4 *vz = reinterpret_cast<u4 *>(wob + head);
        const u4 zero = {0u, 0u, 0u, 0u};
        for (int i = lane; i < nvec; i += 64) vz[i] = zero;
        const int tail0 = head + (nvec << 4);
        if (tail0 + lane < nb) wob[tail0 + lane] = 0;
    }
}

// ---- pass B ---------------------------------------------------------------------------------
// LANES: listed pixels per wave.  The LDS image of 64 series of 48 dates is 110 KB: one wave per
// CU, whose gather (1.13 ms of config 4's share, sector-traffic bound) and search (0.42 ms, latency
// bound) then take turns.  With 32 pixels per wave the image is 55 KB and two waves share a CU: one
// gathers while the other searches.  (The upper 32 lanes of such a wave only idle along.)
template <typename T, bool USE_LDS, int LANES = 64>
__global__ void __launch_bounds__(64) omnibus_c3_search_kernel(const C3Args<T> s)
{
    extern __shared__ __align__(16) unsigned char nd_smem3[];
    T *lds = reinterpret_cast<T *>(nd_smem3);
    const int lane = threadIdx.x;
    const int k = s.k;
    const unsigned shard = blockIdx.x % kC3Shards;
    const unsigned lblock = blockIdx.x / kC3Shards, nlblock = gridDim.x / kC3Shards;
    const uint32_t n = s.flag_count[shard * kC3CounterStride];
    const uint32_t *list = s.flag_idx + (size_t)shard * s.seg;
    // (entries whose series pass A dumped from its registers are omnibus_c3_search_dump_kernel's)
    const unsigned first = s.dump != nullptr ? s.dump_cap : 0u;
    if (first >= n || lblock * (unsigned)LANES >= n - first) return;   // nothing for this block
    if (n <= s.starts_max) return;                        // a short list: omnibus_c3_search_starts_kernel's
    // per-j constants of the screen as four LDS arrays behind the series image (omnibus.hip): every
    // lane looks up its own j in every iteration
    const int kp = k + 1;
    double *scr = reinterpret_cast<double *>(nd_smem3 + (USE_LDS ? (size_t)k * 9 * LANES * sizeof(T) : 0));
    for (int j = lane; j <= k; j += 64) {
        const OmniTabEntry e = s.tab_dev[j];
        scr[j] = e.m2rho;
        scr[kp + j] = e.pklogk;
        scr[2 * kp + j] = e.zlo_a;
        scr[3 * kp + j] = e.zhi_a;
    }
    __syncthreads();

    for (uint32_t base = first + lblock * (unsigned)LANES; base < n; base += nlblock * (unsigned)LANES) {
        const bool mylane = lane < LANES;                   // (LANES = 32: the upper half idles)
        const uint32_t idx = base + lane;
        const bool active = mylane && idx < n;
        const int64_t pix = active ? (int64_t)list[idx] : 0;
        const int64_t row = pix / s.nx_orig, col = pix - row * s.nx_orig;
        const int64_t off = row * s.sy + col * s.sx;
        if (USE_LDS && mylane && s.pm_vec) {
            // Pixel-major inputs: a listed pixel's series is a contiguous run per plane (or per interleaved pair).
            // 16-byte loads, the whole series of a lane in flight group by group: 108 (real planes) or 72 + 36
            // loads of 16 bytes instead of 432 of 4 -- the sectors are the same, the requests a quarter.
            constexpr int VE = 16 / (int)sizeof(T);
            typedef C3Pack<T, VE> PV;
            const bool joint = s.mult[3] == 2;
            for (int t0 = 0; t0 < k; t0 += 2 * VE) {
                PV q[9][2];
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (t0 + h * VE < k) {
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            q[c][h] = *reinterpret_cast<const PV *>(s.pl[c] + off + t0 + h * VE);
                        if (joint) {
                            // VE dates of (re, im): two vectors of the interleaved array
#pragma unroll
                            for (int c = 3; c < 9; c += 2) {
                                const T *p = s.pl[c] + 2 * (off + t0 + h * VE);
                                q[c][h] = *reinterpret_cast<const PV *>(p);
                                q[c + 1][h] = *reinterpret_cast<const PV *>(p + VE);
                            }
                        } else {
#pragma unroll
                            for (int c = 3; c < 9; ++c)
                                q[c][h] = *reinterpret_cast<const PV *>(s.pl[c] + off + t0 + h * VE);
                        }
                    }
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (t0 + h * VE < k) {
#pragma unroll
                        for (int j = 0; j < VE; ++j) {
                            const int t = t0 + h * VE + j;
#pragma unroll
                            for (int c = 0; c < 3; ++c) lds[(t * 9 + c) * LANES + lane] = q[c][h].v[j];
                            if (joint) {
#pragma unroll
                                for (int c = 3; c < 9; c += 2) {
                                    const PV &w0 = j < VE / 2 ? q[c][h] : q[c + 1][h];
                                    lds[(t * 9 + c) * LANES + lane] = w0.v[2 * (j % (VE / 2))];
                                    lds[(t * 9 + c + 1) * LANES + lane] = w0.v[2 * (j % (VE / 2)) + 1];
                                }
                            } else {
#pragma unroll
                                for (int c = 3; c < 9; ++c) lds[(t * 9 + c) * LANES + lane] = q[c][h].v[j];
                            }
                        }
                    }
            }
        } else if (USE_LDS && mylane) {
            // eight dates (72 independent loads per lane) in flight at a time
            for (int t0 = 0; t0 < k; t0 += 8) {
                T q[8][9];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (t0 + u < k) {
                        const int64_t o = off + (int64_t)(t0 + u) * s.st;
#pragma unroll
                        for (int c = 0; c < 9; ++c) q[u][c] = s.pl[c][o * s.mult[c]];
                    }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (t0 + u < k) {
#pragma unroll
                        for (int c = 0; c < 9; ++c) lds[((t0 + u) * 9 + c) * LANES + lane] = q[u][c];
                    }
            }
        }
        // from memory (no LDS image): date t of this lane's series, with the date most likely to be
        // asked for next already in flight.  A lane's nine values of a date are nine isolated 4-byte
        // reads; what hides their latency is occupancy (no 110 KB image: 8+ waves per CU instead of
        // one) plus this one date of read-ahead.
        T nxt[9];
        int nxt_t = -1;
        auto fetch = [&](int t, T (&v)[9]) {
            const int64_t o = off + (int64_t)t * s.st;
#pragma unroll
            for (int c = 0; c < 9; ++c) v[c] = s.pl[c][o * s.mult[c]];
        };
        auto load_step = [&](Accum3<T> &A, int t) {
            T v[9];
            if (USE_LDS) {
#pragma unroll
                for (int c = 0; c < 9; ++c) v[c] = lds[(t * 9 + c) * LANES + lane];
            } else {
                if (nxt_t == t) {
#pragma unroll
                    for (int c = 0; c < 9; ++c) v[c] = nxt[c];
                } else {
                    fetch(t, v);
                }
                const int tn = t + 1 < k ? t + 1 : t;
                fetch(tn, nxt);
                nxt_t = tn;
            }
            A.step(v);
        };

        // one sweep per segment, as in omnibus_c2_search_kernel (nd/_change.pyx:235-257)
        Accum3<T> A;
        A.reset();
        int l = 0, t = 0, fire_at = -1;
        bool done = !active;
        uint8_t *res = s.change + pix * (int64_t)k;
        while (__any(!done)) {
            if (!done) {
                load_step(A, t);
                const int jj = t - l + 1;
                const bool last = (t == k - 1);
                const bool need = (jj >= 2) && (fire_at < 0 || last);
                // screen on the hardware-log2 statistic; the exact evaluation sits behind a
                // wave-uniform branch so that it is not folded into the loop body (omnibus.hip)
                bool fires = false, inband = false;
                if (need) {
                    const double za = z_approx3<T>(A, jj, s.nlooks, scr[jj], scr[kp + jj]);
                    fires = (za > scr[3 * kp + jj]) && (za < INFINITY);
                    inband = (za >= scr[2 * kp + jj]) && !fires;
                }
                if (__any(inband)) {
                    if (inband) {
                        const OmniTabEntry e = s.tab_dev[jj];
                        const T zp = z_stat3<T>(A, jj, s.nlooks, e);
                        const double zd = (double)zp;
                        int verdict = !(zd >= e.zlo) ? 0 : ((zd > e.zhi && zd < INFINITY) ? 1 : 2);
                        if (verdict == 2) {
                            double zv[1] = {zd}, P1[1], P2[1];
                            chisq_pair<1>(zv, 9 * (jj - 1), e.lgam, P1, P2);
                            const T P = combine_P<T>(P1[0], P2[0], e.omega2);
                            verdict = ((double)P > s.alpha) ? 1 : 0;
                        }
                        fires = (verdict == 1);
                    }
                }
                if (fires && fire_at < 0) fire_at = t;
                if (!last) {
                    t = t + 1;
                } else if (fires && jj >= 2) {
                    res[fire_at] = 1;
                    l = fire_at;
                    if (l >= k - 1) {
                        done = true;
                    } else {
                        A.reset();
                        t = l;
                        fire_at = -1;
                    }
                } else {
                    done = true;
                }
            }
        }
    }
}


// ---- pass B on the dump of the time-split pass A (round 6) ------------------------------------------
// A listed pixel's series lies in the dump as one contiguous run [t][9] (omnibus_c3_retain_kernel), so the
// search needs neither the gather nor the LDS image that held it: ONE LANE PER PIXEL ON ALL 64 LANES, the
// series read where it lies in chunks of four dates (nine 16-byte loads per lane: 27 sectors per pixel and
// sweep instead of 432 isolated values, and a second sweep of the same pixel finds them in the L2), no LDS but
// the per-j constants, so that the occupancy is what the registers allow.  (The image form ran 16 pixels
// per wave to fit five waves on a CU: three quarters of every vector instruction idle, 0.84 ms on config
// 4's share with the dump as its source.)  The search itself is omnibus_c3_search_kernel's, step for step:
// one sweep per segment (nd/_change.pyx:235-257), the hardware-log2 screen, the exact evaluation behind a
// wave-uniform branch.
// MODE 0: the dump.  MODE 1 / 2 (round 6 as well): pixel-major inputs (nd_amd_omnibus_c3_pixel_major), where
// a listed pixel's series is a contiguous run per variable in the caller's arrays -- nine real arrays (1) or
// three real and three interleaved complex ones (2): the same chunks of four dates, one or two 16-byte
// pieces per variable, no LDS image (the image form: 0.79 ms on config 4's share).
// (vector, element) of value c of date tt of a chunk, per mode
template <int MODE, int VE, int CD>
__device__ __forceinline__ constexpr int c3_chunk_vec(const int tt, const int c)
{
    return MODE == 0 ? (tt * 9 + c) / VE
           : (MODE == 1 || c < 3) ? c * (CD / VE) + tt / VE
                                  : 3 * (CD / VE) + ((c - 3) / 2) * (2 * CD / VE) + (2 * tt + ((c - 3) & 1)) / VE;
}
template <int MODE, int VE, int CD>
__device__ __forceinline__ constexpr int c3_chunk_elem(const int tt, const int c)
{
    return MODE == 0 ? (tt * 9 + c) % VE : (MODE == 1 || c < 3) ? tt % VE : (2 * tt + ((c - 3) & 1)) % VE;
}

template <typename T, int MODE>
__global__ void __launch_bounds__(64) omnibus_c3_search_dump_kernel(const C3Args<T> s)
{
    constexpr int VE = 16 / (int)sizeof(T);              // values per 16-byte piece
    constexpr int CD = 4;                                // dates per chunk
    constexpr int NV = CD * 9 / VE;                      // pieces per chunk
    typedef C3Pack<T, VE> PV;
    extern __shared__ __align__(16) unsigned char nd_smem3d[];
    double *scr = reinterpret_cast<double *>(nd_smem3d);
    const int lane = threadIdx.x;
    const int k = s.k;
    const unsigned shard = blockIdx.x % kC3Shards;
    const unsigned lblock = blockIdx.x / kC3Shards, nlblock = gridDim.x / kC3Shards;
    const uint32_t nall = s.flag_count[shard * kC3CounterStride];
    const uint32_t n = (MODE != 0 || nall < s.dump_cap) ? nall : s.dump_cap;
    const uint32_t *list = s.flag_idx + (size_t)shard * s.seg;
    if (lblock * 64u >= n) return;
    const int kp = k + 1;
    for (int j = lane; j <= k; j += 64) {
        const OmniTabEntry e = s.tab_dev[j];
        scr[j] = e.m2rho;
        scr[kp + j] = e.pklogk;
        scr[2 * kp + j] = e.zlo_a;
        scr[3 * kp + j] = e.zhi_a;
    }
    __syncthreads();

    for (uint32_t base = lblock * 64u; base < n; base += nlblock * 64u) {
        const uint32_t idx = base + lane;
        const bool active = idx < n;
        const int64_t pix = active ? (int64_t)list[idx] : 0;
        const PV *src = nullptr;
        if (MODE == 0)
            src = reinterpret_cast<const PV *>(s.dump + ((size_t)shard * s.dump_cap + (active ? idx : base)) *
                                                            s.dump_stride);
        const int64_t off = pix * (int64_t)k;              // (MODE 1 / 2: the pixel's run in every variable)
        Accum3<T> A;
        A.reset();
        int l = 0, t = 0, fire_at = -1;
        bool done = !active;
        uint8_t *res = s.change + pix * (int64_t)k;
        // chunk `ci` of this lane's series into q (MODE 1 / 2: whole 16-byte pieces of dates, a piece behind
        // the series is not read)
        auto load_chunk = [&](const int ci, PV (&q)[NV]) {
            if (MODE == 0) {
#pragma unroll
                for (int u = 0; u < NV; ++u) q[u] = src[ci * NV + u];
            } else {
                constexpr int PR = CD / VE;                 // pieces of a real variable per chunk
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const bool real = MODE == 1 || u < 3 * PR;
                    const int c = real ? u / PR : 3 + 2 * ((u - 3 * PR) / (2 * PR));
                    const int wv = real ? u % PR : (u - 3 * PR) % (2 * PR);
                    const int t_first = ci * CD + (real ? wv * VE : (wv * VE) / 2);
                    if (t_first < k)
                        q[u] = real ? reinterpret_cast<const PV *>(s.pl[c] + off + ci * CD)[wv]
                                    : reinterpret_cast<const PV *>(s.pl[c] + 2 * (off + ci * CD))[wv];
                }
            }
        };
        // (reading the next chunk ahead into a second set of registers was built and measured slower:
        //  0.335 against 0.235 ms on config 4's share)
        while (__any(!done)) {
            const int ci = t / CD;                          // this lane's chunk (lanes restart at different dates)
            PV q[NV];
            if (!done) load_chunk(ci, q);
#pragma unroll 1
            for (int tt = 0; tt < CD; ++tt) {
                const bool on = !done && (t == ci * CD + tt);
                if (!__any(on)) continue;
                T v[9];
                // (tt is wave-uniform: a scalar switch over static positions, no indexed register access)
                switch (tt) {
#define ND_C3_PICK(TT)                                                                 \
    case TT:                                                                           \
        _Pragma("unroll") for (int c = 0; c < 9; ++c)                                 \
            v[c] = q[c3_chunk_vec<MODE, VE, CD>(TT, c)].v[c3_chunk_elem<MODE, VE, CD>(TT, c)]; \
        break;
                    ND_C3_PICK(0)
                    ND_C3_PICK(1)
                    ND_C3_PICK(2)
                default:
                    ND_C3_PICK(3)
#undef ND_C3_PICK
                }
                if (on) {
                    A.step(v);
                    const int jj = t - l + 1;
                    const bool last = (t == k - 1);
                    const bool need = (jj >= 2) && (fire_at < 0 || last);
                    bool fires = false, inband = false;
                    if (need) {
                        const double za = z_approx3<T>(A, jj, s.nlooks, scr[jj], scr[kp + jj]);
                        fires = (za > scr[3 * kp + jj]) && (za < INFINITY);
                        inband = (za >= scr[2 * kp + jj]) && !fires;
                    }
                    if (__any(inband)) {
                        if (inband) {
                            const OmniTabEntry e = s.tab_dev[jj];
                            const T zp = z_stat3<T>(A, jj, s.nlooks, e);
                            const double zd = (double)zp;
                            int verdict = !(zd >= e.zlo) ? 0 : ((zd > e.zhi && zd < INFINITY) ? 1 : 2);
                            if (verdict == 2) {
                                double zv[1] = {zd}, P1[1], P2[1];
                                chisq_pair<1>(zv, 9 * (jj - 1), e.lgam, P1, P2);
                                const T P = combine_P<T>(P1[0], P2[0], e.omega2);
                                verdict = ((double)P > s.alpha) ? 1 : 0;
                            }
                            fires = (verdict == 1);
                        }
                    }
                    if (fires && fire_at < 0) fire_at = t;
                    if (!last) {
                        t = t + 1;
                    } else if (fires && jj >= 2) {
                        res[fire_at] = 1;
                        l = fire_at;
                        if (l >= k - 1) {
                            done = true;
                        } else {
                            A.reset();
                            t = l;                          // (an earlier date: met again behind the next chunk load)
                            fire_at = -1;
                        }
                    } else {
                        done = true;
                    }
                }
            }
        }
    }
}

// ---- pass B behind the time-split pass A: lockstep rounds on the blocked dump (round 6) ---------------
// One lane per listed pixel walking its own dates (omnibus_c3_search_dump_kernel<T, 0>, the first form of this
// round: 0.235 ms on config 4's share) issues, at every step, one instruction whose 64 lanes address 64
// different lines; the rate at which a CU looks those up bounds such a sweep (the dual-pol twin measured it:
// DESIGN-EXPERIMENTS.md).  Here the 64 listed pixels of a wave are a BLOCK of the dump -- their series
// interleaved by groups of four dates and by component -- and the date index is wave-uniform: a round walks the
// dates from the earliest segment start among the wave's unfinished pixels to the end, nine contiguous kilobytes
// per group of four dates (the next group in flight), every lane folding from its own segment start on; behind
// the last date a lane commits the first firing date of its segment if the global test fired
// (nd/_change.pyx:235-257, one sweep per segment) and starts its next segment there; rounds repeat while a lane
// has a segment left.  Fold, screen and exact evaluation are omnibus_c3_search_kernel's, operation for operation.
// The screen of a test is the float32 one of the dual-pol sweeps (x = log2 prod det - j log2 det(sum) relative to the
// decision point, from exponents and the hardware log2 of the mantissas of the reference's own running values, against
// the per-j band of make_dense_entry -- p-agnostic): whatever it cannot decide takes the exact evaluation.
template <typename T>
__global__ void __launch_bounds__(64) omnibus_c3_search_rounds_kernel(const C3Args<T> s, const DenseScreen fscr)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __align__(16) unsigned char nd_smem3r[];
    DenseScreenEntry *scr_f = reinterpret_cast<DenseScreenEntry *>(nd_smem3r);
    const int lane = threadIdx.x;
    const int k = s.k;
    const unsigned shard = blockIdx.x % kC3Shards;
    const unsigned lblock = blockIdx.x / kC3Shards, nlblock = gridDim.x / kC3Shards;
    const uint32_t nall = s.flag_count[shard * kC3CounterStride];
    const uint32_t n = nall < s.dump_cap ? nall : s.dump_cap;            // (dump_cap: a multiple of 64)
    const uint32_t *list = s.flag_idx + (size_t)shard * s.seg;
    if (lblock * 64u >= n) return;
    for (int j = lane; j <= k; j += 64) scr_f[j] = fscr.e[j];
    __syncthreads();
    const unsigned G = s.dump_stride / 36u;

    for (uint32_t base = lblock * 64u; base < n; base += nlblock * 64u) {
        const uint32_t idx = base + lane;
        const bool active = idx < n;
        const int64_t pix = active ? (int64_t)list[idx] : 0;
        // group g, component c of this lane's series: blk[(g * 9 + c) * 64]
        const f4 *blk = reinterpret_cast<const f4 *>(s.dump) + ((size_t)shard * (s.dump_cap >> 6) + (base >> 6)) * G * 9 * 64 + lane;
        uint8_t *res = s.change + pix * (int64_t)k;
        Accum3<T> A;
        A.reset();
        int l = 0, fire_at = -1;
        bool done = !active;
        while (__any(!done)) {
            int t0 = done ? k : l;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const int o = __shfl_xor(t0, off);
                t0 = o < t0 ? o : t0;
            }
            t0 = __builtin_amdgcn_readfirstlane(t0);
            const int g0 = t0 >> 2, gn = (k + 3) >> 2;
            bool fires = false;                                  // of the test met at the last date: the global test
            f4 cur[9], nxt[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) nxt[c] = blk[((size_t)g0 * 9 + c) * 64];
            for (int gq = g0; gq < gn; ++gq) {
#pragma unroll
                for (int c = 0; c < 9; ++c) cur[c] = nxt[c];
                const int gq1 = gq + 1 < gn ? gq + 1 : gq;
#pragma unroll
                for (int c = 0; c < 9; ++c) nxt[c] = blk[((size_t)gq1 * 9 + c) * 64];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = 4 * gq + tt;                   // wave-uniform
                    if (t < k && t >= t0) {
                        T v[9];
#pragma unroll
                        for (int c = 0; c < 9; ++c) v[c] = cur[c][tt];
                        const bool on = !done && t >= l;
                        if (on) A.step(v);
                        const int jj = t - l + 1;
                        const bool last = (t == k - 1);
                        const bool need = on && (jj >= 2) && (fire_at < 0 || last);
                        bool f = false, inband = false;
                        if (need) {
                            const T dets = det3<T>(A.s);
                            const bool ok = (dets > (T)0) & (dets < (T)INFINITY) & __builtin_amdgcn_class(A.prod, 0x100);
                            const DenseScreenEntry c = scr_f[jj];
                            int es, eP;
                            float ms, mP;
                            log2_parts(ok ? dets : (T)1, es, ms);
                            log2_parts(ok ? A.prod : 1.0, eP, mP);
                            const int E = (eP - c.re) - __mul24(jj, es);
                            const float x = (float)E + __builtin_fmaf(-(float)jj, ms, mP - c.rf);
                            f = ok & (x < c.a);
                            inband = !(f | (ok & (x > c.b)));
                        }
                        if (__any(inband)) {
                            if (inband) {
                                const OmniTabEntry e = s.tab_dev[jj];
                                const T zp = z_stat3<T>(A, jj, s.nlooks, e);
                                const double zd = (double)zp;
                                int verdict = !(zd >= e.zlo) ? 0 : ((zd > e.zhi && zd < INFINITY) ? 1 : 2);
                                if (verdict == 2) {
                                    double zv[1] = {zd}, P1[1], P2[1];
                                    chisq_pair<1>(zv, 9 * (jj - 1), e.lgam, P1, P2);
                                    const T P = combine_P<T>(P1[0], P2[0], e.omega2);
                                    verdict = ((double)P > s.alpha) ? 1 : 0;
                                }
                                f = (verdict == 1);
                            }
                        }
                        if (need) {
                            if (f && fire_at < 0) fire_at = t;
                            if (last) fires = f;
                        }
                    }
                }
            }
            if (!done) {
                if (fires && (k - l) >= 2) {
                    res[fire_at] = 1;                          // :252
                    l = fire_at;                               // :255
                    if (l >= k - 1) {
                        done = true;                           // :256
                    } else {
                        A.reset();
                        fire_at = -1;
                    }
                } else {
                    done = true;                               // :241-242
                }
            }
        }
    }
}

// ---- pass B, one lane per SEGMENT START (short lists behind the streaming search) ------------------
// The full-pol counterpart of omnibus_c2_search_starts_kernel (omnibus.hip).  What the streaming search
// hands over are a few thousand pixels of 8 M (2 070 at alpha = 0.01 on config 4's share), nearly every
// date of them a change: the gather-and-sweep form above walks their 47 segments one after the other,
// each to the end of the series -- 1 100 dependent date steps, 0.64 ms for any list, the time of ONE
// wave.  Here lane l of a pixel's group sweeps ts[l:] once, on its own: nxt(l) = the date of the first
// firing marginal test if the global test over ts[l:] fires, "stop" otherwise -- the same evaluations,
// in the same order and arithmetic, as the sweep started at l.  The pixel's changes are the chain
// 0 -> nxt(0) -> nxt(nxt(0)) ... (nd/_change.pyx:235-257), followed through wave shuffles.
template <typename T>
__global__ void __launch_bounds__(64) omnibus_c3_search_starts_kernel(const C3Args<T> s)
{
    // the series of the wave's pixels (gathered once, every load of a pixel in flight together: the
    // sweeps then run without a memory access) and the per-j constants of the screen
    __shared__ T ser[9 * (kTabArgs + 2) + 9 * 64];
    __shared__ double scr[4 * (kTabArgs + 1)];
    const int lane = threadIdx.x;
    const int k = s.k;
    const int ns = k - 1;                       // segment starts per pixel: 0 .. k - 2
    const int ppw = ns <= 64 ? 64 / ns : 1;     // pixels per wave
    const int nsw = ns <= 64 ? ns : 64;         // lanes per pixel
    const int nsweep = ns <= 64 ? 1 : 2;        // 65 .. 128 starts: lane l takes the starts l and l + 64
    const int grp = lane / nsw;
    const int l0 = lane - grp * nsw;
    const unsigned shard = blockIdx.x % kC3Shards;
    const unsigned lblock = blockIdx.x / kC3Shards, nlblock = gridDim.x / kC3Shards;
    const uint32_t n = s.flag_count[shard * kC3CounterStride];
    if (n > s.starts_max) return;               // a long list: the gather-and-sweep form's
    if (lblock * (uint32_t)ppw >= n) return;
    const uint32_t *list = s.flag_idx + (size_t)shard * s.seg;
    const int kp = k + 1;
    for (int j = lane; j <= k; j += 64) {
        const OmniTabEntry e = s.tab_dev[j];
        scr[j] = e.m2rho;
        scr[kp + j] = e.pklogk;
        scr[2 * kp + j] = e.zlo_a;
        scr[3 * kp + j] = e.zhi_a;
    }
    for (uint32_t base = lblock * (uint32_t)ppw; base < n; base += nlblock * (uint32_t)ppw) {
        const uint32_t idx = base + (uint32_t)grp;
        const bool active = (grp < ppw) && (idx < n);
        const int64_t pix = active ? (int64_t)list[idx] : 0;
        const int64_t row = pix / s.nx_orig, col = pix - row * s.nx_orig;
        const int64_t off = row * s.sy + col * s.sx;
        T *mine = ser + grp * (k * 9);
        __syncthreads();                        // (one wave per block: orders the LDS accesses of two rounds)
        if (active)
            for (int e = l0; e < k * 9; e += nsw) mine[e] = s.pl[e % 9][(off + (int64_t)(e / 9) * s.st) * s.mult[e % 9]];
        __syncthreads();
        int nxt_a = -1, nxt_b = -1;             // per sweep: the next segment start, or stop
        for (int sweep = 0; sweep < nsweep; ++sweep) {
            const int l = l0 + 64 * sweep;
            const bool lact = active && l < ns;
            Accum3<T> A;
            A.reset();
            int fire_at = -1;
            int nxt = -1;                       // stop
            for (int i = 0; i < k; ++i) {       // the lane's dates are l, l + 1, ...
                const int t = l + i;
                if (!__any(lact && t < k)) break;
                const bool on = lact && t < k;
                T q[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) q[c] = on ? mine[t * 9 + c] : ((c < 3) ? (T)1 : (T)0);
                if (on) A.step(q);
                const int jj = i + 1;
                const bool last = (t == k - 1);
                const bool need = on && (jj >= 2) && (fire_at < 0 || last);
                bool fires = false, inband = false;
                if (need) {
                    const double za = z_approx3<T>(A, jj, s.nlooks, scr[jj], scr[kp + jj]);
                    fires = (za > scr[3 * kp + jj]) && (za < INFINITY);
                    inband = (za >= scr[2 * kp + jj]) && !fires;
                }
                if (__any(inband)) {
                    if (inband) {
                        const OmniTabEntry e = s.tab_dev[jj];
                        const T zp = z_stat3<T>(A, jj, s.nlooks, e);
                        const double zd = (double)zp;
                        int verdict = !(zd >= e.zlo) ? 0 : ((zd > e.zhi && zd < INFINITY) ? 1 : 2);
                        if (verdict == 2) {
                            double zv[1] = {zd}, P1[1], P2[1];
                            chisq_pair<1>(zv, 9 * (jj - 1), e.lgam, P1, P2);
                            const T P = combine_P<T>(P1[0], P2[0], e.omega2);
                            verdict = ((double)P > s.alpha) ? 1 : 0;
                        }
                        fires = (verdict == 1);
                    }
                }
                if (on && fires && fire_at < 0) fire_at = t;
                if (on && last && fires) nxt = fire_at;      // the global test of ts[l:] fired (jj >= 2 here)
            }
            if (sweep == 0) nxt_a = nxt; else nxt_b = nxt;
        }
        // follow the chain from l = 0; the group's first lane writes the changes
        int at = 0;
        for (int step = 0; step < ns; ++step) {
            const int ats = (at >= 0 && at < ns) ? at : 0;
            const int src = grp * nsw + (ats & 63) % nsw;
            const int na = __shfl(nxt_a, src), nb = __shfl(nxt_b, src);
            const int n1 = (ns > 64 && ats >= 64) ? nb : na;
            if (at >= 0 && at < ns) {
                if (n1 < 0) {
                    at = -1;                                   // :241-242
                } else {
                    if (active && l0 == 0) s.change[pix * (int64_t)k + n1] = 1;  // :252 (the row is all zeros)
                    at = n1;                                   // :255; n1 = k - 1 ends the search (:256)
                }
            }
        }
    }
}

// ---- host -----------------------------------------------------------------------------------
struct C3Workspace {
    size_t off_count, off_tab, off_idx, off_dump, total;
    uint32_t seg;
    uint32_t dump_cap, dump_stride;      // slots per shard, elements per slot (0: no dump at this length)
    int kq;                              // dates per slice of the time-split pass A
};

// Dump of the time-split pass A (float32, up to 64 dates): room for one pixel in 16 per shard -- three times
// the candidates of the benchmark's threshold; what does not fit is gathered from the planes as before.
constexpr int kC3RetainMaxK = 64;
constexpr int64_t kC3DumpShare = 16;

static C3Workspace c3_layout(int64_t npix, int64_t ny, int64_t k)
{
    C3Workspace w;
    // (lists: blocks of 64 pixels per row at the finest, i.e. at most ceil(npix / 64) + ny blocks of <= 64 entries)
    const int64_t nb256 = ceil_div(npix, kC3Threads) + ny;
    w.seg = (uint32_t)((ceil_div(nb256, kC3Shards) + 1) * kC3Threads);
    w.off_count = 0;
    w.off_tab = align256(kC3CounterBytes);
    w.off_idx = w.off_tab + align256((size_t)(k + 1) * sizeof(OmniTabEntry));
    w.off_dump = w.off_idx + align256((size_t)w.seg * kC3Shards * sizeof(uint32_t));
    w.kq = (int)(ceil_div(ceil_div(k, kC3Slices), 4) * 4);
    if (k >= 2 && k <= kC3RetainMaxK) {
        w.dump_cap = (uint32_t)(ceil_div(ceil_div(ceil_div(npix, kC3DumpShare), kC3Shards), 64) * 64);   // whole blocks of 64
        w.dump_stride = (uint32_t)(kC3Slices * w.kq * 9);
    } else {
        w.dump_cap = w.dump_stride = 0;
    }
    w.total = w.off_dump + align256((size_t)w.dump_cap * kC3Shards * w.dump_stride * sizeof(float));
    return w;
}

// ND_AMD_C3_RETAIN: 0 = never the time-split pass A, 1 (default) = where it applies; ND_AMD_C3_RETAIN_MIN_K: the
// shortest series it takes (default 2)
template <typename T>
static bool c3_retain_ok(const C3Workspace &w, const C3Args<T> &g)
{
    static const int on = [] {
        const char *e = getenv("ND_AMD_C3_RETAIN");
        return e ? atoi(e) : 1;
    }();
    static const int min_k = [] {
        const char *e = getenv("ND_AMD_C3_RETAIN_MIN_K");
        return e ? atoi(e) : 2;
    }();
    return sizeof(T) == 4 && on != 0 && w.dump_cap > 0 && g.k >= min_k && g.k <= kC3RetainMaxK && g.lane32 != 0;
}

static int c3_launch_retain(const C3Workspace &, C3Args<double> &, const OmniTab &, unsigned char *, hipStream_t)
{
    return ND_AMD_EUNSUPPORTED;          // (float32 only: c3_retain_ok)
}

static int c3_launch_retain(const C3Workspace &w, C3Args<float> &g, const OmniTab &tab, unsigned char *ws,
                            hipStream_t stream)
{
    g.dump = reinterpret_cast<float *>(ws + w.off_dump);
    g.dump_cap = w.dump_cap;
    g.dump_stride = w.dump_stride;
    g.blocks_per_row = ceil_div(g.nx, (int64_t)64);
    const int64_t nblocks = g.blocks_per_row * g.nrows;
    if (nblocks > 0x7fffffffLL) {
        set_error("nd_amd_omnibus_c3: raster too large for one launch");
        return ND_AMD_EUNSUPPORTED;
    }
    const dim3 grid((unsigned)nblocks), block((unsigned)(64 * kC3Slices));
    switch (w.kq) {
    case 4: hipLaunchKernelGGL((omnibus_c3_retain_kernel<4>), grid, block, 0, stream, g, tab); break;
    case 8: hipLaunchKernelGGL((omnibus_c3_retain_kernel<8>), grid, block, 0, stream, g, tab); break;
    case 12: hipLaunchKernelGGL((omnibus_c3_retain_kernel<12>), grid, block, 0, stream, g, tab); break;
    default: hipLaunchKernelGGL((omnibus_c3_retain_kernel<16>), grid, block, 0, stream, g, tab); break;
    }
    return ND_AMD_OK;
}

template <typename T>
static int omnibus_c3_impl(const void *const planes[9], int64_t ny, int64_t nx, int64_t k,
                           int64_t sy, int64_t sx, int64_t st, uint32_t n_looks, double alpha,
                           uint8_t *change, void *z_out, void *p_out, void *workspace,
                           size_t workspace_bytes, hipStream_t stream, const int64_t *pm_ids = nullptr)
{
    const int64_t npix = ny * nx;
    const C3Workspace w = c3_layout(npix, ny, k);
    if (workspace == nullptr || workspace_bytes < w.total) {
        set_error("nd_amd_omnibus_c3: workspace of %zu bytes needed, %zu given", w.total,
                  workspace_bytes);
        return ND_AMD_EWORKSPACE;
    }
    if (((uintptr_t)workspace & 255) != 0) {
        set_error("nd_amd_omnibus_c3: workspace must be 256-byte aligned");
        return ND_AMD_EINVAL;
    }
    if (k > kTabArgs) {
        set_error("nd_amd_omnibus_c3: k = %lld exceeds the supported %d dates", (long long)k, kTabArgs);
        return ND_AMD_EUNSUPPORTED;
    }
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    const std::vector<OmniTabEntry> htab = get_table<T>((int)k, n_looks, alpha, 3);
    OmniTab tab;
    memset(&tab, 0, sizeof(tab));
    memcpy(tab.e, htab.data(), htab.size() * sizeof(OmniTabEntry));

    C3Args<T> g;
    for (int c = 0; c < 9; ++c) {
        g.pl[c] = static_cast<const T *>(planes[c]);
        g.mult[c] = pm_ids ? (int)pm_ids[c] : 1;
    }
    g.pm_vec = 0;
    const bool flat = ((sx == 1) && (sy == nx)) || pm_ids != nullptr;
    g.nx = flat ? npix : nx;
    g.nrows = flat ? 1 : ny;
    g.nx_orig = nx;
    g.sy = sy;
    g.sx = sx;
    g.st = st;
    g.blocks_per_row = ceil_div(g.nx, kC3Threads);
    g.k = (int)k;
    g.write_tab = 1;
    g.off32 = (sx >= 0 && sy >= 0 && st >= 0 &&
               ((nx - 1) * sx + (ny - 1) * sy + (k - 1) * st + 1) * (int64_t)sizeof(T) < 0xffffffffLL) ? 1 : 0;
    g.lane32 = (sx >= 0 && (g.nx - 1) * sx * (int64_t)sizeof(T) < 0xffffffffLL) ? 1 : 0;
    g.nlooks = (double)n_looks;
    g.alpha = alpha;
    g.e = htab[(size_t)k];
    g.change = change;
    g.z_out = static_cast<T *>(z_out);
    g.p_out = static_cast<T *>(p_out);
    g.flag_count = reinterpret_cast<uint32_t *>(ws + w.off_count);
    g.tab_dev = reinterpret_cast<OmniTabEntry *>(ws + w.off_tab);
    g.flag_idx = reinterpret_cast<uint32_t *>(ws + w.off_idx);
    g.seg = w.seg;
    g.starts_max = 0;
    g.dump = nullptr;
    g.dump_cap = g.dump_stride = 0;
    g.retain_rel = 3.f * (21.f * (float)k + 40.f) * 5.9604645e-08f;
    const int64_t nblocks = g.blocks_per_row * g.nrows;
    if (nblocks > 0x7fffffffLL) {
        set_error("nd_amd_omnibus_c3: raster too large for one launch");
        return ND_AMD_EUNSUPPORTED;
    }
    ND_HIP_CHECK(hipMemsetAsync(g.flag_count, 0, kC3CounterBytes, stream));
    const bool stats = z_out != nullptr || p_out != nullptr;
    // Low thresholds: the search fused into the streaming pass (speed only, same map).
    // ND_AMD_C3_FUSED_ALPHA overrides the switch-over (0 = never, 2 = always).
    static const double fused_alpha = [] {
        const char *e = getenv("ND_AMD_C3_FUSED_ALPHA");
        return e ? atof(e) : 0.75;
    }();
    // (round 6) the whole-series test's screen is unusable where omega2 leaves [0, 1] (the default n = 1 on a series of
    // more than a few dates): pass A then evaluates that test exactly and lists only what fires, as the dual-pol entry
    // point does (omnibus.hip, exact_flags) -- instead of handing every pixel to pass B.  ND_AMD_EXACT_FLAGS=0: as before.
    static const bool exact_flags_env = [] {
        const char *e = getenv("ND_AMD_EXACT_FLAGS");
        return e ? atoi(e) != 0 : true;
    }();
    const bool exact_flags = exact_flags_env && pm_ids == nullptr && k >= 2 &&
                             !((htab[(size_t)k].zlo > -INFINITY) || (htab[(size_t)k].zhi < INFINITY));
    const bool fused = pm_ids == nullptr && k >= 2 && k <= kDenseMax && k <= kTabArgs && alpha < fused_alpha && g.off32 &&
                       !exact_flags;
    if (pm_ids != nullptr) {
        // the reference's layout: LDS images folded in place, in the sparse regime (omnibus_c3_pm_kernel)
        constexpr int VE = 16 / (int)sizeof(T);
        bool all_real = true, joint = pm_ids[0] == 1 && pm_ids[1] == 1 && pm_ids[2] == 1;
        for (int c = 0; c < 9; ++c) all_real = all_real && pm_ids[c] == 1;
        for (int c = 3; c < 9; c += 2)
            joint = joint && pm_ids[c] == 2 && pm_ids[c + 1] == 2 && g.pl[c + 1] == g.pl[c] + 1;
        bool aligned = true;
        for (int c = 0; c < 9; ++c)
            if (!(joint && c >= 3 && ((c - 3) & 1))) aligned = aligned && (((uintptr_t)g.pl[c]) & 15) == 0;
        const int64_t per_px = 9 * k * (int64_t)sizeof(T);
        static const int pxw_env = [] {
            const char *e = getenv("ND_AMD_C3_PM_PXW");          // 64 / 32 / 16: pixels per wave (diagnostic)
            return e ? atoi(e) : 0;
        }();
        // images of ~28 KB (five waves per CU, every SIMD with a wave to fold) where that leaves at least 16
        // pixels per wave; 16 pixels up to 56 KB.  48 dates: 32 pixels per wave (55 KB, two waves per CU: two
        // SIMDs idle) 4.56 ms, 16 pixels 3.39 ms
        int pxw = 64 * per_px <= 32 * 1024 ? 64 : (32 * per_px <= 32 * 1024 ? 32 : (16 * per_px <= 56 * 1024 ? 16 : 0));
        if ((pxw_env == 32 || pxw_env == 16) && pxw_env < pxw) pxw = pxw_env;
        if (!(all_real || joint) || !aligned || (k % VE) != 0 || pxw == 0 || !(alpha >= fused_alpha)) {
            set_error("nd_amd_omnibus_c3_pixel_major: nine real (y, x, time) arrays, or three real and three interleaved "
                      "complex ones, 16-byte aligned, a multiple of %d dates up to 56 KB per 16 pixels, alpha >= %g "
                      "(transpose and call nd_amd_omnibus_c3 otherwise)", VE, fused_alpha);
            return ND_AMD_EUNSUPPORTED;
        }
        g.pm_vec = 1;                        // (the checks above are what the 16-byte loads of pass B need as well)
        // (four LANES sharing a pixel's time axis -- 16-byte loads straight into registers, quad broadcasts, no LDS --
        //  were built and measured slower: 3.64 / 4.55 ms against 3.38 / 3.16 ms, DESIGN-EXPERIMENTS.md round 6)
        C3PmArgs pa;
        int off = 0;
        for (int c = 0; c < 9; ++c) {
            if (joint && c >= 3 && ((c - 3) & 1)) {
                pa.img_off[c] = pa.img_off[c - 1];           // the imaginary half: inside the pair's image
                continue;
            }
            pa.img_off[c] = off;
            off += pxw * (int)k * ((joint && c >= 3) ? 2 : 1);
        }
        const size_t lds_pm = (size_t)off * sizeof(T);
        const dim3 gridw((unsigned)ceil_div(npix, (int64_t)pxw)), blockw(64);
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
#define ND_C3_PM(PXW_)                                                                                       \
    do {                                                                                                     \
        if (stats && joint)                                                                                  \
            hipLaunchKernelGGL((omnibus_c3_pm_kernel<T, PXW_, true, true>), gridw, blockw, lds_pm, stream, g, tab, pa);   \
        else if (stats)                                                                                      \
            hipLaunchKernelGGL((omnibus_c3_pm_kernel<T, PXW_, true, false>), gridw, blockw, lds_pm, stream, g, tab, pa);  \
        else if (joint)                                                                                      \
            hipLaunchKernelGGL((omnibus_c3_pm_kernel<T, PXW_, false, true>), gridw, blockw, lds_pm, stream, g, tab, pa);  \
        else                                                                                                 \
            hipLaunchKernelGGL((omnibus_c3_pm_kernel<T, PXW_, false, false>), gridw, blockw, lds_pm, stream, g, tab, pa); \
    } while (0)
        if (pxw == 64)
            ND_C3_PM(64);
        else if (pxw == 32)
            ND_C3_PM(32);
        else
            ND_C3_PM(16);
#undef ND_C3_PM
    } else
    if (!fused && !stats && !exact_flags && c3_retain_ok<T>(w, g)) {
        // the sparse design with the candidates' series handed over from registers (omnibus_c3_retain_kernel)
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
        if (int rc = c3_launch_retain(w, g, tab, ws, stream)) return rc;
    } else
    if (!fused || stats) {
        // the sparse design -- or, with a fused search, only the z / P rasters of it
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
        if (stats || exact_flags)
            hipLaunchKernelGGL((omnibus_c3_global_kernel<T, true>), dim3((unsigned)nblocks),
                               dim3(kC3Threads), 0, stream, g, tab);
        else
            hipLaunchKernelGGL((omnibus_c3_global_kernel<T, false>), dim3((unsigned)nblocks),
                               dim3(kC3Threads), 0, stream, g, tab);
    }
    ND_HIP_CHECK(hipGetLastError());
    if (fused) {
        if (stats) ND_HIP_CHECK(hipMemsetAsync(g.flag_count, 0, kC3CounterBytes, stream));   // its lists are not used
        const DenseScreen scr = make_dense_screen<T>(htab, (int)k, n_looks);
        static const int dense_min = [] {
            const char *e = getenv("ND_AMD_DENSE_MIN");
            return e ? atoi(e) : 16;
        }();
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_FUSED, stream);
        // rounding band of the suffix-sum global tests, 3 x 3: 1.46 j (21 n + 20) u abc / D (the kernel's header)
        const float cu3 = (sizeof(T) == 4 ? 5.9604645e-08f : 1.1102230e-16f) * 1.46f;
        // ND_AMD_C3_FUSED_FORM: 0 the streaming search at every low threshold, 3 the chain search in two
        // streaming passes at every one; unset: the chain search above alpha = 0.02 (where marginal tests over
        // four and more dates stop being rare), speed only
        static const int form_env = [] {
            const char *e = getenv("ND_AMD_C3_FUSED_FORM");
            return e ? atoi(e) : -1;
        }();
        const bool chain = k >= 3 && (form_env == 3 || (form_env < 0 && alpha > 0.02));
        if (k <= 64) {
            StreamScreen<64> ss = make_stream_screen<T, 64>(htab, scr, (int)k, n_looks);
            for (int j = 0; j <= 64; ++j) ss.e[j].cj = cu3 * (21.f * (float)j + 20.f);
            if (chain)
                hipLaunchKernelGGL((omnibus_c3_stream_chain_kernel<T, 1>), dim3((unsigned)nblocks), dim3(kC3Threads), 0,
                                   stream, g, tab, ss, dense_min);
            else
                hipLaunchKernelGGL((omnibus_c3_stream_kernel<T, 1>), dim3((unsigned)nblocks), dim3(kC3Threads), 0,
                                   stream, g, tab, scr, ss, dense_min);
        } else {
            StreamScreen<kDenseMax> ss = make_stream_screen<T, kDenseMax>(htab, scr, (int)k, n_looks);
            for (int j = 0; j <= kDenseMax; ++j) ss.e[j].cj = cu3 * (21.f * (float)j + 20.f);
            if (chain)
                hipLaunchKernelGGL((omnibus_c3_stream_chain_kernel<T, 2>), dim3((unsigned)nblocks), dim3(kC3Threads), 0,
                                   stream, g, tab, ss, dense_min);
            else
                hipLaunchKernelGGL((omnibus_c3_stream_kernel<T, 2>), dim3((unsigned)nblocks), dim3(kC3Threads), 0,
                                   stream, g, tab, scr, ss, dense_min);
        }
        ND_HIP_CHECK(hipGetLastError());
    }

    // behind the streaming search: short lists (what its screen could not decide) one lane per segment
    // start -- ND_AMD_SEARCH_STARTS = list length per shard up to which (0 = never), as for the dual-pol test
    static const int starts_env = [] {
        const char *e = getenv("ND_AMD_SEARCH_STARTS");
        return e ? atoi(e) : 512;
    }();
    if (fused && k >= 3 && starts_env > 0) {
        g.starts_max = (uint32_t)starts_env;
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_SEARCH, stream);
        hipLaunchKernelGGL((omnibus_c3_search_starts_kernel<T>), dim3((unsigned)(32 * kC3Shards)), dim3(64), 0, stream, g);
        ND_HIP_CHECK(hipGetLastError());
    }

    // the LDS image of 64 series: up to 150 KB of the CU's 160 KB (one wave per CU then, which still
    // beats a dependent, TLB-missing plane access per date and lane)
    const size_t scr_bytes = (size_t)(k + 1) * 4 * sizeof(double);
    const size_t lds_bytes = (size_t)k * 9 * 64 * sizeof(T) + scr_bytes;
    // ND_AMD_C3_SEARCH_MODE=1: always from memory (one date of read-ahead, 8+ waves per CU).
    // Measured on config 4's share (48 x 1024 x 8192, 2 % of the pixels listed): 2.02 ms against
    // 1.55 ms with the image at one wave per CU -- of which 1.13 ms is the gather itself: 72 M isolated
    // 4-byte reads = 4.6 GB of 64-byte sectors at 4.1 TB/s, i.e. the pass is bound by the sector
    // traffic of its gather, not by the occupancy the image costs.
    static const int c3_mode = [] {
        const char *e = getenv("ND_AMD_C3_SEARCH_MODE");
        return e ? atoi(e) : 0;
    }();
    const bool use_lds = c3_mode == 1 ? false : lds_bytes <= 150 * 1024;
    if (use_lds && lds_bytes > 64 * 1024) {
        ND_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&omnibus_c3_search_kernel<T, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    int64_t per_shard = ceil_div(ceil_div(npix, kC3Shards), 64);
    if (per_shard > 64) per_shard = 64;
    if (per_shard < 1) per_shard = 1;
    const int64_t sblocks = per_shard * kC3Shards;
    // Pixels per wave: as many as keep the image at ~32 KB, i.e. five waves per CU taking turns at
    // gathering and searching.  Config 4's share (k = 48): 64 / 32 / 16 pixels per wave -> pass B
    // 1.58 / 1.35 / 1.28 ms.  ND_AMD_C3_LANES = 64 / 32 / 16 forces a width.
    static const int c3_lanes_env = [] {
        const char *e = getenv("ND_AMD_C3_LANES");
        return e ? atoi(e) : 0;
    }();
    const int c3_lanes = (c3_lanes_env == 64 || c3_lanes_env == 32 || c3_lanes_env == 16)
                             ? c3_lanes_env
                             : (lds_bytes <= 33 * 1024 ? 64 : (lds_bytes <= 66 * 1024 ? 32 : 16));
    const bool halves = use_lds && c3_lanes != 64;
    if (g.dump != nullptr) {
        // the candidates whose series pass A dumped from its registers
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_SEARCH, stream);
        int64_t per_shard_d = ceil_div((int64_t)g.dump_cap, 64);
        if (per_shard_d > 64) per_shard_d = 64;
        if (per_shard_d < 1) per_shard_d = 1;
        // lockstep rounds on the blocked dump; the lane-per-pixel form (omnibus_c3_search_dump_kernel<T, 1 / 2>) stays
        // for pixel-major inputs, whose series are read where they lie
        const DenseScreen fscr = make_dense_screen<T>(htab, (int)k, n_looks);
        hipLaunchKernelGGL((omnibus_c3_search_rounds_kernel<T>), dim3((unsigned)(per_shard_d * kC3Shards)), dim3(64),
                           (size_t)(k + 1) * sizeof(DenseScreenEntry), stream, g, fscr);
        ND_HIP_CHECK(hipGetLastError());
    }
    // ND_AMD_C3_PM_SEARCH=image: the LDS-image search on pixel-major inputs, as before round 6 (A/B)
    static const bool pm_image = [] {
        const char *e = getenv("ND_AMD_C3_PM_SEARCH");
        return e != nullptr && strcmp(e, "image") == 0;
    }();
    if (g.pm_vec && !pm_image) {
        // pixel-major inputs: every listed pixel read where it lies in the caller's arrays
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_SEARCH, stream);
        int64_t per_shard_d = ceil_div(ceil_div(npix, kC3Shards), 64);
        if (per_shard_d > 64) per_shard_d = 64;
        if (per_shard_d < 1) per_shard_d = 1;
        const dim3 gridd((unsigned)(per_shard_d * kC3Shards));
        if (g.mult[3] == 2)
            hipLaunchKernelGGL((omnibus_c3_search_dump_kernel<T, 2>), gridd, dim3(64), scr_bytes, stream, g);
        else
            hipLaunchKernelGGL((omnibus_c3_search_dump_kernel<T, 1>), gridd, dim3(64), scr_bytes, stream, g);
        ND_HIP_CHECK(hipGetLastError());
    } else
    {
        KernelTimer timer((g.starts_max || g.dump != nullptr) ? ND_AMD_KERNEL_OMNIBUS_EXACT : ND_AMD_KERNEL_OMNIBUS_SEARCH, stream);
        if (halves) {
            const int lanes = c3_lanes == 16 ? 16 : 32;
            const size_t lds_part = (size_t)k * 9 * lanes * sizeof(T) + scr_bytes;
            int64_t per_shard_h = ceil_div(ceil_div(npix, kC3Shards), lanes);
            if (per_shard_h > 256) per_shard_h = 256;
            if (per_shard_h < 1) per_shard_h = 1;
            const dim3 gridh((unsigned)(per_shard_h * kC3Shards));
            if (lanes == 16) {
                if (lds_part > 64 * 1024)
                    ND_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&omnibus_c3_search_kernel<T, true, 16>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part));
                hipLaunchKernelGGL((omnibus_c3_search_kernel<T, true, 16>), gridh, dim3(64), lds_part, stream, g);
            } else {
                if (lds_part > 64 * 1024)
                    ND_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&omnibus_c3_search_kernel<T, true, 32>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part));
                hipLaunchKernelGGL((omnibus_c3_search_kernel<T, true, 32>), gridh, dim3(64), lds_part, stream, g);
            }
        } else if (use_lds)
            hipLaunchKernelGGL((omnibus_c3_search_kernel<T, true>), dim3((unsigned)sblocks), dim3(64),
                               lds_bytes, stream, g);
        else
            hipLaunchKernelGGL((omnibus_c3_search_kernel<T, false>), dim3((unsigned)sblocks),
                               dim3(64), scr_bytes, stream, g);
    }
    ND_HIP_CHECK(hipGetLastError());
    return ND_AMD_OK;
}

}  // namespace nd_amd

using namespace nd_amd;

extern "C" size_t nd_amd_omnibus_c3_workspace_bytes(int64_t ny, int64_t nx, int64_t k)
{
    if (ny < 0 || nx < 0 || k < 0) return 0;
    return c3_layout(ny * nx, ny, k).total;
}

extern "C" int nd_amd_omnibus_c3_pixel_major(const void *const planes[9], int dtype, int64_t ny, int64_t nx,
                                             int64_t k, const int64_t date_stride[9], uint32_t n_looks,
                                             double alpha, uint8_t *change, void *z_out, void *p_out,
                                             void *workspace, size_t workspace_bytes, void *hip_stream)
{
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_omnibus_c3_pixel_major: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (ny < 0 || nx < 0 || k < 0 || !date_stride) {
        set_error("nd_amd_omnibus_c3_pixel_major: bad shape");
        return ND_AMD_EINVAL;
    }
    if (ny == 0 || nx == 0 || k == 0) return ND_AMD_OK;
    if (!planes || !change) {
        set_error("nd_amd_omnibus_c3_pixel_major: null data pointer");
        return ND_AMD_EINVAL;
    }
    for (int c = 0; c < 9; ++c) {
        if (!planes[c]) {
            set_error("nd_amd_omnibus_c3_pixel_major: plane %d is null", c);
            return ND_AMD_EINVAL;
        }
        if (date_stride[c] != 1 && date_stride[c] != 2) {
            set_error("nd_amd_omnibus_c3_pixel_major: date strides must be 1 or 2");
            return ND_AMD_EINVAL;
        }
    }
    if (n_looks == 0) {
        set_error("nd_amd_omnibus_c3_pixel_major: n_looks must be >= 1");
        return ND_AMD_EINVAL;
    }
    if (ny * nx >= 0xffffffffLL) {
        set_error("nd_amd_omnibus_c3_pixel_major: raster exceeds the 32-bit pixel index");
        return ND_AMD_EUNSUPPORTED;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    // unit-stride geometry of the layout: (y, x, t) -> (y * nx + x) * k + t
    if (dtype == ND_AMD_F32)
        return omnibus_c3_impl<float>(planes, ny, nx, k, nx * k, k, 1, n_looks, alpha, change, z_out, p_out,
                                      workspace, workspace_bytes, stream, date_stride);
    return omnibus_c3_impl<double>(planes, ny, nx, k, nx * k, k, 1, n_looks, alpha, change, z_out, p_out,
                                   workspace, workspace_bytes, stream, date_stride);
}

extern "C" int nd_amd_omnibus_c3(const void *const planes[9], int dtype, int64_t ny, int64_t nx,
                                 int64_t k, int64_t stride_y, int64_t stride_x, int64_t stride_t,
                                 uint32_t n_looks, double alpha, uint8_t *change, void *z_out,
                                 void *p_out, void *workspace, size_t workspace_bytes,
                                 void *hip_stream)
{
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_omnibus_c3: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (ny < 0 || nx < 0 || k < 0) {
        set_error("nd_amd_omnibus_c3: negative shape");
        return ND_AMD_EINVAL;
    }
    if (ny == 0 || nx == 0 || k == 0) return ND_AMD_OK;
    if (!planes || !change) {
        set_error("nd_amd_omnibus_c3: null data pointer");
        return ND_AMD_EINVAL;
    }
    for (int c = 0; c < 9; ++c)
        if (!planes[c]) {
            set_error("nd_amd_omnibus_c3: plane %d is null", c);
            return ND_AMD_EINVAL;
        }
    if (n_looks == 0) {
        set_error("nd_amd_omnibus_c3: n_looks must be >= 1");
        return ND_AMD_EINVAL;
    }
    if (ny * nx >= 0xffffffffLL) {
        set_error("nd_amd_omnibus_c3: raster exceeds the 32-bit pixel index");
        return ND_AMD_EUNSUPPORTED;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (dtype == ND_AMD_F32)
        return omnibus_c3_impl<float>(planes, ny, nx, k, stride_y, stride_x, stride_t, n_looks, alpha,
                                      change, z_out, p_out, workspace, workspace_bytes, stream);
    return omnibus_c3_impl<double>(planes, ny, nx, k, stride_y, stride_x, stride_t, n_looks, alpha,
                                   change, z_out, p_out, workspace, workspace_bytes, stream);
}
