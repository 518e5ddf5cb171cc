// nd_amd/csrc/omnibus.hip -- OmnibusTest (complex-Wishart change detector), dual-pol C2,
// for gfx950.  Replaces nd._change.change_detection (nd/_change.pyx:263-287).
//
// Two kernels per call (three at low thresholds, see omnibus_c2_dense_kernel):
//
//   omnibus_c2_global_kernel   ("pass A", HBM-bound)  one thread owns PPT adjacent pixels,
//       streams the k x 4 planes once with 16-byte loads (coalesced along x, time outer),
//       keeps the reference's running state (4 `floating` sums + 1 double product of
//       determinants, nd/_change.pyx:64-69), evaluates the global test over the whole series
//       (nd/_change.pyx:72-76: two double logs), zero-fills its slice of the (y,x,time) change
//       map with 16-byte stores and appends every pixel whose global test CAN fire
//       (z >= a host-computed fast-reject bound, see omni_zlo) to a compact list (wave ballot +
//       one atomic per wave).  When the caller asks for the z / P rasters the chi-square pair
//       is evaluated here for every pixel instead.
//
//   omnibus_c2_search_kernel   ("pass B", FP64-bound)  one lane per listed pixel, series staged
//       in LDS ([time*4+var][lane], conflict free), runs the sequential change-point search of
//       nd/_change.pyx:224-257 as one sweep per segment (one date and at most one test per lane
//       and iteration, so the wave stays converged); tests are decided by host-computed bounds
//       and only the rare in-band ones evaluate the chi-square pair.
//
// Numerics follow the C that Cython generates for the reference (nd/_change.c:3501-3590,
// 6063-6091): `floating` (T) sums and determinants without FMA contraction (this TU is built
// with -ffp-contract=off), double product of determinants, double logs, rho rounded to T,
// z rounded to T, P1/P2 rounded to T, (P2-P1) in T, final combine in double then rounded to T,
// decision (double)P > alpha.  The marginal tests reuse one running accumulation per segment
// start l, which is the same sequence of additions the reference performs when it re-sums
// ts[l:l+j] from scratch for every j.
//
// gsl_cdf_chisq_P(z, f) and (z, f+4) (nd/_change.pyx:147-148): f = (j-1) p^2 with p = 2 is a
// multiple of 4, so a = f/2 = 2(j-1) is an integer and the regularised incomplete gamma
// function has closed recurrences (chisq_pair in omnibus_common.hpp).
#include <type_traits>

#include "omnibus_c2_device.hpp"

namespace nd_amd {

// STATS = false: the kernel only decides which pixels CAN fire (z >= zlo) and lists them; the
//                chi-square evaluation of those happens in pass B.
// STATS = true : the caller wants the z / P rasters, so P is evaluated for every pixel here and
//                the list holds exactly the pixels whose global test fires.
template <typename T, int PPT, bool STATS>
__global__ void __launch_bounds__(kGlobalThreads)
omnibus_c2_global_kernel(const OmniGlobalArgs<T> g, const OmniTab tab)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int64_t b = blockIdx.x;
    const int64_t row = b / g.blocks_per_row;
    const int64_t bx = b - row * g.blocks_per_row;
    const int64_t bpx0 = bx * (int64_t)(kGlobalThreads * PPT);      // first pixel of the block in its row
    const int64_t x0 = bpx0 + (int64_t)tid * PPT;
    const int k = g.k;

    if (g.write_tab && b == 0) {
        for (int j = tid; j <= k; j += kGlobalThreads) g.tab_dev[j] = tab.e[j];
    }

    // ---- zero-fill this block's slice of the change map (np.zeros at nd/_change.pyx:275) ----
    {
        int64_t npx = g.nx - bpx0;
        if (npx > kGlobalThreads * PPT) npx = kGlobalThreads * PPT;
        uint8_t *ob = g.change + (row * g.nx + bpx0) * (int64_t)k;
        const int64_t nb = npx * (int64_t)k;
        int64_t head = (int64_t)((16 - ((uintptr_t)ob & 15)) & 15);
        if (head > nb) head = nb;
        if (tid < head) ob[tid] = 0;
        const int64_t nvec = (nb - head) >> 4;
        uint4 *v = reinterpret_cast<uint4 *>(ob + head);
        for (int64_t i = tid; i < nvec; i += kGlobalThreads) store_zero16_nt(v + i);
        const int64_t tail0 = head + (nvec << 4);
        if (tail0 + tid < nb) ob[tail0 + tid] = 0;
    }

    // ---- stream the series ----
    Accum<T> A[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) A[i].reset();

    const bool any_px = x0 < g.nx;
    const bool full = (x0 + PPT <= g.nx);
    const int64_t off0 = row * g.sy + x0 * g.sx;

    if (any_px) {
        for (int t0 = 0; t0 < k; t0 += kTimeChunk) {
            Pack<T, PPT> v[kTimeChunk][4];
#pragma unroll
            for (int tt = 0; tt < kTimeChunk; ++tt) {
                const int t = t0 + tt;
                if (t < k) {
                    const int64_t off = off0 + (int64_t)t * g.st;
                    if (PPT == 1 || full) {
                        if (PPT == 1) {
                            v[tt][0].v[0] = g.c11[off];
                            v[tt][1].v[0] = g.c12r[off];
                            v[tt][2].v[0] = g.c12i[off];
                            v[tt][3].v[0] = g.c22[off];
                        } else {
                            // read once: past the caches' retention, as in the retaining form (96 dates
                            // on 8.4 Mpx: 2.69 -> 2.21 ms, 0.60 -> 0.73 of the HBM peak)
                            typedef T tvec __attribute__((ext_vector_type(PPT)));
                            const T *const src[4] = {g.c11 + off, g.c12r + off, g.c12i + off, g.c22 + off};
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const tvec q = __builtin_nontemporal_load(reinterpret_cast<const tvec *>(src[c]));
#pragma unroll
                                for (int i = 0; i < PPT; ++i) v[tt][c].v[i] = q[i];
                            }
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < PPT; ++i) {
                            const bool in = (x0 + i < g.nx);
                            v[tt][0].v[i] = in ? g.c11[off + i] : (T)1;
                            v[tt][1].v[i] = in ? g.c12r[off + i] : (T)0;
                            v[tt][2].v[i] = in ? g.c12i[off + i] : (T)0;
                            v[tt][3].v[i] = in ? g.c22[off + i] : (T)1;
                        }
                    }
                }
            }
#pragma unroll
            for (int tt = 0; tt < kTimeChunk; ++tt) {
                if (t0 + tt < k) {
#pragma unroll
                    for (int i = 0; i < PPT; ++i)
                        A[i].step(v[tt][0].v[i], v[tt][1].v[i], v[tt][2].v[i], v[tt][3].v[i]);
                }
            }
        }
    }

    // ---- global test over the whole series ----
    bool flag[PPT];
    unsigned nflag = 0;

    if (STATS) {
        T z[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) z[i] = z_stat<T>(A[i], k, g.nlooks, g.e);
        T P[PPT];
        double zd[PPT], P1[PPT], P2[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) zd[i] = (double)z[i];
        chisq_pair<PPT>(zd, 4 * (k - 1), g.e.lgam, P1, P2);
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            P[i] = combine_P<T>(P1[i], P2[i], g.e.omega2);
            flag[i] = (x0 + i < g.nx) && ((double)P[i] > g.alpha);
        }
        if (any_px) {
            const int64_t pix0 = row * g.nx + x0;
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                if (x0 + i < g.nx) {
                    if (g.z_out) g.z_out[pix0 + i] = z[i];
                    if (g.p_out) g.p_out[pix0 + i] = P[i];
                }
            }
        }
    } else {
        // z_approx < zlo_a (or NaN) cannot fire; everything else is decided exactly in pass B
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            flag[i] = (x0 + i < g.nx) && (z_approx<T>(A[i], k, g.nlooks, g.e) >= g.e.zlo_a);
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) nflag += flag[i] ? 1u : 0u;

    // ---- append candidate pixels to the compact list: one atomic per wave ----
    if (__any(nflag != 0u)) {
        unsigned long long m[PPT];
        unsigned tot = 0;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            m[i] = __ballot(flag[i]);
            tot += (unsigned)__popcll(m[i]);
        }
        const unsigned shard = (unsigned)(b % kShards);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(g.flag_count + shard * kCounterStride, tot);
        base = __shfl(base, 0);
        const unsigned long long lt = (1ull << lane) - 1ull;
        uint32_t *list = g.flag_idx + (size_t)shard * g.seg;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            if (flag[i])
                list[base + (unsigned)__popcll(m[i] & lt)] = (uint32_t)(row * g.nx + x0 + i);
            base += (unsigned)__popcll(m[i]);
        }
    }
}


// -----------------------------------------------------------------------------------------
// pass A, register-retaining form (k <= KMAX): one thread per pixel issues all k x 4 loads at
// once (maximum memory-level parallelism; 4-byte loads, 256 B per wave instruction, coalesced
// along x), then folds them in time order.  Because the series is still in registers when the
// decision falls, a listed pixel writes it to the compact dump ([slot][date][4], 16-byte stores)
// and pass B never has to gather it from the planes again.
// -----------------------------------------------------------------------------------------
// EXACT: k == KMAX and stride_x == 1, so the per-date guards fold away, the loads issue as one
// straight run and address a uniform base plus a 32-bit lane offset
template <typename T, int KMAX, bool EXACT, bool STATS>
__global__ void __launch_bounds__(kRetainThreads)
omnibus_c2_retain_kernel(const OmniGlobalArgs<T> g, const OmniTab tab)
{
    if (omni_gate_skip(g)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int64_t b = blockIdx.x;
    const int64_t row = b / g.blocks_per_row;
    const int64_t bx = b - row * g.blocks_per_row;
    const int64_t bpx0 = bx * (int64_t)kRetainThreads;
    const int64_t x0 = bpx0 + tid;
    const int k = g.k;
    const bool in = x0 < g.nx;

    // ---- issue every load of the series ----
    T v[KMAX][4];
    if (EXACT) {
        // x-contiguous planes: buffer loads through one descriptor per plane (base = first pixel
        // of this block, uniform), a per-date scalar byte offset and a 32-bit lane offset -- no
        // 64-bit per-lane address arithmetic, which keeps the kernel at 4 waves per SIMD.
        const int64_t ub = row * g.sy + bpx0;
        const unsigned lx = in ? (unsigned)tid : (unsigned)(g.nx - 1 - bpx0);   // idle lanes re-read the last pixel
        const unsigned voff = lx * (unsigned)sizeof(T);
        const auto r11 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c11 + ub), 0, 0x7fffffff, 0x00020000);
        const auto r12r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c12r + ub), 0, 0x7fffffff, 0x00020000);
        const auto r12i = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c12i + ub), 0, 0x7fffffff, 0x00020000);
        const auto r22 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c22 + ub), 0, 0x7fffffff, 0x00020000);
        const unsigned sstep = (unsigned)g.st * (unsigned)sizeof(T);   // host guarantees k * st * sizeof(T) < 2^31
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            const unsigned soff = (unsigned)t * sstep;
            v[t][0] = buffer_load<T>(r11, voff, soff);
            v[t][1] = buffer_load<T>(r12r, voff, soff);
            v[t][2] = buffer_load<T>(r12i, voff, soff);
            v[t][3] = buffer_load<T>(r22, voff, soff);
        }
    } else {
        const int64_t xc = in ? x0 : g.nx - 1;
        const int64_t off0 = row * g.sy + xc * g.sx;
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            if (t < k) {
                const int64_t off = off0 + (int64_t)t * g.st;
                v[t][0] = __builtin_nontemporal_load(g.c11 + off);
                v[t][1] = __builtin_nontemporal_load(g.c12r + off);
                v[t][2] = __builtin_nontemporal_load(g.c12i + off);
                v[t][3] = __builtin_nontemporal_load(g.c22 + off);
            }
        }
    }

    if (g.write_tab && b == 0) {
        for (int j = tid; j <= k; j += kRetainThreads) g.tab_dev[j] = tab.e[j];
    }

    // ---- fold in time order ----
    Accum<T> A;
    A.reset();
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
        if (EXACT || t < k) A.step(v[t][0], v[t][1], v[t][2], v[t][3]);

    bool flag;
    if (STATS) {
        const T z = z_stat<T>(A, k, g.nlooks, g.e);
        double zd[1] = {(double)z}, P1[1], P2[1];
        chisq_pair<1>(zd, 4 * (k - 1), g.e.lgam, P1, P2);
        const T P = combine_P<T>(P1[0], P2[0], g.e.omega2);
        flag = in && ((double)P > g.alpha);
        if (in) {
            const int64_t pix = row * g.nx + x0;
            if (g.z_out) g.z_out[pix] = z;
            if (g.p_out) g.p_out[pix] = P;
        }
    } else {
        flag = in && (z_approx<T>(A, k, g.nlooks, g.e) >= g.e.zlo_a);
    }

    // ---- list + dump ----
    const unsigned long long m = __ballot(flag);
    const bool dense_wave = __popcll(m) >= g.dense_min;
    if (m != 0ull) {
        const unsigned shard = (unsigned)(b % kShards);
        if (dense_wave) {
            // dense wave: one entry for all 64 pixels, no dump (the dense kernel reloads them,
            // coalesced, and decides every pixel itself)
            if (lane == 0) {
                const unsigned slot = atomicAdd(g.flag_count + shard * kCounterStride + 1, 1u);
                g.dense_idx[(size_t)shard * g.segd + slot] = (uint32_t)(row * g.nx + bpx0 + (tid & ~63));
            }
        } else {
            unsigned base = 0;
            if (lane == 0)
                base = atomicAdd(g.flag_count + shard * kCounterStride, (unsigned)__popcll(m));
            base = __shfl(base, 0);
            if (flag) {
                const unsigned slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                g.flag_idx[(size_t)shard * g.seg + slot] = (uint32_t)(row * g.nx + x0);
                if (slot < g.dump_cap) {
                    T *d = g.dump + ((int64_t)shard * g.dump_cap + slot) * (int64_t)(4 * k);
#pragma unroll
                    for (int t = 0; t < KMAX; ++t) {
                        if (EXACT || t < k) {
                            Pack<T, 4> q;
                            q.v[0] = v[t][0];
                            q.v[1] = v[t][1];
                            q.v[2] = v[t][2];
                            q.v[3] = v[t][3];
                            *reinterpret_cast<Pack<T, 4> *>(d + 4 * t) = q;
                        }
                    }
                }
            }
        }
    }

    // ---- zero-fill this block's slice of the change map (np.zeros at nd/_change.pyx:275).
    // Issued last so that no wait on the loads or on the atomic above also has to wait for
    // these stores (vmcnt retires in order). ----
    {
        const int64_t left = g.nx - bpx0;
        const int npx = left > kRetainThreads ? kRetainThreads : (int)left;
        uint8_t *ob = g.change + (row * g.nx + bpx0) * (int64_t)k;
        const int nb = npx * k;                              // <= 256 * k bytes
        int head = (int)((16 - ((uintptr_t)ob & 15)) & 15);
        if (head > nb) head = nb;
        if (tid < head) ob[tid] = 0;
        const int nvec = (nb - head) >> 4;
        uint4 *vz = reinterpret_cast<uint4 *>(ob + head);
        for (int i = tid; i < nvec; i += kRetainThreads) store_zero16_nt(vz + i);
        const int tail0 = head + (nvec << 4);
        if (tail0 + tid < nb) ob[tail0 + tid] = 0;
    }
}

// -----------------------------------------------------------------------------------------
// pass A, TIME-SPLIT register-retaining form (round 6; float32 49 .. 192 dates, float64 25 .. 96, sparse regime).
// Beyond 48 dates one thread cannot hold a pixel's series, the plain pass A did not dump, and pass B
// gathered every listed pixel again from the planes: 4 k isolated 4-byte reads, one 64-byte sector each
// (96 dates x 2048 x 4096 at alpha = 0.99: 1.3 ms of a 3.45 ms call, for data pass A had just read).
// Here NS waves share the time axis of 64 pixels: wave w loads the KQ dates [w KQ, (w + 1) KQ) of its 64
// pixels (4 KQ registers, every load in flight at once), folds its slice into four partial sums and a
// partial product of determinants; the partials meet in LDS, the group's first wave screens the combined
// value and lists the candidates, and every wave stores its slice of a candidate's series from its
// registers into the dump ([slot][date][4], 16-byte stores) that pass B reads.
//
// Pass A only screens; exactness lives in pass B, which folds in the reference's order
// (nd/_change.pyx:64-69).  What differs from the forward fold and how the screen pays for it (the
// full-pol twin, omnibus_c3_retain_kernel, has the long form):
//   * the sums are re-associated (((s0 + s1) + s2) + ...).  Every date positive semi-definite (c11 > 0,
//     det > 0: checked, anything else is listed) bounds both orders' errors: |D' - D_ref| <=
//     2 (5 k + 4) u s11 s22 (the streaming search's bound, once per order).  The screen adds
//     |m2rho| n k 1.02 rel, rel = 3 (5 k + 8) u s11 s22 / D', to z_approx; rel >= 0.01 or sums
//     outside [2^-40, 2^40]: listed.
//   * the product of determinants is the product of the slices' double products -- NS - 1 more roundings
//     of 2^-53, inside the tenfold margin of zlo_a -- provided no prefix product the reference forms
//     leaves the normal range: the waves track the exponents their running products pass through, and a
//     pixel whose prefix exponents (slice offsets added) leave +-1000 is listed.
//   * a determinant that is NaN or exactly 0 makes the whole-series statistic NaN or infinite: no change
//     anywhere, not listed (as the plain screen does not list it).
// -----------------------------------------------------------------------------------------
// STATS (round 6, later): the z / P rasters of the whole-series test from the same launch.  They are the reference's
// FORWARD fold (nd/_change.pyx:53-77: four sums in `floating`, the double product, date by date) -- not the
// re-associated sums of the screen -- so the fold is handed from wave to wave through LDS in time order, bit for bit
// (six instructions per date, one wave after the other: about a microsecond per block), and the last slice's wave
// evaluates z_stat / chisq_pair / combine_P exactly as omnibus_c2_retain_kernel<..., STATS> does: the rasters are
// identical to those of every other form.  (Before: a call with rasters took the plain pass A and the gather.)
template <typename T, int KQ, int NS, bool STATS = false>
__global__ void __launch_bounds__(64 * NS)
omnibus_c2_split_kernel(const OmniGlobalArgs<T> g, const OmniTab tab, const float retain_rel)
{
    __shared__ T part_s[NS][4][64];
    __shared__ T fwd_s[STATS ? 4 : 1][64];
    __shared__ double fwd_p[64];
    __shared__ double part_p[NS][64];
    __shared__ int part_e[NS][3][64];              // lowest / highest exponent of the slice's running product, flags
    __shared__ unsigned long long flag_mask;
    __shared__ unsigned list_base;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b = blockIdx.x;
    const int64_t row = b / g.blocks_per_row;
    const int64_t bx = b - row * g.blocks_per_row;
    const int64_t gpx0 = bx * 64;
    const int64_t x0 = gpx0 + lane;
    const int k = g.k;
    const bool in = x0 < g.nx;
    const int t_lo = w * KQ;

    if (g.write_tab && b == 0)
        for (int j = tid; j <= k; j += 64 * NS) g.tab_dev[j] = tab.e[j];

    // every load of the slice in flight: a uniform 64-bit base per plane and date (scalar arithmetic) plus
    // a 32-bit lane offset -- `global_load_dword v, v_off, s[base]`, no per-lane 64-bit addresses, and no
    // limit on the extent of a plane stack (a buffer descriptor's offsets end at 2 GB)
    T v[KQ][4];
    {
        const int64_t ub = row * g.sy + gpx0;
        const unsigned lx = in ? (unsigned)lane : (unsigned)(g.nx - 1 - gpx0);   // idle lanes re-read the last pixel
        const unsigned voff = lx * (unsigned)sizeof(T);
#pragma unroll
        for (int tt = 0; tt < KQ; ++tt) {
            const int t = t_lo + tt < k ? t_lo + tt : k - 1;               // (behind the series: the last date again)
            const int64_t uo = ub + (int64_t)t * g.st;
            v[tt][0] = __builtin_nontemporal_load(reinterpret_cast<const T *>(reinterpret_cast<const char *>(g.c11 + uo) + voff));
            v[tt][1] = __builtin_nontemporal_load(reinterpret_cast<const T *>(reinterpret_cast<const char *>(g.c12r + uo) + voff));
            v[tt][2] = __builtin_nontemporal_load(reinterpret_cast<const T *>(reinterpret_cast<const char *>(g.c12i + uo) + voff));
            v[tt][3] = __builtin_nontemporal_load(reinterpret_cast<const T *>(reinterpret_cast<const char *>(g.c22 + uo) + voff));
        }
    }

    // zero-fill this wave's share of the group's slice of the change map (np.zeros, nd/_change.pyx:275)
    {
        const int64_t left = g.nx - gpx0;
        const int npx = left > 64 ? 64 : (int)left;
        uint8_t *ob = g.change + (row * g.nx + gpx0) * (int64_t)k;
        const int nb = npx * k;
        int head = (int)((16 - ((uintptr_t)ob & 15)) & 15);
        if (head > nb) head = nb;
        if (w == 0 && lane < head) ob[lane] = 0;
        const int nvec = (nb - head) >> 4;
        uint4 *vz = reinterpret_cast<uint4 *>(ob + head);
        for (int i = w * 64 + lane; i < nvec; i += 64 * NS) store_zero16_nt(vz + i);
        const int tail0 = head + (nvec << 4);
        if (w == 0 && tail0 + lane < nb) ob[tail0 + lane] = 0;
    }

    // fold the slice in time order
    T s11 = 0, s12r = 0, s12i = 0, s22 = 0;
    double prod = 1.0;
    int emin = 1, emax = 1;                        // frexp exponent of 1.0
    bool bad = false, dead = false;
#pragma unroll
    for (int tt = 0; tt < KQ; ++tt) {
        if (t_lo + tt < k) {                       // wave-uniform
            const T a = v[tt][0], br = v[tt][1], bi = v[tt][2], d = v[tt][3];
            const T det = (a * d) - ((br * br) + (bi * bi));
            bad = bad | !((a > (T)0) & (det > (T)0));
            dead = dead | !((det > (T)0) | (det < (T)0));
            prod = prod * (double)det;
            const int e = __builtin_amdgcn_frexp_exp(prod);
            emin = e < emin ? e : emin;
            emax = e > emax ? e : emax;
            s11 = s11 + a;
            s12r = s12r + br;
            s12i = s12i + bi;
            s22 = s22 + d;
        }
    }
    part_s[w][0][lane] = s11;
    part_s[w][1][lane] = s12r;
    part_s[w][2][lane] = s12i;
    part_s[w][3][lane] = s22;
    part_p[w][lane] = prod;
    part_e[w][0][lane] = emin;
    part_e[w][1][lane] = emax;
    part_e[w][2][lane] = (bad ? 1 : 0) | (dead ? 2 : 0);
    __syncthreads();

    const unsigned shard = (unsigned)(b % kShards);
    if (w == 0) {
        T S11 = 0, S12r = 0, S12i = 0, S22 = 0;
        double PP = 1.0;
        int fl = 0, eoff = 0;
        bool range_ok = true;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            if (u * KQ < k) {
                const double pu = part_p[u][lane];
                const int lo = part_e[u][0][lane], hi = part_e[u][1][lane];
                range_ok = range_ok & (lo > -1000) & (hi < 1000) & (eoff + lo > -1000) & (eoff + hi < 1000) &
                           (pu > 0.0) & (pu < INFINITY);
                eoff += __builtin_amdgcn_frexp_exp(pu);
                if (u == 0) {
                    S11 = part_s[0][0][lane];
                    S12r = part_s[0][1][lane];
                    S12i = part_s[0][2][lane];
                    S22 = part_s[0][3][lane];
                    PP = pu;
                } else {
                    S11 = S11 + part_s[u][0][lane];
                    S12r = S12r + part_s[u][1][lane];
                    S12i = S12i + part_s[u][2][lane];
                    S22 = S22 + part_s[u][3][lane];
                    PP = PP * pu;
                }
                fl |= part_e[u][2][lane];
            }
        }
        const bool isdead = (fl & 2) != 0;
        bool isbad = ((fl & 1) != 0) | !range_ok;
        const T dd = S11 * S22;
        const T det_of_sum = dd - ((S12r * S12r) + (S12i * S12i));
        const T smin = S11 < S22 ? S11 : S22, smax = S11 > S22 ? S11 : S22;
        // (no under- or overflow inside the determinant of the sums: 2^+-40 in float32, 2^+-400 in float64)
        isbad = isbad | !((smin > (sizeof(T) == 4 ? (T)9.094947e-13 : (T)3.8725919148493183e-121)) &
                          (smax < (sizeof(T) == 4 ? (T)1.0995116e12 : (T)2.5822498780869086e120)));
        const float rel = retain_rel * ((float)dd * __builtin_amdgcn_rcpf((float)det_of_sum));
        isbad = isbad | !((det_of_sum > (T)0) & (rel < 0.01f));
        const double logQ = g.nlooks * ((g.e.pklogk + approx_ln(PP)) - ((double)k * approx_ln((double)det_of_sum)));
        const double za = g.e.m2rho * logQ;
        const double mz = (fabs(g.e.m2rho) * g.nlooks * (double)k * 1.02) * (double)rel;
        const bool flag = in && !isdead && (isbad || (za + mz >= g.e.zlo_a));
        const unsigned long long m = __ballot(flag);
        unsigned base = 0;
        if (m != 0ull) {
            if (lane == 0) base = atomicAdd(g.flag_count + shard * kCounterStride, (unsigned)__popcll(m));
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
    // STATS: the reference's forward fold, slice after slice
    Accum<T> W;
    W.reset();
    const bool last_slice = t_lo < k && t_lo + KQ >= k;
    if (STATS) {
#pragma unroll 1
        for (int sl = 0; sl < NS; ++sl) {
            if (w == sl && t_lo < k) {
                if (sl > 0) {
                    W.s11 = fwd_s[0][lane];
                    W.s12r = fwd_s[1][lane];
                    W.s12i = fwd_s[2][lane];
                    W.s22 = fwd_s[3][lane];
                    W.prod = fwd_p[lane];
                }
#pragma unroll
                for (int tt = 0; tt < KQ; ++tt)
                    if (t_lo + tt < k) W.step(v[tt][0], v[tt][1], v[tt][2], v[tt][3]);
                if (!last_slice) {
                    fwd_s[0][lane] = W.s11;
                    fwd_s[1][lane] = W.s12r;
                    fwd_s[2][lane] = W.s12i;
                    fwd_s[3][lane] = W.s22;
                    fwd_p[lane] = W.prod;
                }
            }
            __syncthreads();
        }
    }
    // a candidate's slice leaves the registers: one 16-byte store per date
    const unsigned long long m = flag_mask;
    if (m != 0ull && t_lo < k) {
        const unsigned slot = list_base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (((m >> lane) & 1ull) && slot < g.dump_cap) {
            // BLOCKED layout (dump_cap is a multiple of 64 here): [shard][slot / 64][date][slot % 64][4] -- the 64
            // series of a block of list entries interleaved date by date, so that pass B, which gives those entries
            // the 64 lanes of a wave and walks the dates in lockstep (omnibus_c2_search_rounds_kernel), reads one
            // contiguous kilobyte per date
            T *dptr = g.dump + ((((int64_t)shard * (g.dump_cap >> 6) + (slot >> 6)) * (int64_t)k + t_lo) * 64 + (slot & 63u)) * 4;
#pragma unroll
            for (int tt = 0; tt < KQ; ++tt) {
                if (t_lo + tt < k) {
                    Pack<T, 4> q;
                    q.v[0] = v[tt][0];
                    q.v[1] = v[tt][1];
                    q.v[2] = v[tt][2];
                    q.v[3] = v[tt][3];
                    *reinterpret_cast<Pack<T, 4> *>(dptr + (int64_t)tt * 256) = q;
                }
            }
        }
    }
    // (behind the dump: the series is dead here, the chi-square series have the registers to themselves)
    if (STATS && last_slice) {
        __builtin_amdgcn_sched_barrier(0);
        const T z = z_stat<T>(W, k, g.nlooks, g.e);
        double zd[1] = {(double)z}, P1[1], P2[1];
        chisq_pair<1>(zd, 4 * (k - 1), g.e.lgam, P1, P2);
        const T P = combine_P<T>(P1[0], P2[0], g.e.omega2);
        if (in) {
            const int64_t pix = row * g.nx + x0;
            if (g.z_out) g.z_out[pix] = z;
            if (g.p_out) g.p_out[pix] = P;
        }
    }
}

// -----------------------------------------------------------------------------------------
// How dense is the raster?  A few hundred pixel blocks spread over it, one pixel per thread,
// streaming fold of the whole series and the same global screen as pass A; the number of
// candidates goes to *gate_out.  ~1 % of the data: microseconds.
// -----------------------------------------------------------------------------------------
template <typename T>
struct OmniSampleArgs {
    const T *c11, *c12r, *c12i, *c22;
    int64_t nx, blocks_per_row, block_stride;   // sampled pixel block = blockIdx.x * block_stride
    int64_t sy, sx, st;
    int m11, m12, m22;                          // element multipliers (pixel-major inputs: 1 or 2)
    int k;
    double nlooks;
    OmniTabEntry e;
    uint32_t *gate_out;
};

template <typename T>
__global__ void __launch_bounds__(kRetainThreads) omnibus_c2_sample_kernel(const OmniSampleArgs<T> a)
{
    const int tid = threadIdx.x;
    // one block out of every `block_stride`, at a pseudo-random place inside its interval: a plain
    // stride aliases with the row length (4096-pixel rows, stride 64: every sample in the first
    // 256 columns -- a nodata margin there made the gate see an empty raster)
    const int64_t jitter = (int64_t)(((unsigned)blockIdx.x * 0x9E3779B1u) >> 8) % a.block_stride;
    const int64_t b = (int64_t)blockIdx.x * a.block_stride + jitter;
    const int64_t row = b / a.blocks_per_row;
    const int64_t x0 = (b - row * a.blocks_per_row) * (int64_t)kRetainThreads + tid;
    const bool in = x0 < a.nx;
    const int64_t off0 = row * a.sy + (in ? x0 : a.nx - 1) * a.sx;
    Accum<T> A;
    A.reset();
    // (twelve dates = 48 loads in flight per thread: the kernel is a few dependent round trips to
    // memory and nothing else, and everything behind it waits for its verdict)
    constexpr int CH = 12;
    for (int t0 = 0; t0 < a.k; t0 += CH) {
        T q[CH][4];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int64_t o = off0 + (int64_t)(t0 + u < a.k ? t0 + u : a.k - 1) * a.st;
            q[u][0] = a.c11[o * a.m11];
            q[u][1] = a.c12r[o * a.m12];
            q[u][2] = a.c12i[o * a.m12];
            q[u][3] = a.c22[o * a.m22];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (t0 + u < a.k) A.step(q[u][0], q[u][1], q[u][2], q[u][3]);
    }
    const bool flag = in && (z_approx<T>(A, a.k, a.nlooks, a.e) >= a.e.zlo_a);
    // one atomic per block: all of them hit one address, which serialises them (one per wave: 2048
    // atomics at ~88 per microsecond were more than half of this kernel's 41 microseconds)
    __shared__ unsigned wave_hits[kRetainThreads / 64];
    const unsigned long long m = __ballot(flag);
    if ((tid & 63) == 0) wave_hits[tid >> 6] = (unsigned)__popcll(m);
    __syncthreads();
    if (tid == 0) {
        unsigned hits = 0;
#pragma unroll
        for (int w = 0; w < kRetainThreads / 64; ++w) hits += wave_hits[w];
        if (hits != 0u) atomicAdd(a.gate_out, hits);
    }
}

// -----------------------------------------------------------------------------------------
// pass A for data in the reference's own layout: every variable is [pixel][date] with the dates
// of a pixel adjacent (nd/change.py:66-67 hands (y, x, time, variable) views of such arrays to the
// native code), C12 possibly interleaved complex.  A block's 256 pixels are one contiguous span
// per variable: the spans are read with fully coalesced loads into an LDS image (pixel-major,
// odd pitch), each thread then picks its own pixel's series out of it, and the kernel continues
// exactly like the register-retaining form above (same fold, screen, list, dump, zero-fill).
// Two variables are staged per round: C11 with C22, then the two halves of C12.
// -----------------------------------------------------------------------------------------
template <typename T>
struct OmniPmArgs {
    int ids[4];               // element distance between consecutive dates, per variable (1 or 2)
    unsigned magic[4];        // ceil(2^32 / (k * ids)): span element -> pixel without a division
    int c12_joint;            // C12 re / im are the halves of one interleaved array: one read
};

template <typename T, int KMAX, bool STATS>
__global__ void __launch_bounds__(kRetainThreads)
omnibus_c2_retain_pm_kernel(const OmniGlobalArgs<T> g, const OmniTab tab, const OmniPmArgs<T> pm)
{
    extern __shared__ __align__(16) unsigned char nd_smem_pm[];
    T *img = reinterpret_cast<T *>(nd_smem_pm);            // two images of kRetainThreads x (k | 1)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int64_t b = blockIdx.x;
    const int64_t row = 0;                                 // the raster is flat in this layout
    const int64_t bpx0 = b * (int64_t)kRetainThreads;
    const int64_t x0 = bpx0 + tid;
    const int k = g.k;
    const bool in = x0 < g.nx;
    const int64_t left = g.nx - bpx0;
    const int np = left > kRetainThreads ? kRetainThreads : (int)left;
    const int pitch = k | 1;
    const int imgsz = kRetainThreads * pitch;

    // stage the contiguous span of `np` pixels of one variable into image `slot`
    auto stage = [&](const T *base, int vi, int slot, bool both_halves) {
        const int ids = pm.ids[vi];
        const int spp = k * ids;
        const int total = np * spp;
        const T *src = base + bpx0 * spp;
        T *dst = img + slot * imgsz;
        for (int e0 = 0; e0 < total; e0 += kRetainThreads * 8) {
            T buf[8];
            int di[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * kRetainThreads + tid;
                di[u] = -1;
                if (e < total) {
                    const int q = (int)__umulhi((unsigned)e, pm.magic[vi]);
                    const int r = e - q * spp;
                    if (ids == 1)
                        di[u] = q * pitch + r;
                    else if (both_halves)                    // even -> this image, odd -> the next
                        di[u] = (r & 1) * imgsz + q * pitch + (r >> 1);
                    else if ((r & 1) == 0)
                        di[u] = q * pitch + (r >> 1);
                    if (di[u] >= 0) buf[u] = __builtin_nontemporal_load(src + e);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (di[u] >= 0) dst[di[u]] = buf[u];
        }
    };

    T v[KMAX][4];
    const int own = (in ? tid : np - 1) * pitch;            // idle lanes copy the last pixel
    // round 1: C11 and C22
    stage(g.c11, 0, 0, false);
    stage(g.c22, 3, 1, false);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
        if (t < k) {
            v[t][0] = img[own + t];
            v[t][3] = img[imgsz + own + t];
        }
    __syncthreads();
    // round 2: the two halves of C12
    if (pm.c12_joint) {
        stage(g.c12r, 1, 0, true);
    } else {
        stage(g.c12r, 1, 0, false);
        stage(g.c12i, 2, 1, false);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
        if (t < k) {
            v[t][1] = img[own + t];
            v[t][2] = img[imgsz + own + t];
        }
    constexpr bool EXACT = false;
    (void)EXACT;

    if (g.write_tab && b == 0) {
        for (int j = tid; j <= k; j += kRetainThreads) g.tab_dev[j] = tab.e[j];
    }

    // ---- fold in time order ----
    Accum<T> A;
    A.reset();
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
        if (t < k) A.step(v[t][0], v[t][1], v[t][2], v[t][3]);

    bool flag;
    if (STATS) {
        const T z = z_stat<T>(A, k, g.nlooks, g.e);
        double zd[1] = {(double)z}, P1[1], P2[1];
        chisq_pair<1>(zd, 4 * (k - 1), g.e.lgam, P1, P2);
        const T P = combine_P<T>(P1[0], P2[0], g.e.omega2);
        flag = in && ((double)P > g.alpha);
        if (in) {
            const int64_t pix = row * g.nx + x0;
            if (g.z_out) g.z_out[pix] = z;
            if (g.p_out) g.p_out[pix] = P;
        }
    } else {
        flag = in && (z_approx<T>(A, k, g.nlooks, g.e) >= g.e.zlo_a);
    }

    // ---- list + dump ----
    const unsigned long long m = __ballot(flag);
    const bool dense_wave = __popcll(m) >= g.dense_min;
    if (m != 0ull) {
        const unsigned shard = (unsigned)(b % kShards);
        if (dense_wave) {
            // dense wave: one entry for all 64 pixels, no dump (the dense kernel reloads them,
            // coalesced, and decides every pixel itself)
            if (lane == 0) {
                const unsigned slot = atomicAdd(g.flag_count + shard * kCounterStride + 1, 1u);
                g.dense_idx[(size_t)shard * g.segd + slot] = (uint32_t)(row * g.nx + bpx0 + (tid & ~63));
            }
        } else {
            unsigned base = 0;
            if (lane == 0)
                base = atomicAdd(g.flag_count + shard * kCounterStride, (unsigned)__popcll(m));
            base = __shfl(base, 0);
            if (flag) {
                const unsigned slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                g.flag_idx[(size_t)shard * g.seg + slot] = (uint32_t)(row * g.nx + x0);
                if (slot < g.dump_cap) {
                    T *d = g.dump + ((int64_t)shard * g.dump_cap + slot) * (int64_t)(4 * k);
#pragma unroll
                    for (int t = 0; t < KMAX; ++t) {
                        if (t < k) {
                            Pack<T, 4> q;
                            q.v[0] = v[t][0];
                            q.v[1] = v[t][1];
                            q.v[2] = v[t][2];
                            q.v[3] = v[t][3];
                            *reinterpret_cast<Pack<T, 4> *>(d + 4 * t) = q;
                        }
                    }
                }
            }
        }
    }

    // ---- zero-fill this block's slice of the change map (np.zeros at nd/_change.pyx:275).
    // Issued last so that no wait on the loads or on the atomic above also has to wait for
    // these stores (vmcnt retires in order). ----
    {
        const int64_t left = g.nx - bpx0;
        const int npx = left > kRetainThreads ? kRetainThreads : (int)left;
        uint8_t *ob = g.change + (row * g.nx + bpx0) * (int64_t)k;
        const int nb = npx * k;                              // <= 256 * k bytes
        int head = (int)((16 - ((uintptr_t)ob & 15)) & 15);
        if (head > nb) head = nb;
        if (tid < head) ob[tid] = 0;
        const int nvec = (nb - head) >> 4;
        uint4 *vz = reinterpret_cast<uint4 *>(ob + head);
        for (int i = tid; i < nvec; i += kRetainThreads) store_zero16_nt(vz + i);
        const int tail0 = head + (nvec << 4);
        if (tail0 + tid < nb) ob[tail0 + tid] = 0;
    }
}



// -----------------------------------------------------------------------------------------
// pass A for the reference's layout, LDS-DMA form.  One wave per 64 pixels: the wave's span of
// every variable (64 pixels x k dates, contiguous in memory) goes straight from memory into a
// wave-private LDS image with `global_load_lds_dwordx4` -- 16 bytes per lane, 1 KiB per instruction,
// no register staging, every transfer of the wave (24 KiB at k = 24) in flight at once, and no
// workgroup barrier: a wave reads only what it loaded itself, after its own s_waitcnt vmcnt(0).
// Each lane then picks its pixel's series out of the image with 16-byte LDS reads and the kernel
// continues like the register-retaining form (same fold, screen, list, dump, zero-fill).
// Needs 16-byte aligned variables and k a multiple of 16 / sizeof(T); the staged form above serves
// the rest.  (omnibus_c2_retain_pm_kernel: two dependent load -> LDS -> register rounds, 0.41 of
// the HBM peak at k = 24; this form: see DESIGN.md.)
// -----------------------------------------------------------------------------------------

template <typename T>
struct OmniPmDmaArgs {
    int ids[4];               // element distance between consecutive dates, per variable (1 or 2)
    int c12_joint;            // C12 re / im are the halves of one interleaved array: one image
    int img_off[4];           // element offset of each variable's image in the wave's LDS region
};

// one variable of this lane's series out of its LDS image (16-byte reads; all register indices
// static).  BOTH: the image is interleaved complex and both halves are wanted (components COMP,
// COMP + 1); otherwise a stride-2 image carries the wanted half at the even offsets.
template <typename T, int KMAX, int COMP, bool BOTH>
__device__ __forceinline__ void pm_pick(T (&v)[KMAX][4], const T *im, const int k, const int ids)
{
    constexpr int VE = 16 / (int)sizeof(T);
    // (round 6) a series length that is not a multiple of the 16-byte vector -- three of four lengths in float32 --
    // leaves the lanes' runs unaligned in the image: element by element then (wave-uniform branch).  Before, such
    // lengths took the register-staged kernel and, below the sparse regime, pass B for every pixel: 21 dates x
    // 2048 x 4096 at alpha = 0.01 2.9 ms against 0.85 ms for 24 dates.
    if (((k * ids) % VE) != 0) {
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            if (t < k) {
                v[t][COMP] = im[t * ids];
                if (BOTH) v[t][COMP + (BOTH ? 1 : 0)] = im[t * ids + 1];
            }
        }
        return;
    }
    if (ids == 1) {
#pragma unroll
        for (int u = 0; u < KMAX / VE; ++u) {
            if (u * VE < k) {
                const Pack<T, VE> q = *reinterpret_cast<const Pack<T, VE> *>(im + u * VE);
#pragma unroll
                for (int i = 0; i < VE; ++i) v[u * VE + i][COMP] = q.v[i];
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < 2 * KMAX / VE; ++u) {
            if (u * VE < 2 * k) {
                const Pack<T, VE> q = *reinterpret_cast<const Pack<T, VE> *>(im + u * VE);
#pragma unroll
                for (int i = 0; i < VE; i += 2) {
                    v[(u * VE + i) / 2][COMP] = q.v[i];
                    if (BOTH) v[(u * VE + i) / 2][COMP + (BOTH ? 1 : 0)] = q.v[i + 1];
                }
            }
        }
    }
}


// CHAIN: the search fused in (dense_chain on the registers the series was picked into), for
// thresholds between the streaming search's and the sparse regime; `ss` is only read then.
// DIRECT (with CHAIN): only C12 goes through an LDS image; each lane reads its own C11 / C22 series
// (k contiguous values) straight into registers, every 16-byte piece in flight at once -- the lines
// are shared by neighbouring lanes and consecutive pieces, so they are fetched once -- and the image
// is half the size: twice the waves per CU (24 x 4096^2: 2.5 -> 2.0 ms).  C12 as well, no image at
// all: 3.3 ms with non-temporal loads (round 3), 1.49 against 1.45 ms with plain ones (round 5): no gain.
// (CHAIN, float32: capped at the registers of three waves per SIMD -- the 24-date DIRECT form took 170,
// two short of it, and ran at two: 1.99 -> 1.86 ms; the float64 forms would spill under the cap.  A
// persistent form -- as many waves as the chip holds, each walking spans gridDim.x apart, to spare the
// dispatch gap between 13 us blocks -- was measured at 3.4 ms: with the span loop around it the body
// needs more than the cap however its invariants are hidden from the hoisting passes.)
template <typename T, int KMAX, bool STATS, bool CHAIN = false, bool DIRECT = false>
__global__ void __launch_bounds__(64, (CHAIN && sizeof(T) == 4) ? 3 : 1)
omnibus_c2_pm_dma_kernel(const OmniGlobalArgs<T> g, const OmniTab tab, const OmniPmDmaArgs<T> pm,
                         const StreamScreen<32> ss)
{
    if (omni_gate_skip(g)) return;
    __shared__ StreamEntry tab_lds[CHAIN ? 33 : 1];
    __shared__ __align__(16) uint32_t out_img[CHAIN ? 16 * KMAX : 4];
    extern __shared__ __align__(16) unsigned char nd_smem_dma[];
    T *img = reinterpret_cast<T *>(nd_smem_dma);
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t px0 = b * 64;
    const int64_t x0 = px0 + lane;
    const int k = g.k;
    const bool in = x0 < g.nx;
    const int64_t left = g.nx - px0;
    const int np = left > 64 ? 64 : (int)left;

    // ---- every transfer of the wave in flight ----
    auto stage = [&](const T *base, int vi) {
        const int wpp = k * pm.ids[vi];                     // elements per pixel in memory
        const int bytes = np * wpp * (int)sizeof(T);        // a multiple of 16 for whole spans of 64 pixels
        const unsigned char *src = reinterpret_cast<const unsigned char *>(base + px0 * wpp);
        unsigned char *dst = reinterpret_cast<unsigned char *>(img + pm.img_off[vi]);
        const int bytes16 = bytes & ~15;
        for (int c0 = 0; c0 < bytes16; c0 += 1024) {
            const int eb = c0 + lane * 16;
            if (eb < bytes16)
                __builtin_amdgcn_global_load_lds((glb_u8_t *)(src + eb), (lds_u8_t *)(dst + c0), 16, 0, kNtAux);
        }
        // (the last span of a variable whose series length is not a multiple of the vector: up to three words more)
        if (bytes16 < bytes && bytes16 + lane * 4 < bytes)
            __builtin_amdgcn_global_load_lds((glb_u8_t *)(src + bytes16 + lane * 4), (lds_u8_t *)(dst + bytes16), 4, 0, kNtAux);
    };
    if (!DIRECT) {
        stage(g.c11, 0);
        stage(g.c22, 3);
    }
    stage(g.c12r, 1);
    if (!pm.c12_joint) stage(g.c12i, 2);

    if (g.write_tab && b == 0) {
        for (int j = lane; j <= k; j += 64) g.tab_dev[j] = tab.e[j];
    }
    const int own = in ? lane : np - 1;
    // DIRECT: this lane's C11 / C22 pieces are requested behind the transfers and in front of the one
    // wait, so that both are in flight together
    constexpr int VE = 16 / (int)sizeof(T);
    typedef T tv __attribute__((ext_vector_type(VE)));
    tv qa[DIRECT ? KMAX / VE : 1], qd[DIRECT ? KMAX / VE : 1];
    if (DIRECT) {
        const T *p11 = g.c11 + (px0 + own) * (int64_t)k;
        const T *p22 = g.c22 + (px0 + own) * (int64_t)k;
#pragma unroll
        for (int u = 0; u < KMAX / VE; ++u) {
            if (u * VE < k) {
                // PLAIN loads, not non-temporal ones: a wave instruction takes 16 bytes of every lane's
                // series (pitch 4 k bytes), i.e. a part of each 128-byte line, and the other parts of
                // the line are asked for by the instructions that follow.  A non-temporal line is dropped
                // from the L2 after its first use and fetched again for the next piece: 8.19 GB of
                // traffic for 6.44 GB of data, 1.86 ms; plain loads: 6.44 GB, 1.49 ms
                // (tools/probe_fetch.hip, pitch96: 0.62 against 0.50 of FETCH_SIZE per byte, 758 against
                // 339 us; gpurun_out/r5_exp1 -> profiles/r05_fetch_calibration.json).
                qa[u] = *(reinterpret_cast<const tv *>(p11) + u);
                qd[u] = *(reinterpret_cast<const tv *>(p22) + u);
            }
        }
#pragma unroll
        for (int u = 0; u < KMAX / VE; ++u) asm volatile("" : "+v"(qa[u]), "+v"(qd[u]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- this lane's series out of the images (idle lanes copy the last pixel) ----
    T v[KMAX][4];
    if (CHAIN) {
        if (lane <= 32) tab_lds[lane] = ss.e[lane];
#pragma unroll
        for (int t = 0; t < KMAX; ++t) v[t][0] = v[t][1] = v[t][2] = v[t][3] = (T)1;   // dates behind k: masked
        __syncthreads();
    }
    if (DIRECT) {
#pragma unroll
        for (int u = 0; u < KMAX / VE; ++u) {
            if (u * VE < k) {
#pragma unroll
                for (int i = 0; i < VE; ++i) {
                    v[u * VE + i][0] = qa[u][i];
                    v[u * VE + i][3] = qd[u][i];
                }
            }
        }
    } else {
    pm_pick<T, KMAX, 0, false>(v, img + pm.img_off[0] + own * k * pm.ids[0], k, pm.ids[0]);
    pm_pick<T, KMAX, 3, false>(v, img + pm.img_off[3] + own * k * pm.ids[3], k, pm.ids[3]);
    }
    if (pm.c12_joint) {
        pm_pick<T, KMAX, 1, true>(v, img + pm.img_off[1] + own * k * 2, k, 2);
    } else {
        pm_pick<T, KMAX, 1, false>(v, img + pm.img_off[1] + own * k * pm.ids[1], k, pm.ids[1]);
        pm_pick<T, KMAX, 2, false>(v, img + pm.img_off[2] + own * k * pm.ids[2], k, pm.ids[2]);
    }

    bool flag;
    bool dense = false;
    if (CHAIN) {
        // ---- the search; a wave with few candidates lists them for pass B instead of using it ----
        unsigned mask;
        bool handoff, cand;
        int ks = g.k;
        asm volatile("" : "+s"(ks));
        dense_chain<T, KMAX, 32>(v, ks, in, ss, tab_lds, mask, handoff, cand);
        dense = true;                 // the search is done: its result serves every pixel of the wave
        flag = cand;
        if (dense) {
            if (handoff) mask = 0u;                           // pass B writes that pixel's changes
            uint8_t *wob = g.change + px0 * (int64_t)k;
            if (change_rows_wave_ok(wob, k, np)) {
                store_change_rows_wave(wob, out_img, k, mask, lane);
            } else if (in) {
                uint8_t *res = wob + (int64_t)lane * k;
                for (int t = 0; t < k; ++t) res[t] = (uint8_t)((mask >> t) & 1u);
            }
            flag = handoff;
        }
    } else {
    // ---- fold in time order ----
    Accum<T> A;
    A.reset();
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
        if (t < k) A.step(v[t][0], v[t][1], v[t][2], v[t][3]);

    if (STATS) {
        const T z = z_stat<T>(A, k, g.nlooks, g.e);
        double zd[1] = {(double)z}, P1[1], P2[1];
        chisq_pair<1>(zd, 4 * (k - 1), g.e.lgam, P1, P2);
        const T P = combine_P<T>(P1[0], P2[0], g.e.omega2);
        flag = in && ((double)P > g.alpha);
        if (in) {
            if (g.z_out) g.z_out[x0] = z;
            if (g.p_out) g.p_out[x0] = P;
        }
    } else {
        flag = in && (z_approx<T>(A, k, g.nlooks, g.e) >= g.e.zlo_a);
    }
    }

    // ---- list + dump ----
    const unsigned long long m = __ballot(flag);
    if (m != 0ull) {
        const unsigned shard = (unsigned)(b % kShards);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(g.flag_count + shard * kCounterStride, (unsigned)__popcll(m));
        base = __shfl(base, 0);
        if (flag) {
            const unsigned slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            g.flag_idx[(size_t)shard * g.seg + slot] = (uint32_t)x0;
            if (slot < g.dump_cap) {
                T *d = g.dump + ((int64_t)shard * g.dump_cap + slot) * (int64_t)(4 * k);
#pragma unroll
                for (int t = 0; t < KMAX; ++t) {
                    if (t < k) {
                        Pack<T, 4> q;
                        q.v[0] = v[t][0];
                        q.v[1] = v[t][1];
                        q.v[2] = v[t][2];
                        q.v[3] = v[t][3];
                        *reinterpret_cast<Pack<T, 4> *>(d + 4 * t) = q;
                    }
                }
            }
        }
    }
    // ---- zero-fill this wave's slice of the change map (np.zeros at nd/_change.pyx:275) ----
    if (!dense) zero_fill_span(g.change + px0 * (int64_t)k, np * k, lane);
}


// -----------------------------------------------------------------------------------------
// pass A for the reference's layout BEYOND the register-retaining sizes (25 .. 192 dates): the wave's span
// of every variable (PXW pixels x k dates, contiguous in memory) goes into wave-private LDS images exactly
// as in omnibus_c2_pm_dma_kernel -- every transfer in flight at once, no barrier -- and each lane folds
// its pixel's series out of the images in time order (16-byte LDS reads) instead of retaining it.  PXW =
// 64, 32 or 16 pixels per wave (the upper lanes idle), whatever keeps the images within 48 KB: three
// waves per CU and 144 KB of transfers in flight.  No dump: pass B reads a listed pixel's series from the
// variables themselves, where it is contiguous (nd/change.py:66-67 hands the native code this layout for
// any series length; up to round 4 such inputs went through the transpose kernels and the planar path:
// three trips over the stack).
// -----------------------------------------------------------------------------------------
// JOINT: C12 is one interleaved complex array (one image, re at the even places); otherwise four real
// arrays.  (Other mixes of date strides are left to the transpose route.)
template <typename T, int PXW, bool STATS, bool JOINT>
__global__ void __launch_bounds__(64) omnibus_c2_pm_long_kernel(const OmniGlobalArgs<T> g, const OmniTab tab,
                                                             const OmniPmDmaArgs<T> pm)
{
    constexpr int VE = 16 / (int)sizeof(T);
    extern __shared__ __align__(16) unsigned char nd_smem_dma[];
    T *img = reinterpret_cast<T *>(nd_smem_dma);
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t px0 = b * PXW;
    const int64_t x0 = px0 + lane;
    const int k = g.k;
    const bool in = lane < PXW && x0 < g.nx;
    const int64_t left = g.nx - px0;
    const int np = left > PXW ? PXW : (int)left;

    // ---- every transfer of the wave in flight ----
    auto stage = [&](const T *base, int vi) {
        const int wpp = k * pm.ids[vi];                     // elements per pixel in memory
        const int bytes = np * wpp * (int)sizeof(T);        // multiple of 16 (host checks k)
        const unsigned char *src = reinterpret_cast<const unsigned char *>(base + px0 * wpp);
        unsigned char *dst = reinterpret_cast<unsigned char *>(img + pm.img_off[vi]);
        for (int c0 = 0; c0 < bytes; c0 += 1024) {
            const int eb = c0 + lane * 16;
            if (eb < bytes)
                __builtin_amdgcn_global_load_lds((glb_u8_t *)(src + eb), (lds_u8_t *)(dst + c0), 16, 0, kNtAux);
        }
    };
    stage(g.c11, 0);
    stage(g.c12r, 1);
    if (!JOINT) stage(g.c12i, 2);
    stage(g.c22, 3);
    if (g.write_tab && b == 0) {
        for (int j = lane; j <= k; j += 64) g.tab_dev[j] = tab.e[j];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- fold this lane's series in time order (idle lanes fold the last pixel of the span) ----
    const int own = (lane < np) ? lane : np - 1;
    typedef Pack<T, VE> PV;
    const PV *i11 = reinterpret_cast<const PV *>(img + pm.img_off[0] + own * k);
    const PV *i22 = reinterpret_cast<const PV *>(img + pm.img_off[3] + own * k);
    const PV *i12 = reinterpret_cast<const PV *>(img + pm.img_off[1] + own * k * (JOINT ? 2 : 1));
    const PV *i12i = reinterpret_cast<const PV *>(img + pm.img_off[2] + own * k);        // (not JOINT)
    Accum<T> A;
    A.reset();
    const int nv = k / VE;
    for (int u = 0; u < nv; ++u) {
        const PV a = i11[u], d = i22[u];
        PV br, bi;
        if (JOINT) {
            const PV q0 = i12[2 * u], q1 = i12[2 * u + 1];      // (re, im) pairs of VE dates
#pragma unroll
            for (int j = 0; j < VE; ++j) {
                br.v[j] = (j < VE / 2 ? q0 : q1).v[2 * (j % (VE / 2))];
                bi.v[j] = (j < VE / 2 ? q0 : q1).v[2 * (j % (VE / 2)) + 1];
            }
        } else {
            br = i12[u];
            bi = i12i[u];
        }
#pragma unroll
        for (int j = 0; j < VE; ++j) A.step(a.v[j], br.v[j], bi.v[j], d.v[j]);
    }

    bool flag;
    if (STATS) {
        const T z = z_stat<T>(A, k, g.nlooks, g.e);
        double zd[1] = {(double)z}, P1[1], P2[1];
        chisq_pair<1>(zd, 4 * (k - 1), g.e.lgam, P1, P2);
        const T P = combine_P<T>(P1[0], P2[0], g.e.omega2);
        flag = in && ((double)P > g.alpha);
        if (in) {
            if (g.z_out) g.z_out[x0] = z;
            if (g.p_out) g.p_out[x0] = P;
        }
    } else {
        flag = in && (z_approx<T>(A, k, g.nlooks, g.e) >= g.e.zlo_a);
    }

    // ---- list (no dump: the series of a listed pixel is contiguous in the variables themselves) ----
    const unsigned long long m = __ballot(flag);
    if (m != 0ull) {
        const unsigned shard = (unsigned)(b % kShards);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(g.flag_count + shard * kCounterStride, (unsigned)__popcll(m));
        base = __shfl(base, 0);
        if (flag) g.flag_idx[(size_t)shard * g.seg + base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)x0;
    }
    // ---- zero-fill this wave's slice of the change map (np.zeros at nd/_change.pyx:275) ----
    zero_fill_span(g.change + px0 * (int64_t)k, np * k, lane);
}

// =========================================================================================
// dense waves: the change-point search from registers
// =========================================================================================
// Where a large share of the pixels is listed (low thresholds: the reference's default alpha = 0.01
// lists 99 % of them) pass B's one-lane-per-candidate LDS image is the wrong shape: it pays for a
// compaction nobody needs and runs at 1.5 waves per SIMD.  A dense wave is searched in place
// instead: one thread per pixel reloads its series (coalesced, all loads in flight, like pass A)
// and runs nd/_change.pyx:235-257 on registers.  Register arrays can only be indexed statically, so
// every segment is a pass over all k dates with the dates before the segment start predicated off
// -- the same additions in the same order as the sweep of pass B, hence the same decisions.
// Only the f32-log screen lives here.  A lane whose statistic falls inside the screen's band
// (~1e-5 of the tests) stops and appends its pixel to pass B's list, which then redoes that pixel
// from its first date with the exact evaluation (rewriting the same change bytes): the exact code
// would otherwise be instantiated 24 times and cost every wave 80 registers.
template <typename T>
struct OmniDenseArgs {
    const T *c11, *c12r, *c12i, *c22;
    int64_t nx, npix;         // pixels per row of the raster, pixels in all
    int64_t sy, sx, st;
    int k, flat;              // flat: rows are contiguous, a wave may run across a row end
    double nlooks, alpha;
    uint8_t *change;
    uint32_t *flag_count;     // word 0: pixel list (appended to here), word 1: dense list
    const uint32_t *dense_idx;
    uint32_t segd;
    const OmniTabEntry *tab;
    // a pixel whose test falls inside the screen's band is handed to pass B through the ordinary
    // pixel list (with its series in the dump, as pass A would have written it)
    uint32_t *flag_idx;
    uint32_t seg;
    T *dump;
    uint32_t dump_cap;
};

// ---- the search itself, on a series held in registers ------------------------------------
// (log2_parts, the screen registers and dense_x live in omnibus_common.hpp: the full-pol kernels use them too)


// One lane's whole row of the change map from its mask: every byte is written.
__device__ __forceinline__ void store_change_row(uint8_t *res, const int k, const unsigned long long mask)
{
    if ((k & 3) == 0 && ((uintptr_t)res & 3) == 0) {   // rows start on 4-byte boundaries: whole words
        uint32_t *w = reinterpret_cast<uint32_t *>(res);
        for (int q = 0; q < (k >> 2); ++q) {
            const unsigned nib = (unsigned)(mask >> (4 * q)) & 0xFu;
            w[q] = (nib * 0x00204081u) & 0x01010101u;       // bit i -> byte i
        }
    } else {
        for (int t = 0; t < k; ++t) res[t] = (uint8_t)((mask >> t) & 1ull);
    }
}

template <typename T, int KMAX>
__global__ void __launch_bounds__(64, 2) omnibus_c2_dense_kernel(const OmniDenseArgs<T> s,
                                                              const StreamScreen<32> ss)
{
    __shared__ StreamEntry tab_lds[33];
    const int lane = threadIdx.x;
    const unsigned shard = blockIdx.x % kShards;
    const unsigned lblock = blockIdx.x / kShards, nlblock = gridDim.x / kShards;
    const uint32_t n = s.flag_count[shard * kCounterStride + 1];
    if (lblock >= n) return;                  // the usual case in the sparse regime: nothing listed
    const uint32_t *list = s.dense_idx + (size_t)shard * s.segd;
    if (lane <= 32) tab_lds[lane] = ss.e[lane];
    __syncthreads();

    for (uint32_t w = lblock; w < n; w += nlblock) {
        // (opaque copy of k per wave of pixels: everything derived from k alone -- per-date
        // predicates, (float)j, bound coefficients -- would otherwise be hoisted out of this loop
        // and held in ~100 vector and scalar registers for the whole kernel)
        int k = s.k;
        asm volatile("" : "+s"(k));
        const int64_t pix0 = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)list[w]);   // wave-uniform
        const int64_t pix = pix0 + lane;
        const int64_t row0 = pix0 / s.nx;
        const int64_t col0 = pix0 - row0 * s.nx;
        // a wave of pass A never leaves its row unless the rows are contiguous (flat: sx = 1,
        // sy = nx); idle lanes re-read the last pixel of the row / raster
        int64_t ub;           // uniform element offset of lane 0's pixel
        unsigned delta;       // this lane's element distance from it (>= 0)
        bool active;
        if (s.flat) {
            active = pix < s.npix;
            ub = pix0;
            delta = (unsigned)((active ? pix : s.npix - 1) - pix0);
        } else {
            active = col0 + lane < s.nx;
            ub = row0 * s.sy + col0 * s.sx;
            delta = (unsigned)(((active ? col0 + lane : s.nx - 1) - col0) * s.sx);
        }
        // every load of the series in flight at once: no per-date branch (dates beyond k re-read
        // the last one; the search never looks at them), scalar base + 32-bit lane offset
        T v[KMAX][4];
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            const int64_t ot = ub + (int64_t)(t < k ? t : k - 1) * s.st;
            v[t][0] = (s.c11 + ot)[delta];
            v[t][1] = (s.c12r + ot)[delta];
            v[t][2] = (s.c12i + ot)[delta];
            v[t][3] = (s.c22 + ot)[delta];
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the loads together, ahead of every use
        // (round 3: the two linear passes of dense_chain; the triangle of dense_search took 1.18 ms
        // here at alpha = 0.8, where a fifth of the pixels are flagged and most waves are dense)
        unsigned mask;
        bool handoff, cand;
        dense_chain<T, KMAX, 32>(v, k, active, ss, tab_lds, mask, handoff, cand);
        if (active && !handoff && mask != 0u) store_change_row(s.change + pix * (int64_t)k, k, mask);
        if (__any(handoff)) {
            const unsigned long long m = __ballot(handoff);
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(s.flag_count + shard * kCounterStride, (unsigned)__popcll(m));
            base = __shfl(base, 0);
            if (handoff) {
                const unsigned slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                s.flag_idx[(size_t)shard * s.seg + slot] = (uint32_t)pix;
                if (slot < s.dump_cap) {
                    T *d = s.dump + ((int64_t)shard * s.dump_cap + slot) * (int64_t)(4 * k);
#pragma unroll
                    for (int t = 0; t < KMAX; ++t)
                        if (t < k) {
                            Pack<T, 4> q;
                            q.v[0] = v[t][0];
                            q.v[1] = v[t][1];
                            q.v[2] = v[t][2];
                            q.v[3] = v[t][3];
                            *reinterpret_cast<Pack<T, 4> *>(d + 4 * t) = q;
                        }
                }
            }
        }
    }
}

// -----------------------------------------------------------------------------------------
// pass A with the search fused in, two linear passes over the retained series (dense_chain): the
// form for thresholds at which marginal tests beyond three dates are common (everything between
// the reference's default 0.01 and the sparse regime).  Same loads as omnibus_c2_retain_kernel.
// -----------------------------------------------------------------------------------------

// STATS: the z / P rasters of the whole-series test as well (nd/_change.pyx:46-77) -- the reference's
// forward fold of the retained series and the exact chi-square pair, as omnibus_c2_retain_kernel<..., true>
// evaluates them, so that a call with rasters reads the planes once (up to round 3: a pass A of its own
// for the rasters in front of the search).
template <typename T, int KMAX, bool EXACT, bool STATS = false>
__global__ void __launch_bounds__(kRetainThreads, chain_waves(KMAX, sizeof(T)))
omnibus_c2_chain_kernel(const OmniGlobalArgs<T> g, const OmniTab tab, const StreamScreen<chain_nj(KMAX)> ss)
{
    constexpr int NJ = chain_nj(KMAX);
    typedef typename std::conditional<(KMAX > 32), unsigned long long, unsigned>::type MT;
    if (omni_gate_skip(g)) return;
    __shared__ StreamEntry tab_lds[NJ + 1];   // the marginal tests' constants: a lane reads those of its own j
    __shared__ __align__(16) uint32_t out_img[(kRetainThreads / 64) * 16 * KMAX];   // store_change_rows_wave
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int64_t b = blockIdx.x;
    const int64_t row = b / g.blocks_per_row;
    const int64_t bx = b - row * g.blocks_per_row;
    const int64_t bpx0 = bx * (int64_t)kRetainThreads;
    const int64_t x0 = bpx0 + tid;
    const int k = EXACT ? KMAX : g.k;
    const bool in = x0 < g.nx;

    // ---- issue every load of the series (as in omnibus_c2_retain_kernel) ----
    T v[KMAX][4];
    if (EXACT) {
        const int64_t ub = row * g.sy + bpx0;
        const unsigned lx = in ? (unsigned)tid : (unsigned)(g.nx - 1 - bpx0);   // idle lanes re-read the last pixel
        const unsigned voff = lx * (unsigned)sizeof(T);
        const auto r11 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c11 + ub), 0, 0x7fffffff, 0x00020000);
        const auto r12r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c12r + ub), 0, 0x7fffffff, 0x00020000);
        const auto r12i = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c12i + ub), 0, 0x7fffffff, 0x00020000);
        const auto r22 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c22 + ub), 0, 0x7fffffff, 0x00020000);
        const unsigned sstep = (unsigned)g.st * (unsigned)sizeof(T);   // host guarantees k * st * sizeof(T) < 2^31
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            const unsigned soff = (unsigned)t * sstep;
            v[t][0] = buffer_load<T>(r11, voff, soff);
            v[t][1] = buffer_load<T>(r12r, voff, soff);
            v[t][2] = buffer_load<T>(r12i, voff, soff);
            v[t][3] = buffer_load<T>(r22, voff, soff);
        }
    } else {
        const int64_t xc = in ? x0 : g.nx - 1;
        const int64_t off0 = row * g.sy + xc * g.sx;
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            const int64_t off = off0 + (int64_t)(t < k ? t : k - 1) * g.st;    // no per-date branch
            v[t][0] = __builtin_nontemporal_load(g.c11 + off);
            v[t][1] = __builtin_nontemporal_load(g.c12r + off);
            v[t][2] = __builtin_nontemporal_load(g.c12i + off);
            v[t][3] = __builtin_nontemporal_load(g.c22 + off);
        }
    }
    if (tid <= NJ) tab_lds[tid] = ss.e[tid];
    if (g.write_tab && b == 0) {
        for (int j = tid; j <= k; j += kRetainThreads) g.tab_dev[j] = tab.e[j];
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);      // keep the loads together, ahead of every use

    const unsigned shard = (unsigned)(b % kShards);
    const int64_t wpx0 = bpx0 + (tid & ~63);                  // first pixel of this wave in its row
    const int64_t wleft = g.nx - wpx0;
    const int wnp = wleft > 64 ? 64 : (wleft > 0 ? (int)wleft : 0);
    uint8_t *wob = g.change + (row * g.nx + wpx0) * (int64_t)k;

    // STATS: the reference's forward fold of the whole series IN FRONT of the search -- its result is six
    // registers carried through the search (four sums, the double product), while the logarithms and the
    // chi-square series behind the search then run with the series itself dead.  (Round 4 had the fold behind
    // the search as well: the series stayed alive into the chi-square code, 140 bytes per lane spilled --
    // 41 scratch operations in the 24-date float32 instantiation.)
    // The fold's result waits in LDS (this thread's own five words, no barrier): carried in registers it cost
    // the search its last free ones (the 24-date float32 form is held to three waves per SIMD: 88 scratch
    // operations).
    __shared__ T stat_s[STATS ? 4 : 1][kRetainThreads];
    __shared__ double stat_p[STATS ? kRetainThreads : 1];
    if (STATS) {
        Accum<T> Aw;
        Aw.reset();
#pragma unroll
        for (int t = 0; t < KMAX; ++t)
            if (EXACT || t < k) {
                // (opaque to the optimiser: it would otherwise share these determinants with the search's
                //  first pass and keep all of them alive in between)
                T a0 = v[t][0];
                if (sizeof(T) == 4) asm volatile("" : "+v"(a0));      // (float64, 24 dates: 288 instead of 108 bytes spilled with it)
                Aw.step(a0, v[t][1], v[t][2], v[t][3]);
            }
        stat_s[0][tid] = Aw.s11;
        stat_s[1][tid] = Aw.s12r;
        stat_s[2][tid] = Aw.s12i;
        stat_s[3][tid] = Aw.s22;
        stat_p[tid] = Aw.prod;
    }
    // ---- the search; a wave with few candidates lists them for pass B instead of using it ----
    MT mask;
    bool handoff, cand;
    // (k as a run-time value even when it is known to equal KMAX: with every guard folded away the
    // two passes become one straight block for the scheduler)
    int ks = g.k;
    asm volatile("" : "+s"(ks));
    dense_chain<T, KMAX, NJ>(v, ks, in, ss, tab_lds, mask, handoff, cand);
    // The search of every pixel is done whatever the wave's share of candidates: its result is
    // used for all of them (only what the screen could not decide goes to pass B; `dense_min`
    // belongs to the forms that search on demand).
    bool listed = cand;
    const bool dense = true;
    if (dense) {
        if (handoff) mask = 0;                                // pass B writes that pixel's changes
        if (change_rows_wave_ok(wob, k, wnp)) {
            store_change_rows_wave(wob, out_img + (tid >> 6) * (16 * KMAX), k, mask, lane);
        } else if (in) {
            uint8_t *res = wob + (int64_t)lane * k;
            if ((k & 3) == 0 && ((uintptr_t)res & 3) == 0) {
                uint32_t *w = reinterpret_cast<uint32_t *>(res);
#pragma unroll
                for (int q = 0; q < KMAX / 4; ++q)
                    if (q < (k >> 2)) w[q] = (mask_nibble(mask, q) * 0x00204081u) & 0x01010101u;
            } else {
                for (int t = 0; t < k; ++t) res[t] = (uint8_t)((mask >> t) & (MT)1);
            }
        }
        listed = handoff;
    }
    if (__any(listed)) {
        const unsigned long long lm_ = __ballot(listed);
        unsigned base = 0;
        if (lane == 0)
            base = atomicAdd(g.flag_count + shard * kCounterStride, (unsigned)__popcll(lm_));
        base = __shfl(base, 0);
        if (listed) {
            const unsigned slot = base + (unsigned)__popcll(lm_ & ((1ull << lane) - 1ull));
            g.flag_idx[(size_t)shard * g.seg + slot] = (uint32_t)(row * g.nx + x0);
            if (slot < g.dump_cap) {
                T *d = g.dump + ((int64_t)shard * g.dump_cap + slot) * (int64_t)(4 * k);
#pragma unroll
                for (int t = 0; t < KMAX; ++t) {
                    if (EXACT || t < k) {
                        Pack<T, 4> q;
                        q.v[0] = v[t][0];
                        q.v[1] = v[t][1];
                        q.v[2] = v[t][2];
                        q.v[3] = v[t][3];
                        *reinterpret_cast<Pack<T, 4> *>(d + 4 * t) = q;
                    }
                }
            }
        }
    }
    // ---- a sparse wave zero-fills its own slice of the change map (np.zeros, nd/_change.pyx:275)
    if (!dense && wnp > 0) zero_fill_span(wob, wnp * k, lane);
    // (behind the search, where only the series itself is still alive: in front of it the fold's and the
    // chi-square series' registers pushed the 24-date form over the cap of three waves per SIMD)
    __builtin_amdgcn_sched_barrier(0);
    if (STATS) {
        Accum<T> Aw;
        Aw.s11 = stat_s[0][tid];
        Aw.s12r = stat_s[1][tid];
        Aw.s12i = stat_s[2][tid];
        Aw.s22 = stat_s[3][tid];
        Aw.prod = stat_p[tid];
        const T z = z_stat<T>(Aw, k, g.nlooks, g.e);
        double zd[1] = {(double)z}, P1[1], P2[1];
        chisq_pair<1>(zd, 4 * (k - 1), g.e.lgam, P1, P2);
        const T P = combine_P<T>(P1[0], P2[0], g.e.omega2);
        if (in) {
            const int64_t pix = row * g.nx + x0;
            if (g.z_out) g.z_out[pix] = z;
            if (g.p_out) g.p_out[pix] = P;
        }
    }
}

// -----------------------------------------------------------------------------------------
// pass A with the search fused in, streaming form (low thresholds).  The series is NOT kept in
// registers: the dates are consumed as they arrive, last date first, PF dates in flight.
//
//   phase 1 (per date t, branch-free): logarithm of the date's determinant; suffix sums over
//           ts[t:] in double -> the global test G(t) as in dense_search; the 2- and 3-date sums
//           a_t + a_t+1 (+ a_t+2) in the reference's `floating` and order from a rolling window of
//           three dates -> the marginal tests M2(t), M3(t) with the tight band.  Each test leaves
//           two bits per lane (fires / undecided) at position t of six 32-bit masks.
//   walk    (per segment start l, a handful of bit operations): G(l) undecided -> pass B; does not
//           fire -> finished; else the first firing marginal: M2(l), M3(l), and only if neither
//           fires (about 1e-3 of the rows at alpha = 0.01) the deeper marginals j = 4, 5, ... with
//           the dates read again from memory (L2-resident) in a rolled loop.
// About 100 registers -> 4 to 5 waves per SIMD instead of 2, and any k <= 32.
// Same decisions as dense_search (same sums, same bounds), hence the same map.
// -----------------------------------------------------------------------------------------
template <typename T>
struct DateVal {
    T a, b, c, d;
};

// MODE 0: plain pointers (any strides); 1: x-contiguous planes through buffer descriptors + 32-bit
// lane offset + scalar date offset; 2: the wave's LDS images of pixel-major variables (LDS-DMA);
// 3: pixel-major variables read straight from memory, each lane its own series in 16-byte pieces
// (64 different sectors per load instruction, every sector touched by two to four consecutive
// groups of dates: the re-reads are L2 hits) -- no LDS, the occupancy of the planar form
template <typename T, int MODE>
struct PlaneReader {
    __amdgpu_buffer_rsrc_t r11, r12r, r12i, r22;
    unsigned voff, sstep;
    const T *p11, *p12r, *p12i, *p22;      // MODE 2 / 3: this lane's series inside each image / variable
    int64_t st;
    int i11, i12, i22, joint;              // MODE 2 / 3: date strides
    __device__ __forceinline__ DateVal<T> load(const int t) const
    {
        DateVal<T> q;
        if (MODE == 1) {
            const unsigned soff = (unsigned)t * sstep;
            q.a = buffer_load<T>(r11, voff, soff);
            q.b = buffer_load<T>(r12r, voff, soff);
            q.c = buffer_load<T>(r12i, voff, soff);
            q.d = buffer_load<T>(r22, voff, soff);
        } else if (MODE >= 2) {
            q.a = p11[t * i11];
            q.d = p22[t * i22];
            if (joint) {
                const Pack<T, 2> bc = *reinterpret_cast<const Pack<T, 2> *>(p12r + 2 * t);
                q.b = bc.v[0];
                q.c = bc.v[1];
            } else {
                q.b = p12r[t * i12];
                q.c = p12i[t * i12];
            }
        } else {
            const int64_t o = (int64_t)t * st;
            q.a = __builtin_nontemporal_load(p11 + o);
            q.b = __builtin_nontemporal_load(p12r + o);
            q.c = __builtin_nontemporal_load(p12i + o);
            q.d = __builtin_nontemporal_load(p22 + o);
        }
        return q;
    }
    // the same with a date index of the lane's own (the deep marginal searches)
    __device__ __forceinline__ DateVal<T> load_lane(const int t) const
    {
        if (MODE == 1) {
            DateVal<T> q;
            const unsigned vo = voff + (unsigned)t * sstep;
            q.a = buffer_load<T>(r11, vo, 0u);
            q.b = buffer_load<T>(r12r, vo, 0u);
            q.c = buffer_load<T>(r12i, vo, 0u);
            q.d = buffer_load<T>(r22, vo, 0u);
            return q;
        }
        return load(t);
    }
    // MODE 2: the VE = 16 / sizeof(T) dates t0 .. t0 + VE - 1 (t0 a multiple of VE) with 16-byte LDS
    // reads where the variable's dates are adjacent (a lane's series is k * ids elements from the
    // next lane's: single-element reads of one date collide 8-way on the 32 banks, 16-byte reads 2-way)
    template <int VE>
    __device__ __forceinline__ void load_group(const int t0, DateVal<T> (&q)[VE]) const
    {
        auto var = [&](const T *p, int ids, auto put) {
            if (ids == 1) {
                const Pack<T, VE> w = *reinterpret_cast<const Pack<T, VE> *>(p + t0);
#pragma unroll
                for (int i = 0; i < VE; ++i) put(i, w.v[i]);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const Pack<T, VE> w = *reinterpret_cast<const Pack<T, VE> *>(p + 2 * t0 + h * VE);
#pragma unroll
                    for (int i = 0; i < VE; i += 2) put(h * (VE / 2) + i / 2, w.v[i]);
                }
            }
        };
        var(p11, i11, [&](int i, T x) { q[i].a = x; });
        var(p22, i22, [&](int i, T x) { q[i].d = x; });
        if (joint) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const Pack<T, VE> w = *reinterpret_cast<const Pack<T, VE> *>(p12r + 2 * t0 + h * VE);
#pragma unroll
                for (int i = 0; i < VE; i += 2) {
                    q[h * (VE / 2) + i / 2].b = w.v[i];
                    q[h * (VE / 2) + i / 2].c = w.v[i + 1];
                }
            }
        } else {
            var(p12r, i12, [&](int i, T x) { q[i].b = x; });
            var(p12i, i12, [&](int i, T x) { q[i].c = x; });
        }
    }
};

// MW: width of the six test masks and of the change mask -- 0: 32 bits (k <= 32), 1: 64 bits
// (k <= 64), 2: two 64-bit words (k <= 128; the screen's entries 65 .. 128 then sit in a second
// set of registers, the sum of the mantissa logs is 64 bits wide)
// (MW = 3, round 6: 129 .. 192 dates, the two-pass chain search only)
constexpr int kStreamChainMax = 192;
constexpr int stream_nj(const int MW) { return MW == 0 ? 32 : (MW == 1 ? 64 : (MW == 2 ? kDenseMax : kStreamChainMax)); }

template <typename T, int PF, int MODE, int MW = 0>
__global__ void __launch_bounds__(MODE == 2 ? 64 : kRetainThreads)
omnibus_c2_stream_kernel(const OmniGlobalArgs<T> g, const OmniTab tab, const OmniPmDmaArgs<T> pm,
                         const StreamScreen<stream_nj(MW)> ss)
{
    if (omni_gate_skip(g)) return;
    constexpr int kThreads = MODE == 2 ? 64 : kRetainThreads;
    constexpr int kMaxDates = stream_nj(MW);
    __shared__ __align__(16) uint32_t out_img[(kThreads / 64) * 16 * kMaxDates];   // 64 rows of the map per wave
    extern __shared__ __align__(16) unsigned char nd_smem_stream[];    // MODE 2: the wave's images
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int64_t b = blockIdx.x;
    const int64_t row = MODE >= 2 ? 0 : b / g.blocks_per_row;           // pixel-major: one flat row
    const int64_t bx = MODE >= 2 ? b : b - row * g.blocks_per_row;
    const int64_t bpx0 = bx * (int64_t)kThreads;
    const int64_t x0 = bpx0 + tid;
    const int k = g.k;
    const bool in = x0 < g.nx;

    PlaneReader<T, MODE> rd;
    if (MODE == 2) {
        // the wave's span of every variable into its LDS image (as in omnibus_c2_pm_dma_kernel)
        T *img = reinterpret_cast<T *>(nd_smem_stream);
        const int64_t left = g.nx - bpx0;
        const int np = left > 64 ? 64 : (int)left;
        auto stage = [&](const T *base, int vi) {
            const int wpp = k * pm.ids[vi];
            const int bytes = np * wpp * (int)sizeof(T);
            const unsigned char *src = reinterpret_cast<const unsigned char *>(base + bpx0 * wpp);
            unsigned char *dst = reinterpret_cast<unsigned char *>(img + pm.img_off[vi]);
            for (int c0 = 0; c0 < bytes; c0 += 1024) {
                const int eb = c0 + lane * 16;
                if (eb < bytes)
                    __builtin_amdgcn_global_load_lds((glb_u8_t *)(src + eb), (lds_u8_t *)(dst + c0), 16, 0, kNtAux);
            }
        };
        stage(g.c11, 0);
        stage(g.c22, 3);
        stage(g.c12r, 1);
        if (!pm.c12_joint) stage(g.c12i, 2);
        const int own = in ? lane : np - 1;
        rd.i11 = pm.ids[0];
        rd.i12 = pm.ids[1];
        rd.i22 = pm.ids[3];
        rd.joint = pm.c12_joint;
        rd.p11 = img + pm.img_off[0] + own * k * pm.ids[0];
        rd.p22 = img + pm.img_off[3] + own * k * pm.ids[3];
        rd.p12r = img + pm.img_off[1] + own * k * pm.ids[1];
        rd.p12i = pm.c12_joint ? rd.p12r + 1 : img + pm.img_off[2] + own * k * pm.ids[2];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE >= 3) {
        const int64_t xc = in ? x0 : g.nx - 1;                  // idle lanes re-read the last pixel
        rd.i11 = pm.ids[0];
        rd.i12 = pm.ids[1];
        rd.i22 = pm.ids[3];
        rd.joint = pm.c12_joint;
        rd.p11 = g.c11 + xc * k * pm.ids[0];
        rd.p22 = g.c22 + xc * k * pm.ids[3];
        rd.p12r = g.c12r + xc * k * pm.ids[1];
        rd.p12i = pm.c12_joint ? rd.p12r + 1 : g.c12i + xc * k * pm.ids[2];
    } else {
        const int64_t xc = in ? x0 : g.nx - 1;                  // idle lanes re-read the last pixel
        if (MODE == 1) {
            const int64_t ub = row * g.sy + bpx0;
            rd.voff = (unsigned)(xc - bpx0) * (unsigned)sizeof(T);
            rd.r11 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c11 + ub), 0, 0x7fffffff, 0x00020000);
            rd.r12r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c12r + ub), 0, 0x7fffffff, 0x00020000);
            rd.r12i = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c12i + ub), 0, 0x7fffffff, 0x00020000);
            rd.r22 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c22 + ub), 0, 0x7fffffff, 0x00020000);
            rd.sstep = (unsigned)g.st * (unsigned)sizeof(T);   // host guarantees k * st * sizeof(T) < 2^31
        } else {
            const int64_t off0 = row * g.sy + xc * g.sx;
            rd.p11 = g.c11 + off0;
            rd.p12r = g.c12r + off0;
            rd.p12i = g.c12i + off0;
            rd.p22 = g.c22 + off0;
            rd.st = g.st;
        }
    }
    // ---- first dates in flight (last date first) ----
    // NS = PF + 2 slots: slot s of a group of NS dates holds date tb - s, and the two dates behind
    // the one being worked on (the window of the 2- and 3-date tests) are read from their slots, so
    // no value ever moves from one register to another.  A slot is free again two steps after its
    // own date: step u re-loads slot u - 2, PF dates ahead.
    constexpr int NS = PF + 2;
    DateVal<T> ring[NS];
    if (MODE < 2) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int t = k - 1 - u;
            if (u < PF) {
                ring[u] = rd.load(t > 0 ? t : 0);
            } else {                                   // "dates" behind the series: the unit matrix
                ring[u].a = ring[u].d = (T)1;
                ring[u].b = ring[u].c = (T)0;
            }
        }
    }
    if (g.write_tab && b == 0) {
        for (int j = tid; j <= k; j += kThreads) g.tab_dev[j] = tab.e[j];
    }

    // ---- phase 1 ----
    typedef typename std::conditional<MW == 2, Bits128,
                                      typename std::conditional<MW == 1, unsigned long long, unsigned>::type>::type MT;
    typedef typename std::conditional<MW == 2, long long, int>::type LmT;
    // per test two bits: F = fires for certain, C = cannot fire; neither = undecided
    MT gF = mask_zero<MT>(), gC = mask_zero<MT>(), m2F = mask_zero<MT>(), m2C = mask_zero<MT>(),
       m3F = mask_zero<MT>(), m3C = mask_zero<MT>();
    bool bad = false;
    bool dead = false;       // a date whose determinant is NaN or exactly 0: see below
    double S11 = 0.0, S12r = 0.0, S12i = 0.0, S22 = 0.0;
    // The product of the determinants of ts[t:] in double, as the reference forms it (its own runs
    // forward from each segment start; the two differ by rounding errors of 1e-16 per factor).  One
    // logarithm of its mantissa per date serves the global test; emin / emax are the extreme binary
    // exponents it passes through (1 = the empty product behind the last date).
    double PP = 1.0;
    int emin = 1, emax = 1;
    T det1 = (T)1, prod12 = (T)1;              // det(t + 1);  det(t + 1) * det(t + 2)
    const T dlo = (T)ss.dlo, dhi = (T)ss.dhi;
    const T ca2 = (T)ss.ca.x, cb2 = (T)ss.cb.x, ca3 = (T)ss.ca.y, cb3 = (T)ss.cb.y;

    // Every date pushes one bit into each of the six masks (m = 2 m + bit): dates arrive last to
    // first, so the bit of date t ends at position t.  Tests that do not exist (the global test of the
    // last date alone, marginal tests reaching behind the series) are cleared behind the loop.
    // q: date t;  d1, d2: dates t + 1, t + 2 (the unit matrix behind the series)
    // c: the constants of the global test over k - t dates (ss.e[k - t])
    auto process = [&](const DateVal<T> &q, const DateVal<T> &d1, const DateVal<T> &d2, const int t,
                       const StreamEntry &c) {
        const T det = (q.a * q.d) - ((q.b * q.b) + (q.c * q.c));
        // (the range of the determinant is checked together with those of the 2- and 3-date sums)
        bad = bad | !(q.a > (T)0);
        dead = dead | !((det > (T)0) | (det < (T)0));
        const T ds = det;
        PP = PP * (double)ds;
        S11 += (double)q.a;
        S12r += (double)q.b;
        S12i += (double)q.c;
        S22 += (double)q.d;
        {                                                   // global test of ts[t:], j = k - t
            const int jj = k - t;
            const double pp = S11 * S22;
            const double dets = pp - ((S12r * S12r) + (S12i * S12i));
            const float df = (float)dets;
            bool okd;
            int es, eP;
            float ms, mP;
            // (where a test is not `okd`, or a date not `ok`, the pixel goes to the exact pass whatever
            // the bits say: nothing below needs a stand-in value)
            if (sizeof(T) == 4) {
                // sums of float32 data: their determinant in float32 (relative rounding 6e-8, i.e.
                // jj * 8.6e-8 in x, inside the screen's margin) -- single-precision frexp and compares
                // (an infinite df makes rel NaN or 0 * inf: caught by the test of rel below)
                okd = df > 7.888609052210118e-31f;
                log2_parts(df, es, ms);
            } else {
                okd = (dets > 0.0) & (dets < (double)INFINITY);
                log2_parts(dets, es, ms);
            }
            log2_parts(PP, eP, mP);
            emin = eP < emin ? eP : emin;
            emax = eP > emax ? eP : emax;
            const int E = (eP - c.re) - __mul24(jj, es);
            const float x = (float)E + __builtin_fmaf(-c.jf, ms, mP - c.rf);
            const float qq = (float)pp * __builtin_amdgcn_rcpf(df);
            const float rel = c.cj * qq;                    // 1.46 * 5 n u * s11 s22 / det
            const float m2 = c.mj * rel;
            bad = bad | !(okd & (rel < 0.01f));             // a test the screen cannot take: exact pass
            mask_push(gF, x + m2 < c.a);
            mask_push(gC, x - m2 > c.b);
        }
        if constexpr (sizeof(T) == 4) {                     // marginal tests over 2 and 3 dates
            // the reference's sums, in its type and order: (0 + a_t) + a_t+1 (+ a_t+2); decided from
            // products of determinants against powers of the sum's determinant (StreamScreen).
            // float32: the two tests side by side in the halves of packed instructions (the sums
            // are written into the halves by scalar additions: no moves to assemble the pairs)
            f2_t p11, p12r, p12i, p22;                      // .x: over 2 dates, .y: over 3
            p11.x = q.a + d1.a;
            p12r.x = q.b + d1.b;
            p12i.x = q.c + d1.c;
            p22.x = q.d + d1.d;
            p11.y = p11.x + d2.a;
            p12r.y = p12r.x + d2.b;
            p12i.y = p12i.x + d2.c;
            p22.y = p22.x + d2.d;
            const f2_t dp = (p11 * p22) - ((p12r * p12r) + (p12i * p12i));
            const f2_t sq = dp * dp;
            f2_t pw;
            pw.x = sq.x;
            pw.y = sq.y * dp.y;
            const f2_t ta = ss.ca * pw, tb = ss.cb * pw;
            const T prod2 = ds * det1, prod3 = ds * prod12;
            // (bitwise: a short-circuit chain of four compares becomes exec-masked branches)
            const bool above = fminf(fminf(det, dp.x), dp.y) > dlo, below = fmaxf(fmaxf(det, dp.x), dp.y) < dhi;
            bad = bad | !(above & below);
            mask_push(m2F, prod2 < ta.x);
            mask_push(m2C, prod2 > tb.x);
            mask_push(m3F, prod3 < ta.y);
            mask_push(m3C, prod3 > tb.y);
            prod12 = prod2;
            det1 = ds;
        } else {
            T s11 = q.a + d1.a, s12r = q.b + d1.b, s12i = q.c + d1.c, s22 = q.d + d1.d;
            const T prod2 = ds * det1;
            {
                const T dets = (s11 * s22) - ((s12r * s12r) + (s12i * s12i));
                bad = bad | !((dets > dlo) & (dets < dhi) & (det > dlo) & (det < dhi));
                const T r = dets * dets;
                mask_push(m2F, prod2 < ca2 * r);
                mask_push(m2C, prod2 > cb2 * r);
            }
            {
                s11 = s11 + d2.a;
                s12r = s12r + d2.b;
                s12i = s12i + d2.c;
                s22 = s22 + d2.d;
                const T dets = (s11 * s22) - ((s12r * s12r) + (s12i * s12i));
                bad = bad | !((dets > dlo) & (dets < dhi));
                const T r = (dets * dets) * dets;
                const T prod3 = ds * prod12;
                mask_push(m3F, prod3 < ca3 * r);
                mask_push(m3C, prod3 > cb3 * r);
            }
            prod12 = prod2;
            det1 = ds;
        }
    };
    if (MODE >= 2) {
        // LDS-resident (or pixel-major) series: groups of VE dates, one group of 16-byte reads ahead
        constexpr int VE = 16 / (int)sizeof(T);
        DateVal<T> w1, w2;                                     // dates t + 1, t + 2
        w1.a = w1.d = w2.a = w2.d = (T)1;
        w1.b = w1.c = w2.b = w2.c = (T)0;
        // The entry of the NEXT date is requested before this date is worked on: at one or two
        // waves per SIMD (LDS images) nothing else hides the scalar load's latency, 24 times per wave.
        StreamEntry enext = ss.e[1];
        auto process_w = [&](const DateVal<T> &q, const int t) {
            const StreamEntry ecur = enext;
            enext = ss.e[t > 0 ? k - t + 1 : 1];
            process(q, w1, w2, t, ecur);
            w2 = w1;
            w1 = q;
        };
        DateVal<T> cur[VE], nxt[VE];
        rd.template load_group<VE>(k - VE, nxt);
        if (MODE == 4 && (k % (2 * VE)) == 0) {
            // From memory, two adjacent 16-byte pieces per variable and step: both land in the same
            // 64-byte sector, so a sector of C11 / C22 is fetched by two steps instead of four and
            // a sector of an interleaved C12 exactly once (a lane's sectors do not survive in the
            // caches from one step to the next: 16 waves per CU x 64 lanes x 4 variables).
            constexpr int V2 = 2 * VE;
            DateVal<T> cu2[V2], nx2[V2];
            auto load2 = [&](int t0, DateVal<T> (&q)[V2]) {
                DateVal<T> lo[VE], hi[VE];
                rd.template load_group<VE>(t0, lo);
                rd.template load_group<VE>(t0 + VE, hi);
#pragma unroll
                for (int i = 0; i < VE; ++i) {
                    q[i] = lo[i];
                    q[VE + i] = hi[i];
                }
            };
            load2(k - V2, nx2);
            for (int t0 = k - V2; t0 >= 0; t0 -= V2) {
#pragma unroll
                for (int i = 0; i < V2; ++i) cu2[i] = nx2[i];
                if (t0 >= V2) load2(t0 - V2, nx2);
#pragma unroll
                for (int i = V2 - 1; i >= 0; --i) process_w(cu2[i], t0 + i);
            }
        } else if (MODE >= 3) {
            // from memory: two groups ahead (the scattered 16-byte reads take longer than a group's
            // worth of arithmetic)
            DateVal<T> nx2[VE];
            if (k >= 2 * VE) rd.template load_group<VE>(k - 2 * VE, nx2);
            for (int t0 = k - VE; t0 >= 0; t0 -= VE) {
#pragma unroll
                for (int i = 0; i < VE; ++i) {
                    cur[i] = nxt[i];
                    nxt[i] = nx2[i];
                }
                if (t0 >= 2 * VE) rd.template load_group<VE>(t0 - 2 * VE, nx2);
#pragma unroll
                for (int i = VE - 1; i >= 0; --i) process_w(cur[i], t0 + i);
            }
        } else {
            for (int t0 = k - VE; t0 >= 0; t0 -= VE) {
#pragma unroll
                for (int i = 0; i < VE; ++i) cur[i] = nxt[i];
                if (t0 >= VE) rd.template load_group<VE>(t0 - VE, nxt);
#pragma unroll
                for (int i = VE - 1; i >= 0; --i) process_w(cur[i], t0 + i);
            }
        }
    } else {
        // Whole groups of NS dates first, with nothing conditional inside a group (a date skipped
        // under a run-time condition is a control-flow join inside the loop).  The last groups re-read
        // date 0 in place of the dates in front of the series (cache hits).
        int tb = k - 1;
        for (; tb >= NS - 1; tb -= NS) {
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int t = tb - u;
                process(ring[u], ring[(u + NS - 1) % NS], ring[(u + NS - 2) % NS], t, ss.e[k - t]);   // wave-uniform: one scalar load
                ring[(u + NS - 2) % NS] = rd.load(t >= PF ? t - PF : 0);     // PF dates in flight
            }
        }
        // what is left in front: fewer than NS dates
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int t = tb - u;
            if (t >= 0) {
                process(ring[u], ring[(u + NS - 1) % NS], ring[(u + NS - 2) % NS], t, ss.e[k - t]);   // wave-uniform: one scalar load
                if (u < 2) ring[(u + NS - 2) % NS] = rd.load(t >= PF ? t - PF : 0);
            }
        }
    }
    // Every product the reference forms is PP(l) / PP(l + m): with the exponents of PP within 900
    // of each other none of them overflows or loses precision to subnormals, so its logarithm is what
    // the screen models.  (PP moves by at most 36 -- float64 data: 100 -- binary orders per date, so it
    // is still a normal number, with a true exponent, at the first date that breaks the bound.)
    bad = bad || (emax - emin > 900);
    // tests that do not exist: the global test of the last date alone, the 2- / 3-date marginal tests
    // reaching behind the series
    MT gI = mask_undecided(gF, gC), m2I = mask_undecided(m2F, m2C), m3I = mask_undecided(m3F, m3C);
    mask_keep_low(gF, k - 1);
    mask_keep_low(gI, k - 1);
    mask_keep_low(m2F, k - 2);
    mask_keep_low(m2I, k - 2);
    mask_keep_low(m3F, k >= 3 ? k - 3 : 0);
    mask_keep_low(m3I, k >= 3 ? k - 3 : 0);
    // Nodata pixels need no exact pass.  A NaN determinant at any date makes the product of
    // determinants NaN, an exactly zero one makes it 0 (or NaN): ln Q of the test over the whole
    // series is then NaN or -inf, z is NaN or +-inf, and P is NaN or 0 -- never above alpha
    // (nd/_change.pyx:239-242), so the search ends at its first step with no change anywhere.
    if (dead) {
        bad = false;
        gF = mask_zero<MT>();
        gI = mask_zero<MT>();
    }

    const unsigned shard = (unsigned)(b % kShards);
    const int64_t wpx0 = bpx0 + (tid & ~63);                  // first pixel of this wave in its row
    const int64_t wleft = g.nx - wpx0;
    const int wnp = wleft > 64 ? 64 : (wleft > 0 ? (int)wleft : 0);
    uint8_t *wob = g.change + (row * g.nx + wpx0) * (int64_t)k;

    // ---- is this wave dense?  candidates = pixels whose global test over the whole series can fire
    const bool cand = in && (bad || mask_bit(gF, 0) || mask_bit(gI, 0));
    const bool dense = __popcll(__ballot(cand)) >= g.dense_min;
    bool listed = cand;                                       // a sparse wave lists its candidates
    MT mask = mask_zero<MT>();
    if (dense) {
        bool handoff = in && bad;
        bool done = !in || bad;
        int cur = 0;
        // marginal tests over 4 and more dates of the segment starting at l (the lane's own): the
        // dates of ts[l:] once more, from memory.  Sets `fire` (the date of the change) or hands over.
        auto deep_search = [&](const int l, int &fire) {
            T s11 = (T)0, s12r = (T)0, s12i = (T)0, s22 = (T)0;
            int Ld = 0;
            LmT Lmd = 0;
            bool searching = true;
            // four dates in flight per round trip (most searches end at j = 4 or 5)
            for (int t0 = l; t0 < k && searching; t0 += 4) {
                DateVal<T> qb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) qb[u] = rd.load_lane(t0 + u < k ? t0 + u : k - 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + u;
                    if (searching && t < k) {
                        const DateVal<T> q = qb[u];
                        s11 = s11 + q.a;
                        s12r = s12r + q.b;
                        s12i = s12i + q.c;
                        s22 = s22 + q.d;
                        const T det = (q.a * q.d) - ((q.b * q.b) + (q.c * q.c));
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
                                const T dets = (s11 * s22) - ((s12r * s12r) + (s12i * s12i));
                                const bool oks = (dets > (T)0) && (dets < (T)INFINITY);
                                const StreamEntry se = ss.e[jj];       // the lane's own jj: a vector load
                                DenseScreenEntry c;
                                c.re = se.re;
                                c.rf = se.rf;
                                const float x = dense_x<T>(dets, oks, Ld, Lmd, jj, c);
                                if (oks && (x < se.a)) {
                                    fire = t;
                                    searching = false;
                                } else if (!(oks && (x > se.b))) {      // undecided
                                    handoff = true;
                                    done = true;
                                    searching = false;
                                }
                            }
                        }
                    }
                }
            }
        };
        if constexpr (MW < 2) {
            // The walk of single_pixel_change_detection (nd/_change.pyx:224-257) over the test bits,
            // whole runs at a time.  At a segment start l (bit l of each mask):
            //   A1  the global test fires and the 2-date marginal fires (or l + 1 is the last date,
            //       where it IS the global test)                        -> change at l + 1, go on there
            //   A2  ... 2 dates do not fire, 3 dates do (or l + 2 is the last date) -> change at l + 2
            //   HO  a test the walk needs is undecided                  -> exact pass
            //   DP  2 and 3 dates decidedly do not fire                 -> deeper marginals from memory
            //   otherwise the global test decidedly does not fire       -> the search ends (:241-242)
            // A run of A1 bits is a run of changes at consecutive dates: one count-trailing-zeros.
            const MT one = (MT)1;
            const MT LA = k >= 2 ? one << (k - 2) : (MT)0;         // (a single date: no test at all)
            const MT LB = k >= 3 ? one << (k - 3) : (MT)0;
            const MT gOK = gF & ~gI;
            const MT A1 = gOK & (LA | m2F);
            const MT n2 = gOK & ~A1 & ~m2I;
            const MT A2 = n2 & (LB | m3F);
            const MT HO = gI | (gOK & ~A1 & m2I) | (n2 & ~A2 & m3I);
            const MT DP = n2 & ~A2 & ~m3I;
            while (__any(!done)) {
                bool deep = false;
                if (!done) {
                    const int z = mask_ctz((MT)~(A1 >> cur));          // A1 has no bit at k - 1 or above
                    mask |= ((one << z) - one) << (cur + 1);           // :252, l + r with r = j - 1
                    cur += z;                                          // :255
                    if (cur >= k - 1) {
                        done = true;                                   // :256
                    } else {
                        const MT bit = one << cur;
                        if (A2 & bit) {
                            mask |= bit << 2;
                            cur += 2;
                            if (cur >= k - 1) done = true;
                        } else if (HO & bit) {
                            handoff = true;
                            done = true;
                        } else if (DP & bit) {
                            deep = true;
                        } else {
                            done = true;
                        }
                    }
                }
                if (__any(deep)) {
                    if (deep) {
                        int fire = -1;
                        deep_search(cur, fire);
                        if (fire >= 0) {
                            mask_set(mask, fire, true);
                            cur = fire;
                            if (cur >= k - 1) done = true;
                        }
                    }
                }
            }
        } else {
            for (int l = 0; l < k - 1; ++l) {
                bool act = !done && (cur == l);
                if (!__any(act)) continue;
                const bool gi = mask_bit(gI, l), gf = mask_bit(gF, l);
                const bool i2 = mask_bit(m2I, l), f2 = mask_bit(m2F, l);
                const bool i3 = mask_bit(m3I, l), f3 = mask_bit(m3F, l);
                int fire = -1;
                bool deep = false;
                if (act) {
                    if (gi) {                                     // global test undecided
                        handoff = true;
                        done = true;
                    } else if (!gf) {                             // :241-242
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
                    if (deep) deep_search(l, fire);
                }
                if (fire >= 0) {
                    mask_set(mask, fire, true);                   // :252, l + r with r = j - 1
                    cur = fire;                                   // :255
                    if (cur >= k - 1) done = true;                // :256
                }
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
            base = atomicAdd(g.flag_count + shard * kCounterStride, (unsigned)__popcll(lm_));
        base = __shfl(base, 0);
        if (listed) {
            const unsigned slot = base + (unsigned)__popcll(lm_ & ((1ull << lane) - 1ull));
            g.flag_idx[(size_t)shard * g.seg + slot] = (uint32_t)(row * g.nx + x0);
            if (slot < g.dump_cap) {
                // the series was streamed, not kept: the (rare) listed pixel is read once more
                T *d = g.dump + ((int64_t)shard * g.dump_cap + slot) * (int64_t)(4 * k);
                for (int t = 0; t < k; ++t) {
                    const DateVal<T> q = rd.load(t);
                    Pack<T, 4> o;
                    o.v[0] = q.a;
                    o.v[1] = q.b;
                    o.v[2] = q.c;
                    o.v[3] = q.d;
                    *reinterpret_cast<Pack<T, 4> *>(d + 4 * t) = o;
                }
            }
        }
    }
    // ---- a sparse wave zero-fills its own slice of the change map (np.zeros, nd/_change.pyx:275)
    if (!dense && wnp > 0) zero_fill_span(wob, wnp * k, lane);
}

// -----------------------------------------------------------------------------------------
// The chain search for series that do not fit the registers (33 .. 128 dates; float64: 25 .. 128), in
// TWO STREAMING PASSES over the planes: pass 1 as omnibus_c2_stream_kernel's global tests (dates last
// to first, suffix sums, two mask bits per date), pass 2 as dense_chain's forward pass with the dates
// read again, first to last (the wave's 4 k lines are re-read a few microseconds after pass 1
// touched them).  ~110 vector instructions per date whatever the threshold, against the streaming
// search's deep searches from memory, which at thresholds between 0.02 and the sparse regime take
// several times the streaming pass itself (48 x 2048 x 4096, alpha = 0.2: 6.6 ms).  Twice the
// traffic of the one-pass forms: the choice for these thresholds only.
// -----------------------------------------------------------------------------------------
// STATS (round 6): the z / P rasters of the whole-series test from the SAME two passes -- the second pass
// walks the dates first to last anyway, so it carries the reference's forward fold of the whole series
// (nd/_change.pyx:53-77: four sums in `floating`, the double product) next to the running state of the
// current segment, five additions and a multiplication per date; every wave then takes the second pass.
// (Up to round 5 a plain pass A of its own produced the rasters in front of the search: a third read of the
// stack -- 96 dates x 2048 x 4096: + 2.3 ms.  The one-pass streaming search cannot do this: it walks the
// dates last to first, and the reference's forward float32 sums cannot be formed backwards.)
template <typename T, int PF, int MODE, int MW, bool STATS = false>
__global__ void __launch_bounds__(kRetainThreads)
omnibus_c2_stream_chain_kernel(const OmniGlobalArgs<T> g, const OmniTab tab, const StreamScreen<stream_nj(MW)> ss)
{
    static_assert(MODE == 0 || MODE == 1, "planar inputs");
    if (omni_gate_skip(g)) return;
    constexpr int kMaxDates = stream_nj(MW);
    typedef typename std::conditional<MW == 3, Bits192, typename std::conditional<MW == 2, Bits128,
                                      typename std::conditional<MW == 1, unsigned long long, unsigned>::type>::type>::type MT;
    __shared__ StreamEntry tab_lds[kMaxDates + 1];
    __shared__ __align__(16) uint32_t out_img[(kRetainThreads / 64) * 16 * kMaxDates];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int64_t b = blockIdx.x;
    const int64_t row = b / g.blocks_per_row;
    const int64_t bx = b - row * g.blocks_per_row;
    const int64_t bpx0 = bx * (int64_t)kRetainThreads;
    const int64_t x0 = bpx0 + tid;
    const int k = g.k;
    const bool in = x0 < g.nx;
    for (int j = tid; j <= kMaxDates; j += kRetainThreads) tab_lds[j] = ss.e[j];
    if (g.write_tab && b == 0) {
        for (int j = tid; j <= k; j += kRetainThreads) g.tab_dev[j] = tab.e[j];
    }
    PlaneReader<T, MODE> rd;
    {
        const int64_t xc = in ? x0 : g.nx - 1;                  // idle lanes re-read the last pixel
        if (MODE == 1) {
            const int64_t ub = row * g.sy + bpx0;
            rd.voff = (unsigned)(xc - bpx0) * (unsigned)sizeof(T);
            rd.r11 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c11 + ub), 0, 0x7fffffff, 0x00020000);
            rd.r12r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c12r + ub), 0, 0x7fffffff, 0x00020000);
            rd.r12i = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c12i + ub), 0, 0x7fffffff, 0x00020000);
            rd.r22 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(g.c22 + ub), 0, 0x7fffffff, 0x00020000);
            rd.sstep = (unsigned)g.st * (unsigned)sizeof(T);
        } else {
            const int64_t off0 = row * g.sy + xc * g.sx;
            rd.p11 = g.c11 + off0;
            rd.p12r = g.c12r + off0;
            rd.p12i = g.c12i + off0;
            rd.p22 = g.c22 + off0;
            rd.st = g.st;
        }
    }
    __syncthreads();

    // ---- pass 1: the global test of every segment start, dates last to first ----
    const T dlo = (T)ss.dlo, dhi = (T)ss.dhi;
    MT gF = mask_zero<MT>(), gC = mask_zero<MT>();
    bool bad = false, dead = false;
    {
        DateVal<T> ring[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = k - 1 - u;
            ring[u] = rd.load(t > 0 ? t : 0);
        }
        double S11 = 0.0, S12r = 0.0, S12i = 0.0, S22 = 0.0, PP = 1.0;
        int emin = 1, emax = 1;
        auto back = [&](const DateVal<T> &q, const int t, const StreamEntry &c) {
            const T det = (q.a * q.d) - ((q.b * q.b) + (q.c * q.c));
            bad = bad | !((q.a > (T)0) & (det > dlo) & (det < dhi));
            dead = dead | !((det > (T)0) | (det < (T)0));
            PP = PP * (double)det;
            S11 += (double)q.a;
            S12r += (double)q.b;
            S12i += (double)q.c;
            S22 += (double)q.d;
            const int jj = k - t;
            const double pp = S11 * S22;
            const double dets = pp - ((S12r * S12r) + (S12i * S12i));
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
            const int E = (eP - c.re) - __mul24(jj, es);
            const float x = (float)E + __builtin_fmaf(-c.jf, ms, mP - c.rf);
            const float qq = (float)pp * __builtin_amdgcn_rcpf(df);
            const float rel = c.cj * qq;
            const float m2 = c.mj * rel;
            bad = bad | !(okd & (rel < 0.01f));
            mask_push(gF, x + m2 < c.a);
            mask_push(gC, x - m2 > c.b);
        };
        int tb = k - 1;
        for (; tb >= PF - 1; tb -= PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = tb - u;
                back(ring[u], t, ss.e[k - t]);
                ring[u] = rd.load(t >= PF ? t - PF : 0);
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
    if (dead) {
        bad = false;
        gF = mask_zero<MT>();
        gI = mask_zero<MT>();
    }
    const unsigned shard = (unsigned)(b % kShards);
    const int64_t wpx0 = bpx0 + (tid & ~63);
    const int64_t wleft = g.nx - wpx0;
    const int wnp = wleft > 64 ? 64 : (wleft > 0 ? (int)wleft : 0);
    uint8_t *wob = g.change + (row * g.nx + wpx0) * (int64_t)k;
    const bool cand = in && (bad || mask_bit(gF, 0) || mask_bit(gI, 0));
    const bool dense = STATS || __popcll(__ballot(cand)) >= g.dense_min;
    bool listed = cand;
    if (dense) {
        // ---- pass 2: marginal tests and restarts, dates first to last ----
        // A global test pass 1 could not decide (its band grows with the square of the series
        // length: at 96 dates a few per cent of the pixels have one) is decided HERE, exactly: the
        // search goes on as if it fired, and a second running state -- the reference's own sums and
        // product from that segment start, carried on across the restarts to the last date -- gives
        // the test's determinants bit for bit, hence the tight band of the marginal tests.  If it
        // turns out not to fire, the changes recorded behind its start are dropped.  One pending test
        // per pixel; a second one hands the pixel over.
        bool handoff = in && bad;
        bool done = !in || bad || !(mask_bit(gF, 0) || mask_bit(gI, 0));
        bool pend = !done && mask_bit(gI, 0);
        int lp = 0;
        T p11 = (T)0, p12r = (T)0, p12i = (T)0, p22 = (T)0;
        double PQ = 1.0;
        MT mask = mask_zero<MT>();
        DateVal<T> ring[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) ring[u] = rd.load(u < k ? u : k - 1);
        T s11 = (T)0, s12r = (T)0, s12i = (T)0, s22 = (T)0;
        double PP = 1.0;
        int j = 0;
        Accum<T> W;                                              // STATS: the whole series, never restarted
        W.reset();
        auto fwd = [&](const DateVal<T> &q, const int t) {
            const bool last = (t == k - 1);
            const T det = (q.a * q.d) - ((q.b * q.b) + (q.c * q.c));
            if (STATS) W.step(q.a, q.b, q.c, q.d);
            s11 = s11 + q.a;
            s12r = s12r + q.b;
            s12i = s12i + q.c;
            s22 = s22 + q.d;
            PP = PP * (double)det;
            j = j + 1;
            const T dets = (s11 * s22) - ((s12r * s12r) + (s12i * s12i));
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
            // (the first date of the series only starts the state: j = 1, no test)
            const bool tested = t > 0;
            const bool fires = tested & (last | (oks & (x < ca)));
            const bool cant = !tested | (!last & oks & (x > cb));
            const bool act = !done;
            const bool und = act & !(fires | cant);
            const bool f = act & fires;
            handoff = handoff | und;
            mask_set(mask, t, f);                                // :252
            const bool gi = mask_bit(gI, t), gf = mask_bit(gF, t);
            const bool newp = f & !last & gi;                    // an undecided global test starts here
            handoff = handoff | (newp & pend);
            done = done | und | (newp & pend) | (f & (last | !(gf | gi)));   // :256, :241-242
            p11 = p11 + q.a;
            p12r = p12r + q.b;
            p12i = p12i + q.c;
            p22 = p22 + q.d;
            PQ = PQ * (double)det;
            const bool startp = newp & !pend;
            p11 = startp ? q.a : p11;
            p12r = startp ? q.b : p12r;
            p12i = startp ? q.c : p12i;
            p22 = startp ? q.d : p22;
            PQ = startp ? (double)det : PQ;
            lp = startp ? t : lp;
            pend = pend | startp;
            s11 = f ? q.a : s11;
            s12r = f ? q.b : s12r;
            s12i = f ? q.c : s12i;
            s22 = f ? q.d : s22;
            PP = f ? (double)det : PP;
            j = f ? 1 : j;
        };
        int tb = 0;
        for (; tb + PF <= k; tb += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = tb + u;
                fwd(ring[u], t);
                ring[u] = rd.load(t + PF < k ? t + PF : k - 1);
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
            const T dets = (p11 * p22) - ((p12r * p12r) + (p12i * p12i));
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
        if (STATS) {
            const T z = z_stat<T>(W, k, g.nlooks, g.e);
            double zd[1] = {(double)z}, P1[1], P2[1];
            chisq_pair<1>(zd, 4 * (k - 1), g.e.lgam, P1, P2);
            const T P = combine_P<T>(P1[0], P2[0], g.e.omega2);
            if (in) {
                const int64_t pix = row * g.nx + x0;
                if (g.z_out) g.z_out[pix] = z;
                if (g.p_out) g.p_out[pix] = P;
            }
        }
    }
    if (__any(listed)) {
        const unsigned long long lm_ = __ballot(listed);
        unsigned base = 0;
        if (lane == 0)
            base = atomicAdd(g.flag_count + shard * kCounterStride, (unsigned)__popcll(lm_));
        base = __shfl(base, 0);
        if (listed) {
            const unsigned slot = base + (unsigned)__popcll(lm_ & ((1ull << lane) - 1ull));
            g.flag_idx[(size_t)shard * g.seg + slot] = (uint32_t)(row * g.nx + x0);
            if (slot < g.dump_cap) {
                T *d = g.dump + ((int64_t)shard * g.dump_cap + slot) * (int64_t)(4 * k);
                for (int t = 0; t < k; ++t) {
                    const DateVal<T> q = rd.load(t);
                    Pack<T, 4> o;
                    o.v[0] = q.a;
                    o.v[1] = q.b;
                    o.v[2] = q.c;
                    o.v[3] = q.d;
                    *reinterpret_cast<Pack<T, 4> *>(d + 4 * t) = o;
                }
            }
        }
    }
    if (!dense && wnp > 0) zero_fill_span(wob, wnp * k, lane);
}

// =========================================================================================
// pass B
// =========================================================================================
template <typename T>
struct OmniSearchArgs {
    const T *c11, *c12r, *c12i, *c22;
    int64_t nx;
    int64_t sy, sx, st;
    int m11, m12, m22;        // element-offset multipliers per variable (pixel-major inputs: 1 or 2)
    int k;
    double nlooks, alpha;
    uint8_t *change;
    const uint32_t *flag_count, *flag_idx;
    uint32_t seg;
    const OmniTabEntry *tab;
    const T *dump;
    uint32_t dump_cap;
    // Hand-over between the register form and the exact form of pass B: one word per 64 list
    // entries of a shard (bit = the screen could not decide a test of that pixel).  The register
    // form writes every word of the lists it walks; the LDS form, run behind it with the same
    // lists, searches exactly the marked pixels.  nullptr: the LDS form searches every pixel.
    unsigned long long *hand_bits;
    uint32_t hand_words;          // words per shard
    uint32_t *hand_count;         // number of marked pixels (zeroed with the list counters)
    // Shards whose list holds at most this many pixels are searched by omnibus_c2_search_starts_kernel
    // (one lane per segment start), the longer ones by the LDS form; 0: the LDS form takes everything.
    uint32_t starts_max;
    // omnibus_c2_search_kernel starts at this entry of every shard's list (the entries in front of it
    // belong to omnibus_c2_search_rounds_kernel); 0: every entry
    uint32_t first;
};

// MODE 0: series staged in LDS; MODE 1: no LDS, each date read straight from the dump (or the
// planes) with the next date's load issued one iteration ahead -- 4 waves per SIMD instead of the
// 1.5 the 24.5 KB LDS image allows.
// PXW: pixels per wave (64, or 32 / 16 with the upper lanes idle: the LDS image of PXW series of a long
// stack is what limits the waves per CU -- 98 KB for 64 series of 96 dates, one wave; the sweep costs a wave
// the same whatever its number of live lanes, so narrower waves, more of them per CU, finish the list sooner)
// FS (round 6): the screen of a test is the float32 one of dense_chain's second pass (omnibus_c2_device.hpp) --
// x = log2(prod det) - j log2(det of sum) relative to the decision point, from the exponents and the hardware
// log2 of the mantissas of the reference's own running values, against the per-j band of make_dense_entry --
// instead of z_approx in double: ~30 float32 instructions per date where the double form takes ~25 double
// ones on top (the sweep of a long series is bound by them: 96 dates x 164 000 pixels 0.41 -> 0.2 ms).
// Whatever it cannot decide (band, non-positive or non-finite determinants, a product outside the normal
// doubles) takes the exact evaluation, as before.  The table covers tests over up to 192 dates.
constexpr int kScreenLong = 192;
struct DenseScreenLong {
    DenseScreenEntry e[kScreenLong + 1];
};

template <typename T, int MODE, int PXW = 64, bool FS = false>
__global__ void __launch_bounds__(64) omnibus_c2_search_kernel(const OmniSearchArgs<T> s, const DenseScreenLong fscr)
{
    constexpr bool USE_LDS = (MODE == 0);
    extern __shared__ __align__(16) unsigned char nd_smem[];
    T *lds = reinterpret_cast<T *>(nd_smem);
    const int lane = threadIdx.x;
    const int k = s.k;
    // Per-j constants of the screen (m2rho, pklogk, zlo_a, zhi_a): every lane looks up its own j
    // in every iteration, so they sit in LDS as four separate arrays of doubles -- consecutive j in
    // consecutive banks (as 64-byte records all j of one parity would share a bank).
    const OmniTabEntry *tabp = s.tab;          // full records, read only on the rare exact path
    const int kp = k + 1;
    // (MODE 1 keeps no series image: the constants alone, 32 (k + 1) bytes)
    double *scr = reinterpret_cast<double *>(nd_smem + (USE_LDS ? (size_t)k * 4 * PXW * sizeof(T) : 0));
    DenseScreenEntry *scr_f = reinterpret_cast<DenseScreenEntry *>(scr);      // FS: one 16-byte record per j
    if (FS) {
        for (int j = lane; j <= k; j += 64) scr_f[j] = fscr.e[j];
        __syncthreads();
    } else {
        for (int j = lane; j <= k; j += 64) {
            const OmniTabEntry e = s.tab[j];
            scr[j] = e.m2rho;
            scr[kp + j] = e.pklogk;
            scr[2 * kp + j] = e.zlo_a;
            scr[3 * kp + j] = e.zhi_a;
        }
        __syncthreads();
    }
    auto screen = [&](int f, int j) -> double { return scr[f * kp + j]; };
    // blocks shard, shard + kShards, ... work through the list of one shard
    const unsigned shard = blockIdx.x % kShards;
    const unsigned lblock = blockIdx.x / kShards;
    const unsigned nlblock = gridDim.x / kShards;
    const uint32_t n = s.flag_count[shard * kCounterStride];
    const uint32_t *list = s.flag_idx + (size_t)shard * s.seg;
    if (n <= s.starts_max) return;             // omnibus_c2_search_starts_kernel searches this shard

    for (uint32_t base = s.first + lblock * (unsigned)PXW; base < n; base += nlblock * (unsigned)PXW) {
        const uint32_t idx = base + lane;
        bool active = idx < n && lane < PXW;
        if (s.hand_bits != nullptr) {
            // behind the register form: only the pixels it marked (usually none at all)
            if (__builtin_nontemporal_load(s.hand_count) == 0u) return;
            const unsigned long long marked = s.hand_bits[(size_t)shard * s.hand_words + (base >> 6)];
            if (marked == 0ull) continue;
            active = active && ((marked >> lane) & 1ull);
        }
        const int64_t pix = active ? (int64_t)list[idx] : 0;
        const int64_t row = pix / s.nx;
        const int64_t col = pix - row * s.nx;
        const int64_t off = row * s.sy + col * s.sx;

        if (USE_LDS && (PXW == 64 || lane < PXW)) {
            // stage this lane's series: lds[(t*4+v)*PXW + lane]; each lane reads back only its
            // own column, so no barrier is needed.  Source: the dump pass A wrote (one 16/32-byte
            // load per date, eight dates in flight), or the planes for pixels beyond the dump
            // capacity.
            if (idx < s.dump_cap) {
                const T *d = s.dump + ((int64_t)shard * s.dump_cap + idx) * (int64_t)(4 * k);
                for (int t0 = 0; t0 < k; t0 += 8) {
                    Pack<T, 4> q[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (t0 + u < k)
                            q[u] = *reinterpret_cast<const Pack<T, 4> *>(d + 4 * (t0 + u));
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (t0 + u < k) {
                            const int t = t0 + u;
                            lds[(t * 4 + 0) * PXW + lane] = q[u].v[0];
                            lds[(t * 4 + 1) * PXW + lane] = q[u].v[1];
                            lds[(t * 4 + 2) * PXW + lane] = q[u].v[2];
                            lds[(t * 4 + 3) * PXW + lane] = q[u].v[3];
                        }
                }
            } else {
                for (int t0 = 0; t0 < k; t0 += 4) {
                    T q[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (t0 + u < k) {
                            const int64_t o = off + (int64_t)(t0 + u) * s.st;
                            q[u][0] = s.c11[o * s.m11];
                            q[u][1] = s.c12r[o * s.m12];
                            q[u][2] = s.c12i[o * s.m12];
                            q[u][3] = s.c22[o * s.m22];
                        }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (t0 + u < k) {
                            const int t = t0 + u;
                            lds[(t * 4 + 0) * PXW + lane] = q[u][0];
                            lds[(t * 4 + 1) * PXW + lane] = q[u][1];
                            lds[(t * 4 + 2) * PXW + lane] = q[u][2];
                            lds[(t * 4 + 3) * PXW + lane] = q[u][3];
                        }
                }
            }
        }
        // MODE 1: date t of this lane's series; `nxt` holds date `nxt_t` fetched one iteration ago
        const bool from_dump = idx < s.dump_cap;
        const T *dsrc = s.dump + ((int64_t)shard * s.dump_cap + (from_dump ? idx : 0)) * (int64_t)(4 * k);
        Pack<T, 4> nxt;
        int nxt_t = -1;
        auto fetch = [&](int t) -> Pack<T, 4> {
            Pack<T, 4> q;
            if (from_dump) {
                q = *reinterpret_cast<const Pack<T, 4> *>(dsrc + 4 * t);
            } else {
                const int64_t o = off + (int64_t)t * s.st;
                q.v[0] = s.c11[o * s.m11];
                q.v[1] = s.c12r[o * s.m12];
                q.v[2] = s.c12i[o * s.m12];
                q.v[3] = s.c22[o * s.m22];
            }
            return q;
        };
        auto load_step = [&](Accum<T> &A, int t) {
            if (USE_LDS) {
                A.step(lds[(t * 4 + 0) * PXW + lane], lds[(t * 4 + 1) * PXW + lane],
                       lds[(t * 4 + 2) * PXW + lane], lds[(t * 4 + 3) * PXW + lane]);
            } else {
                const Pack<T, 4> cur = (nxt_t == t) ? nxt : fetch(t);
                const int tn = t + 1 < k ? t + 1 : t;
                nxt = fetch(tn);                       // in flight while this date is evaluated
                nxt_t = tn;
                A.step(cur.v[0], cur.v[1], cur.v[2], cur.v[3]);
            }
        };

        // nd/_change.pyx:235-257 as ONE sweep per segment.  The reference first tests the global
        // hypothesis over ts[l:], then the marginal ones over ts[l:l+j], j = 2, 3, ...; both are
        // functions of the same running accumulation started at l (the global test is the state
        // after the last date), so each lane sweeps t = l .. k-1 once, remembers the first date at
        // which a marginal test fires, and commits it only if the global test -- known at the end
        // of the sweep -- passes.  Every iteration of the wave is then the same small body (one
        // date + at most one test per lane), whatever segment each lane is in.
        Accum<T> A;
        A.reset();
        int l = 0;                 // segment start
        int t = 0;                 // next date to fold
        int fire_at = -1;          // first date of this segment whose marginal test fired
        bool done = !active;
        uint8_t *res = s.change + pix * (int64_t)k;

        while (__any(!done)) {
            if (!done) {
                load_step(A, t);
                const int jj = t - l + 1;
                const bool last = (t == k - 1);
                // a marginal test while none has fired yet (j >= 2); at the last date the same
                // evaluation is the global test, needed even if a marginal fired earlier
                const bool need = (jj >= 2) && (fire_at < 0 || last);
                // Screen: z from the hardware log2 against the widened bounds.  fires = above the
                // band for certain; inband = the screen cannot tell.  (NaN compares false twice.)
                bool fires = false, inband = false;
                if (need && FS) {
                    const T dets = (A.s11 * A.s22) - ((A.s12r * A.s12r) + (A.s12i * A.s12i));
                    // (the product a positive normal double: its logarithm below is then meaningful, and the
                    //  reference's own log() of it finite)
                    const bool ok = (dets > (T)0) & (dets < (T)INFINITY) & __builtin_amdgcn_class(A.prod, 0x100);
                    const DenseScreenEntry c = scr_f[jj];
                    int es, eP;
                    float ms, mP;
                    log2_parts(ok ? dets : (T)1, es, ms);
                    log2_parts(ok ? A.prod : 1.0, eP, mP);
                    const int E = (eP - c.re) - __mul24(jj, es);
                    const float x = (float)E + __builtin_fmaf(-(float)jj, ms, mP - c.rf);
                    fires = ok & (x < c.a);
                    inband = !(fires | (ok & (x > c.b)));
                } else if (need) {
                    const double za = z_approx<T>(A, jj, s.nlooks, screen(0, jj), screen(1, jj));
                    fires = (za > screen(3, jj)) && (za < INFINITY);
                    inband = (za >= screen(2, jj)) && !fires;
                }
                // The exact evaluation (two double logs, possibly the chi-square pair) sits behind
                // a wave-uniform branch so that the compiler cannot fold it into the loop body: it
                // is needed for ~1e-5 of the tests, the loop body runs for all of them.
                if (__any(inband)) {
                    if (inband) {
                        const OmniTabEntry e = tabp[jj];
                        const T zp = z_stat<T>(A, jj, s.nlooks, e);
                        const double zd = (double)zp;
                        // 0 = cannot fire (z < zlo, or NaN), 1 = fires for certain
                        // (zhi < z < inf), 2 = inside the exact band: needs the chi-square pair
                        int verdict = !(zd >= e.zlo) ? 0 : ((zd > e.zhi && zd < INFINITY) ? 1 : 2);
                        if (verdict == 2) {
                            double zv[1] = {zd}, P1[1], P2[1];
                            chisq_pair<1>(zv, 4 * (jj - 1), e.lgam, P1, P2);
                            const T P = combine_P<T>(P1[0], P2[0], e.omega2);
                            verdict = ((double)P > s.alpha) ? 1 : 0;
                        }
                        fires = (verdict == 1);
                    }
                }
                if (!last) {
                    if (fires && fire_at < 0) fire_at = t;
                    t = t + 1;
                } else {
                    // `fires` is the global test of ts[l:] (for jj == 1 there is nothing to test)
                    if (fires && fire_at < 0) fire_at = t;
                    if (fires && jj >= 2) {
                        res[fire_at] = 1;                  // :252, l + r with r = j - 1
                        l = fire_at;                       // :255
                        if (l >= k - 1) {
                            done = true;                   // :256
                        } else {
                            A.reset();
                            t = l;
                            fire_at = -1;
                        }
                    } else {
                        done = true;                       // :241-242
                    }
                }
            }
        }
    }
}

// -----------------------------------------------------------------------------------------
// pass B behind the time-split pass A (round 6): the sweep in LOCKSTEP ROUNDS on the blocked dump.
// The from-memory form (MODE 1 above) lets every lane walk its own dates: each step of the wave is one
// instruction whose 64 lanes address 64 different lines, and the rate at which a CU looks those up bounds
// the kernel (96 dates x 164 000 pixels: 0.40 ms whatever the read-ahead, the arithmetic or the chunking:
// DESIGN-EXPERIMENTS.md).  Here the 64 listed pixels of a wave are a BLOCK of the dump -- their series
// interleaved date by date -- and the date index is wave-uniform: a round walks the dates from the earliest
// segment start among the wave's unfinished pixels to the end, one contiguous kilobyte per date (four dates
// in flight), every lane folding from its own segment start on; behind the last date a lane commits the first
// firing date of its segment if the global test fired (nd/_change.pyx:235-257, one sweep per segment as in
// omnibus_c2_search_kernel) and starts its next segment there; rounds repeat while a lane has a segment
// left.  The fold, the float32 screen and the exact evaluation behind a wave-uniform branch are the
// sweep's, operation for operation.  List entries beyond the dump's capacity: omnibus_c2_search_kernel
// (OmniSearchArgs::first).
// -----------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64) omnibus_c2_search_rounds_kernel(const OmniSearchArgs<T> s, const DenseScreenLong fscr)
{
    constexpr int PF = 4;                                // dates in flight
    extern __shared__ __align__(16) unsigned char nd_smem_rd[];
    DenseScreenEntry *scr_f = reinterpret_cast<DenseScreenEntry *>(nd_smem_rd);
    const int lane = threadIdx.x;
    const int k = s.k;
    const OmniTabEntry *tabp = s.tab;
    for (int j = lane; j <= k; j += 64) scr_f[j] = fscr.e[j];
    __syncthreads();
    const unsigned shard = blockIdx.x % kShards;
    const unsigned lblock = blockIdx.x / kShards;
    const unsigned nlblock = gridDim.x / kShards;
    const uint32_t nall = s.flag_count[shard * kCounterStride];
    const uint32_t n = nall < s.dump_cap ? nall : s.dump_cap;          // (dump_cap: a multiple of 64)
    const uint32_t *list = s.flag_idx + (size_t)shard * s.seg;

    for (uint32_t base = lblock * 64u; base < n; base += nlblock * 64u) {
        const uint32_t idx = base + lane;
        const bool active = idx < n;
        const int64_t pix = active ? (int64_t)list[idx] : 0;
        // this lane's values of date t: blk + t * 256 (elements)
        const T *blk = s.dump + (((int64_t)shard * (s.dump_cap >> 6) + (base >> 6)) * (int64_t)k * 64 + lane) * 4;
        uint8_t *res = s.change + pix * (int64_t)k;
        Accum<T> A;
        A.reset();
        int l = 0, fire_at = -1;
        bool done = !active;
        while (__any(!done)) {
            // the round starts at the earliest segment start among the unfinished lanes
            int t0 = done ? k : l;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const int o = __shfl_xor(t0, off);
                t0 = o < t0 ? o : t0;
            }
            t0 = __builtin_amdgcn_readfirstlane(t0);
            bool fires = false;                                  // of the test met last: at t = k - 1 the global test
            Pack<T, 4> ring[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int tu = t0 + u < k ? t0 + u : k - 1;
                ring[u] = *reinterpret_cast<const Pack<T, 4> *>(blk + (int64_t)tu * 256);
            }
            for (int tb = t0; tb < k; tb += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int t = tb + u;                        // wave-uniform
                    if (t < k) {
                        const Pack<T, 4> cur = ring[u];
                        const int tn = t + PF < k ? t + PF : k - 1;
                        ring[u] = *reinterpret_cast<const Pack<T, 4> *>(blk + (int64_t)tn * 256);
                        const bool on = !done && t >= l;
                        if (on) A.step(cur.v[0], cur.v[1], cur.v[2], cur.v[3]);
                        const int jj = t - l + 1;
                        const bool last = (t == k - 1);
                        const bool need = on && (jj >= 2) && (fire_at < 0 || last);
                        bool f = false, inband = false;
                        if (need) {
                            const T dets = (A.s11 * A.s22) - ((A.s12r * A.s12r) + (A.s12i * A.s12i));
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
                                const OmniTabEntry e = tabp[jj];
                                const T zp = z_stat<T>(A, jj, s.nlooks, e);
                                const double zd = (double)zp;
                                int verdict = !(zd >= e.zlo) ? 0 : ((zd > e.zhi && zd < INFINITY) ? 1 : 2);
                                if (verdict == 2) {
                                    double zv[1] = {zd}, P1[1], P2[1];
                                    chisq_pair<1>(zv, 4 * (jj - 1), e.lgam, P1, P2);
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
            // behind the last date: `fires` is the global test of ts[l:] (a segment of one date has none)
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

// -----------------------------------------------------------------------------------------
// pass B, one lane per SEGMENT START (short lists; 3 <= k <= 65).  Behind the fused search the list
// holds the few pixels its screen could not decide (a few thousand of 16.7 M), and nearly every
// date of such a pixel is a change: the LDS form walks two dozen segments one after the other,
// each to the end of the series (the global test) -- ~300 dependent date steps, 0.13 - 0.17 ms
// for 5 000 pixels, the time of ONE wave.  Here lane l of a pixel's group sweeps ts[l:] once, on its
// own: nxt(l) = the date of the first firing marginal test if the global test over ts[l:] fires,
// "stop" otherwise -- the same evaluations, in the same order and arithmetic, as the sweep of
// the LDS form started at l.  The pixel's changes are then the chain 0 -> nxt(0) -> nxt(nxt(0)) ...
// (nd/_change.pyx:235-257), followed by the group's first lane through wave shuffles.  The work is
// the same k^2 / 2 date steps per pixel as the LDS form's in this regime; the dependent chain is k.
// -----------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64) omnibus_c2_search_starts_kernel(const OmniSearchArgs<T> s)
{
    const int lane = threadIdx.x;
    const int k = s.k;
    const int ns = k - 1;                       // segment starts per pixel: 0 .. k - 2
    // up to 64 starts: 64 / ns pixels per wave, one start per lane; 65 .. 192 starts (series of up
    // to 193 dates): one pixel per wave, lane l takes the starts l, l + 64 (and l + 128), one after the other
    const int ppw = ns <= 64 ? 64 / ns : 1;     // pixels per wave
    const int nsw = ns <= 64 ? ns : 64;         // lanes per pixel
    const int nsweep = ns <= 64 ? 1 : (ns <= 128 ? 2 : 3);
    const int grp = lane / nsw;
    const int l0 = lane - grp * nsw;
    const unsigned shard = blockIdx.x % kShards;
    const unsigned lblock = blockIdx.x / kShards;
    const unsigned nlblock = gridDim.x / kShards;
    const uint32_t n = s.flag_count[shard * kCounterStride];
    // Two uses.  hand_bits == nullptr: every pixel of a SHORT list (n <= starts_max; the long ones
    // belong to the register forms).  hand_bits != nullptr: behind a register form, the pixels of
    // a long list that it marked (its screen could not decide one of their tests) -- a few hundred
    // of 3e5, again a job whose time is the dependent chain of one wave.
    const bool marked_mode = s.hand_bits != nullptr;
    if (marked_mode ? (n <= s.starts_max) : (n > s.starts_max)) return;
    if (marked_mode && __builtin_nontemporal_load(s.hand_count) == 0u) return;
    const uint32_t *list = s.flag_idx + (size_t)shard * s.seg;
    const OmniTabEntry *tabp = s.tab;

    // marked mode: the list in words of 64 entries, `todo` = the marks of the current word
    uint32_t word = lblock;
    unsigned long long todo = 0ull;
    for (uint32_t base = lblock * (uint32_t)ppw;; base += nlblock * (uint32_t)ppw) {
        uint32_t idx;
        bool active;
        if (marked_mode) {
            while (todo == 0ull && (uint64_t)word * 64u < n) {
                todo = s.hand_bits[(size_t)shard * s.hand_words + word];
                if (todo == 0ull) word += nlblock;
            }
            if (todo == 0ull) break;
            unsigned long long m = todo;            // the grp-th lowest mark is this group's pixel
            for (int i = 0; i < grp; ++i) m &= m - 1ull;
            active = (grp < ppw) && (m != 0ull);
            idx = word * 64u + (uint32_t)(m != 0ull ? __builtin_ctzll(m) : 0);
            for (int i = 0; i < ppw; ++i) todo &= todo - 1ull;
            if (todo == 0ull) word += nlblock;
        } else {
            if (base >= n) break;
            idx = base + (uint32_t)grp;
            active = (grp < ppw) && (idx < n);
        }
        const int64_t pix = active ? (int64_t)list[idx] : 0;
        const int64_t row = pix / s.nx;
        const int64_t col = pix - row * s.nx;
        const int64_t off = row * s.sy + col * s.sx;
        const bool from_dump = active && idx < s.dump_cap;
        const T *dsrc = s.dump + ((int64_t)shard * s.dump_cap + (from_dump ? idx : 0)) * (int64_t)(4 * k);
        auto fetch = [&](int t) -> Pack<T, 4> {
            Pack<T, 4> q;
            if (from_dump) {
                q = *reinterpret_cast<const Pack<T, 4> *>(dsrc + 4 * t);
            } else {
                const int64_t o = off + (int64_t)t * s.st;
                q.v[0] = s.c11[o * s.m11];
                q.v[1] = s.c12r[o * s.m12];
                q.v[2] = s.c12i[o * s.m12];
                q.v[3] = s.c22[o * s.m22];
            }
            return q;
        };
        int nxt_a = -1, nxt_b = -1, nxt_c = -1; // per sweep: the next segment start, or stop
        for (int sweep = 0; sweep < nsweep; ++sweep) {
        const int l = l0 + 64 * sweep;
        const bool lact = active && l < ns;
        Accum<T> A;
        A.reset();
        int fire_at = -1;
        int nxt = -1;                           // stop
        // Four dates per trip, the next four in flight meanwhile: the kernel's time is the dependent
        // chain of one wave, and a load per date (even one date ahead) was most of it.  (Idle lanes
        // load nothing: a stand-in read of pixel 0 from 4 x k planes far apart would put its address
        // translations on every trip's critical path.)
        auto fetch4 = [&](const int i0, Pack<T, 4> (&q)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = l + i0 + u;
                q[u].v[0] = q[u].v[1] = q[u].v[2] = q[u].v[3] = (T)0;
                if (lact && t < k) q[u] = fetch(t);
            }
        };
        Pack<T, 4> qn[4];
        fetch4(0, qn);
        for (int i0 = 0; i0 < k; i0 += 4) {     // the lane's dates are l + i0 .. l + i0 + 3
            if (!__any(lact && l + i0 < k)) break;
            Pack<T, 4> q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = qn[u];
            fetch4(i0 + 4, qn);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                const int t = l + i;
                const bool on = lact && t < k;
                if (on) A.step(q[u].v[0], q[u].v[1], q[u].v[2], q[u].v[3]);
                const int jj = i + 1;
                const bool last = (t == k - 1);
                const bool need = on && (jj >= 2) && (fire_at < 0 || last);
                bool fires = false, inband = false;
                if (need) {
                    const OmniTabEntry &e = tabp[jj];
                    const double za = z_approx<T>(A, jj, s.nlooks, e.m2rho, e.pklogk);
                    fires = (za > e.zhi_a) && (za < INFINITY);
                    inband = (za >= e.zlo_a) && !fires;
                }
                if (__any(inband)) {
                    if (inband) {
                        const OmniTabEntry e = tabp[jj];
                        const T zp = z_stat<T>(A, jj, s.nlooks, e);
                        const double zd = (double)zp;
                        int verdict = !(zd >= e.zlo) ? 0 : ((zd > e.zhi && zd < INFINITY) ? 1 : 2);
                        if (verdict == 2) {
                            double zv[1] = {zd}, P1[1], P2[1];
                            chisq_pair<1>(zv, 4 * (jj - 1), e.lgam, P1, P2);
                            const T P = combine_P<T>(P1[0], P2[0], e.omega2);
                            verdict = ((double)P > s.alpha) ? 1 : 0;
                        }
                        fires = (verdict == 1);
                    }
                }
                if (on && fires && fire_at < 0) fire_at = t;
                if (on && last && fires) nxt = fire_at;      // the global test of ts[l:] fired (jj >= 2 here)
            }
        }
        if (sweep == 0) nxt_a = nxt; else if (sweep == 1) nxt_b = nxt; else nxt_c = nxt;
        }
        // follow the chain from l = 0; the group's first lane writes the changes
        int at = 0;
        for (int step = 0; step < ns; ++step) {
            const int ats = (at >= 0 && at < ns) ? at : 0;
            const int src = grp * nsw + (ats & 63) % nsw;
            const int na = __shfl(nxt_a, src), nb = __shfl(nxt_b, src), nc = __shfl(nxt_c, src);
            const int n1 = (ns > 64 && ats >= 128) ? nc : ((ns > 64 && ats >= 64) ? nb : na);
            if (at >= 0 && at < ns) {
                if (n1 < 0) {
                    at = -1;                                   // :241-242
                } else {
                    if (active && l0 == 0) s.change[pix * (int64_t)k + n1] = 1;  // :252 (the row was zero-filled)
                    at = n1;                                   // :255; n1 = k - 1 ends the search (:256)
                }
            }
        }
    }
}

// -----------------------------------------------------------------------------------------
// pass B, chain form (k <= 32 float / 16 double; round 3): dense_chain on the listed pixel's
// series -- one backward pass for the global tests, one forward pass for the marginal tests and
// the restarts, ~105 vector instructions per date, against one round of the register form above
// PER SEGMENT (a wave lives as long as its lane with the most changes).  The screen only decides
// what is certain; a pixel with a test it cannot decide (or outside its domain) is marked in
// hand_bits and searched exactly by omnibus_c2_search_starts_kernel behind this kernel.
// -----------------------------------------------------------------------------------------
template <typename T, int KMAX>
__global__ void __launch_bounds__(64, 3)
omnibus_c2_search_chain_kernel(const OmniSearchArgs<T> s, const StreamScreen<32> ss)
{
    __shared__ StreamEntry tab_lds[33];
    const int lane = threadIdx.x;
    if (lane <= 32) tab_lds[lane] = ss.e[lane];
    __syncthreads();
    int k = s.k;
    asm volatile("" : "+s"(k));
    const unsigned shard = blockIdx.x % kShards;
    const unsigned lblock = blockIdx.x / kShards;
    const unsigned nlblock = gridDim.x / kShards;
    const uint32_t n = s.flag_count[shard * kCounterStride];
    if (n <= s.starts_max) return;             // omnibus_c2_search_starts_kernel searches this shard
    const uint32_t *list = s.flag_idx + (size_t)shard * s.seg;

    for (uint32_t base = lblock * 64u; base < n; base += nlblock * 64u) {
        const uint32_t idx = base + lane;
        const bool active = idx < n;
        const int64_t pix = active ? (int64_t)list[idx] : 0;
        T v[KMAX][4];
        if (idx < s.dump_cap) {
            const T *d = s.dump + ((int64_t)shard * s.dump_cap + idx) * (int64_t)(4 * k);
#pragma unroll
            for (int t = 0; t < KMAX; ++t) {
                // (dates behind the series: a copy of the last one, masked out by the search)
                const Pack<T, 4> q = *reinterpret_cast<const Pack<T, 4> *>(d + 4 * (t < k ? t : k - 1));
                v[t][0] = q.v[0];
                v[t][1] = q.v[1];
                v[t][2] = q.v[2];
                v[t][3] = q.v[3];
            }
        } else {
            const int64_t row = pix / s.nx;
            const int64_t col = pix - row * s.nx;
            const int64_t off = row * s.sy + col * s.sx;
            const T *p11 = s.c11 + off * s.m11, *p12r = s.c12r + off * s.m12;
            const T *p12i = s.c12i + off * s.m12, *p22 = s.c22 + off * s.m22;
            const int64_t d11 = s.st * s.m11, d12 = s.st * s.m12, d22 = s.st * s.m22;
#pragma unroll
            for (int t = 0; t < KMAX; ++t) {
                v[t][0] = *p11;
                v[t][1] = *p12r;
                v[t][2] = *p12i;
                v[t][3] = *p22;
                if (t < k - 1) {
                    p11 += d11;
                    p12r += d12;
                    p12i += d12;
                    p22 += d22;
                }
            }
        }
        unsigned cmask;
        bool handoff, cand;
        dense_chain<T, KMAX, 32>(v, k, active, ss, tab_lds, cmask, handoff, cand);
        {
            const unsigned long long hm = __ballot(handoff);
            if (lane == 0) {
                s.hand_bits[(size_t)shard * s.hand_words + (base >> 6)] = hm;
                if (hm != 0ull) atomicAdd(s.hand_count, (unsigned)__popcll(hm));
            }
        }
        if (active && !handoff) {
            uint8_t *res = s.change + pix * (int64_t)k;
            for (int u = 1; u < k; ++u)                    // the row was zero-filled by pass A
                if ((cmask >> u) & 1u) res[u] = 1;
        }
    }
}

// =========================================================================================
// host side
// =========================================================================================
std::vector<OmniTabEntry> get_table_impl(int k, uint32_t n_looks, double alpha, int dtype, int pol)
{
    static std::mutex mu;
    static std::vector<TabCacheEntry> cache;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const auto &c : cache)
            if (c.key.k == k && c.key.dtype == dtype && c.key.pol == pol && c.key.n == n_looks &&
                memcmp(&c.key.alpha, &alpha, sizeof(double)) == 0)
                return c.tab;
    }
    std::vector<OmniTabEntry> tab((size_t)k + 1);
    memset(tab.data(), 0, tab.size() * sizeof(OmniTabEntry));
    for (int j = 1; j <= k; ++j)
        tab[j] = dtype == ND_AMD_F32 ? make_entry<float>(j, n_looks, alpha, pol)
                                     : make_entry<double>(j, n_looks, alpha, pol);
    {
        std::lock_guard<std::mutex> lk(mu);
        if (cache.size() >= 32) cache.erase(cache.begin());
        TabCacheEntry c;
        c.key.k = k;
        c.key.dtype = dtype;
        c.key.pol = pol;
        c.key.n = n_looks;
        c.key.alpha = alpha;
        c.tab = tab;
        cache.push_back(c);
    }
    return tab;
}

// workspace: [counters][per-j table][pixel list: npix x u32][dense-wave list][dump: cap x k x 4 x T]
// `min_total` is what the call needs; everything beyond it is used as dump capacity.  The
// recommended size holds the series of 1/8 of the pixels (a listed pixel beyond the capacity is
// gathered from the planes by pass B instead: slower, never wrong).
struct OmniWorkspace {
    size_t off_count, off_tab, off_idx, off_dense, off_hand, off_dump, min_total, recommended;
    uint32_t seg, segd, hand_words;
};

constexpr int kRetainMaxF32 = 48, kRetainMaxF64 = 24;

constexpr size_t kCounterBytes = (size_t)kShards * kCounterStride * sizeof(uint32_t);

// List entries per shard: an upper bound on the pixels of the blocks one shard can receive,
// valid for both pass-A forms (256-pixel blocks, or 256*VPPT-pixel blocks with VPPT <= 4) and for
// row-wise as well as flattened launches:
//   ceil(nb_v / kShards) * 256 * VPPT  <=  2 * nb256 + (2 * ny + 256) * VPPT   (kShards = 128)
static uint32_t omni_seg(int64_t npix, int64_t ny)
{
    static_assert(kShards == 128, "bound below assumes 128 shards");
    const int64_t nb256 = ceil_div(npix, 256) + ny;
    return (uint32_t)(2 * nb256 + (2 * ny + 256) * 4 + 256);
}

// seg_ml != 0: the multilooking pass A (omnibus_ml.hip) numbers its waves differently -- its own list
// length per shard -- and the dump must hold EVERY listed pixel (the multilooked series exists
// nowhere else: pass B cannot gather it from the planes), so the dump is part of the minimum.
static OmniWorkspace omni_layout(int64_t npix, int64_t ny, int64_t k, size_t elem, uint32_t seg_ml = 0)
{
    OmniWorkspace w;
    const uint32_t seg = seg_ml ? seg_ml : omni_seg(npix, ny);
    w.seg = seg;
    w.off_count = 0;
    w.off_tab = align256(kCounterBytes);
    w.off_idx = w.off_tab + align256((size_t)(k + 1) * sizeof(OmniTabEntry));
    w.off_dense = w.off_idx + align256((size_t)seg * kShards * sizeof(uint32_t));
    // dense-wave list: at most one entry per 64 listed pixels of a shard
    w.segd = seg / 64 + 8;
    w.off_hand = w.off_dense + align256((size_t)w.segd * kShards * sizeof(uint32_t));
    // hand-over marks of the register form of pass B: one bit per list entry (OmniSearchArgs::hand_bits)
    w.hand_words = seg / 64 + 1;
    w.off_dump = w.off_hand + align256((size_t)w.hand_words * kShards * sizeof(unsigned long long));
    w.min_total = w.off_dump;
    const size_t per = (size_t)k * 4 * elem;
    size_t cap = ((size_t)npix / 8 / kShards + 63) & ~(size_t)63;   // per shard
    w.recommended = w.off_dump + align256(cap * kShards * per);
    if (seg_ml) {
        w.min_total = w.off_dump + align256((size_t)seg * kShards * per);
        w.recommended = w.min_total;
    }
    return w;
}

template <typename T, int PPT>
static void launch_global(const OmniGlobalArgs<T> &g, const OmniTab &tab, int64_t nblocks,
                          bool stats, hipStream_t stream)
{
    if (stats)
        hipLaunchKernelGGL((omnibus_c2_global_kernel<T, PPT, true>), dim3((unsigned)nblocks),
                           dim3(kGlobalThreads), 0, stream, g, tab);
    else
        hipLaunchKernelGGL((omnibus_c2_global_kernel<T, PPT, false>), dim3((unsigned)nblocks),
                           dim3(kGlobalThreads), 0, stream, g, tab);
}

template <typename T, int KMAX>
static void launch_retain_k(const OmniGlobalArgs<T> &g, const OmniTab &tab, int64_t nblocks,
                            bool stats, hipStream_t stream)
{
    const dim3 grid((unsigned)nblocks), block(kRetainThreads);
#ifndef ND_RETAIN_EXACT
#define ND_RETAIN_EXACT 1
#endif
    if (ND_RETAIN_EXACT && g.k == KMAX && g.sx == 1 &&
        (int64_t)g.k * g.st * (int64_t)sizeof(T) < 0x7fffffffLL && g.st >= 0) {
        if (stats)
            hipLaunchKernelGGL((omnibus_c2_retain_kernel<T, KMAX, true, true>), grid, block, 0,
                               stream, g, tab);
        else
            hipLaunchKernelGGL((omnibus_c2_retain_kernel<T, KMAX, true, false>), grid, block, 0,
                               stream, g, tab);
    } else {
        if (stats)
            hipLaunchKernelGGL((omnibus_c2_retain_kernel<T, KMAX, false, true>), grid, block, 0,
                               stream, g, tab);
        else
            hipLaunchKernelGGL((omnibus_c2_retain_kernel<T, KMAX, false, false>), grid, block, 0,
                               stream, g, tab);
    }
}

template <typename T, int KMAX>
static void launch_chain_k(const OmniGlobalArgs<T> &g, const OmniTab &tab, const StreamScreen<chain_nj(KMAX)> &ss,
                           int64_t nblocks, hipStream_t stream, bool stats = false)
{
    const dim3 grid((unsigned)nblocks), block(kRetainThreads);
    const bool exact = g.k == KMAX && g.sx == 1 && (int64_t)g.k * g.st * (int64_t)sizeof(T) < 0x7fffffffLL && g.st >= 0;
    if (stats) {
        if (exact)
            hipLaunchKernelGGL((omnibus_c2_chain_kernel<T, KMAX, true, true>), grid, block, 0, stream, g, tab, ss);
        else
            hipLaunchKernelGGL((omnibus_c2_chain_kernel<T, KMAX, false, true>), grid, block, 0, stream, g, tab, ss);
    } else if (exact)
        hipLaunchKernelGGL((omnibus_c2_chain_kernel<T, KMAX, true>), grid, block, 0, stream, g, tab, ss);
    else
        hipLaunchKernelGGL((omnibus_c2_chain_kernel<T, KMAX, false>), grid, block, 0, stream, g, tab, ss);
}

// k <= 32 (float) / 24 (double): the series lengths the chain form keeps in registers
template <typename T>
static void launch_chain(const OmniGlobalArgs<T> &g, const OmniTab &tab, const std::vector<OmniTabEntry> &htab,
                         const DenseScreen &scr, uint32_t n_looks, int64_t nblocks, hipStream_t stream,
                         bool stats = false)
{
    const int k = g.k;
    if (sizeof(T) == 8 && k > 16) {
        const StreamScreen<32> ss = make_stream_screen<T, 32>(htab, scr, k, n_looks);
        launch_chain_k<double, 24>(reinterpret_cast<const OmniGlobalArgs<double> &>(g), tab, ss, nblocks, stream, stats);
        return;
    }
    const StreamScreen<32> ss = make_stream_screen<T, 32>(htab, scr, k, n_looks);
    if (k <= 8)
        launch_chain_k<T, 8>(g, tab, ss, nblocks, stream, stats);
    else if (k <= 16)
        launch_chain_k<T, 16>(g, tab, ss, nblocks, stream, stats);
    else if (sizeof(T) == 4) {
        if (k <= 24)
            launch_chain_k<float, 24>(reinterpret_cast<const OmniGlobalArgs<float> &>(g), tab, ss, nblocks, stream, stats);
        else
            launch_chain_k<float, 32>(reinterpret_cast<const OmniGlobalArgs<float> &>(g), tab, ss, nblocks, stream, stats);
    }
}

template <typename T>
static void launch_retain(const OmniGlobalArgs<T> &g, const OmniTab &tab, int64_t nblocks,
                          bool stats, hipStream_t stream)
{
    const int k = g.k;
    if (k <= 8)
        launch_retain_k<T, 8>(g, tab, nblocks, stats, stream);
    else if (k <= 16)
        launch_retain_k<T, 16>(g, tab, nblocks, stats, stream);
    else if (k <= 24)
        launch_retain_k<T, 24>(g, tab, nblocks, stats, stream);
    else if (sizeof(T) == 4) {
        if (k <= 32)
            launch_retain_k<float, 32>(reinterpret_cast<const OmniGlobalArgs<float> &>(g), tab,
                                       nblocks, stats, stream);
        else
            launch_retain_k<float, 48>(reinterpret_cast<const OmniGlobalArgs<float> &>(g), tab,
                                       nblocks, stats, stream);
    }
}

// ND_AMD_SEARCH_FS=0: the sweeps of pass B screen a test with z_approx in double, as before round 6, instead of the
// float32 screen -- and the sparse regime of long series takes the plain pass A and the gather (the lockstep sweep
// behind the time-split pass A has the float32 screen only)
static bool search_fs_enabled()
{
    static const bool v = [] {
        const char *e = getenv("ND_AMD_SEARCH_FS");
        return e ? atoi(e) != 0 : true;
    }();
    return v;
}

// ND_AMD_FUSED_FORM: which fused search serves the thresholds below the sparse regime -- 0 the
// streaming search, 2 dense_chain on the retained series, 3 dense_chain in two streaming passes
// (longer series); unset (-1): by threshold and series length.  Speed only: every form gives the same
// map.  (Form 1, the triangle of dense_search on the retained series -- 3.9 - 5.4 ms where the chain
// form takes 1.55 -- and the round-based register form of pass B, ND_AMD_SEARCH_MODE=2 -- 0.097 ms
// where the chain form takes 0.081 -- lost at every threshold and series length and were deleted in
// round 4.)
static int fused_form_env()
{
    static const int v = [] {
        const char *e = getenv("ND_AMD_FUSED_FORM");
        return e ? atoi(e) : -1;
    }();
    return v;
}

template <typename T>
static int omnibus_c2_impl(const void *c11, const void *c12re, const void *c12im, const void *c22,
                           int64_t ny, int64_t nx, int64_t k, int64_t sy, int64_t sx, int64_t st,
                           uint32_t n_looks, double alpha, uint8_t *change, void *z_out,
                           void *p_out, void *workspace, size_t workspace_bytes,
                           hipStream_t stream, const int64_t *pm_ids = nullptr,
                           const OmniMlPlan *mlp = nullptr)
{
    // mlp != null: OmnibusTest(ml=w) -- pass A multilooks on the fly (omnibus_ml.hip; float32 only)
    // pm_ids != null: pixel-major inputs, element (y, x, t) of variable v at
    // ptr_v[((y * nx + x) * k + t) * pm_ids[v]]; sy / sx / st then describe the unit-stride case
    const int64_t npix = ny * nx;
    const OmniWorkspace w = omni_layout(npix, ny, k, sizeof(T), mlp ? mlp->seg : 0);
    if (workspace == nullptr || workspace_bytes < w.min_total) {
        set_error("nd_amd_omnibus_c2: workspace of at least %zu bytes needed, %zu given",
                  w.min_total, workspace_bytes);
        return ND_AMD_EWORKSPACE;
    }
    if (((uintptr_t)workspace & 255) != 0) {
        set_error("nd_amd_omnibus_c2: workspace must be 256-byte aligned");
        return ND_AMD_EINVAL;
    }
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    uint32_t *flag_count = reinterpret_cast<uint32_t *>(ws + w.off_count);
    OmniTabEntry *tab_dev = reinterpret_cast<OmniTabEntry *>(ws + w.off_tab);
    uint32_t *flag_idx = reinterpret_cast<uint32_t *>(ws + w.off_idx);

    // per-j constants (host, double, same expression order as nd/_change.c:2926-2975)
    const std::vector<OmniTabEntry> htab = get_table<T>((int)k, n_looks, alpha, 2);
    OmniTab tab;
    memset(&tab, 0, sizeof(tab));
    const bool tab_in_args = (k <= kTabArgs);
    if (tab_in_args) {
        memcpy(tab.e, htab.data(), htab.size() * sizeof(OmniTabEntry));
    } else {
        // large k: the table goes through a pageable host copy (synchronises the stream once)
        hipError_t e = hipMemcpyAsync(tab_dev, htab.data(), htab.size() * sizeof(OmniTabEntry),
                                      hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        ND_HIP_CHECK(e);
    }

    ND_HIP_CHECK(hipMemsetAsync(flag_count, 0, kCounterBytes, stream));

    // ---- pass A ----
    OmniGlobalArgs<T> g;
    g.c11 = static_cast<const T *>(c11);
    g.c12r = static_cast<const T *>(c12re);
    g.c12i = static_cast<const T *>(c12im);
    g.c22 = static_cast<const T *>(c22);
    g.sy = sy;
    g.sx = sx;
    g.st = st;
    g.k = (int)k;
    g.write_tab = tab_in_args ? 1 : 0;
    g.nlooks = (double)n_looks;
    g.alpha = alpha;
    g.e = htab[(size_t)k];
    g.change = change;
    g.z_out = static_cast<T *>(z_out);
    g.p_out = static_cast<T *>(p_out);
    g.flag_count = flag_count;
    g.flag_idx = flag_idx;
    g.tab_dev = tab_dev;
    g.dense_idx = reinterpret_cast<uint32_t *>(ws + w.off_dense);
    g.segd = w.segd;
    // the register search exists for the series lengths whose arrays fit the register file
    static const int dense_env = [] {
        const char *e = getenv("ND_AMD_DENSE_MIN");
        return e ? atoi(e) : 16;
    }();
    const bool dense_ok = k <= (sizeof(T) == 4 ? 32 : 16) && pm_ids == nullptr && mlp == nullptr;
    g.dense_min = dense_ok ? dense_env : 65;
    const bool stats = (z_out != nullptr) || (p_out != nullptr);
    const bool retain = k <= (sizeof(T) == 4 ? kRetainMaxF32 : kRetainMaxF64);
    // (round 6) The screen of the whole-series test is unusable where omega2 leaves [0, 1] (omni_bounds): P is then not
    // monotonic in z.  That is the case for the reference's DEFAULT n = 1 on a series of more than a few dates, whatever
    // the data were multilooked with -- and every screen-based form then handed EVERY pixel to the exact pass B
    // (24 x 2048 x 4096, n = 1: 9.2 ms against 0.7 ms with n = 9).  Pass A evaluates that one test exactly instead
    // (its STATS instantiation: z, the chi-square pair, P > alpha as the reference decides it) and lists only the
    // pixels whose whole-series test fires -- for n = 1 on multilooked data none, as in the reference.
    // ND_AMD_EXACT_FLAGS=0: as before.
    static const bool exact_flags_env = [] {
        const char *e = getenv("ND_AMD_EXACT_FLAGS");
        return e ? atoi(e) != 0 : true;
    }();
    const bool global_screen_ok = (htab[(size_t)k].zlo > -INFINITY) || (htab[(size_t)k].zhi < INFINITY);
    const bool exact_flags = exact_flags_env && !global_screen_ok && mlp == nullptr && k >= 2;
    const bool stats_a = stats || exact_flags;      // the pass A instantiation that evaluates the whole-series test exactly
    // Low thresholds make nearly every wave dense (P > alpha holds for a fraction 1 - alpha of
    // stationary pixels): the search is then fused into pass A.  Speed only -- every form gives the
    // same map.  ND_AMD_FUSED_ALPHA overrides the switch-over (0 = never fuse, 2 = always).
    static const double fused_alpha = [] {
        const char *e = getenv("ND_AMD_FUSED_ALPHA");
        return e ? atof(e) : 0.75;
    }();
    // The chain form in registers costs the same at every threshold (1.55 ms on 24 x 4096^2), and the
    // sparse design passes that where a tenth of the pixels are candidates (alpha = 0.76 / 0.8 / 0.85:
    // 2.9 / 2.5 / 2.1 ms): it is offered up to 0.93, and the device-side sample decides.  The forms
    // that pay more for fusing (pixel-major: the LDS images; longer series: two streaming passes)
    // keep 0.75.
    static const double fused_alpha_regs = [] {
        const char *e = getenv("ND_AMD_FUSED_ALPHA");
        return e ? atof(e) : 0.93;
    }();
    // (z / P rasters asked for on top: they come from one launch of the plain pass A first, see below)
    const bool fused = retain && dense_ok && g.dense_min <= 64 && alpha < fused_alpha_regs && !exact_flags;
    const bool fused_stats = fused && stats;
    // Series beyond the register forms (33 .. 128 dates; float64: 17 .. 128): the streaming search with
    // 64- or 128-bit masks.  Without it every pixel of a low-threshold run went through pass B one by one
    // (k = 48 at alpha = 0.01: 118 ms per 16.7 Mpx).
    // (round 6, later) 129 .. 192 dates: the chain search in two streaming passes with three-word masks -- before,
    // every pixel of a low-threshold run on such a series went through pass B (136 dates x 512 x 4096 at the
    // reference's default alpha = 0.01: 340 ms; 128 dates: 2.3 ms).  ND_AMD_STREAM_CHAIN_192=0: as before.
    static const bool chain192_env = [] {
        const char *e = getenv("ND_AMD_STREAM_CHAIN_192");
        return e ? atoi(e) != 0 : true;
    }();
    const bool stream_long = !dense_ok && pm_ids == nullptr && mlp == nullptr &&
                             k <= (chain192_env ? kStreamChainMax : kDenseMax) &&
                             dense_env <= 64 && alpha < fused_alpha && !exact_flags;
    // multilooking pass A: the search fused in (dense_chain on the retained, multilooked series) at
    // every threshold below the sparse regime -- by alpha alone: the density sample reads the planes
    // as they are, not multilooked
    const bool ml_chain = mlp != nullptr && k >= 2 && dense_env <= 64 && alpha < fused_alpha_regs;
    // z / P rasters asked for on top: they come from one launch of the plain pass A (which
    // evaluates the whole-series test of every pixel anyway), the map from the streaming search
    // ND_AMD_STATS_SPLIT=1: the rasters from a pass of their own in front of every fused search, as before round 6
    static const bool stats_split_env0 = [] {
        const char *e = getenv("ND_AMD_STATS_SPLIT");
        return e ? atoi(e) != 0 : false;
    }();
    // (round 6) with rasters, a long series takes the chain form at EVERY low threshold -- in registers (float64,
    // 17 .. 24 dates) or in two streaming passes -- whose forward pass carries the whole-series fold: no separate
    // read for the rasters
    const bool stats_long_chain = stream_long && stats && !stats_split_env0 && k >= 3;
    const bool stats_split = stream_long && stats && !stats_long_chain;
    // (round 6) the sparse regime beyond the register-retaining lengths: the time-split pass A hands the
    // candidates' series to pass B from its registers (omnibus_c2_split_kernel).  ND_AMD_C2_SPLIT=0: the
    // plain pass A and the gather, as before.
    static const bool split_env = [] {
        const char *e = getenv("ND_AMD_C2_SPLIT");
        return e ? atoi(e) != 0 : true;
    }();
    // ND_AMD_C2_SPLIT_STATS=0: with rasters the plain pass A and the gather, as before
    static const bool split_stats_env = [] {
        const char *e = getenv("ND_AMD_C2_SPLIT_STATS");
        return e ? atoi(e) != 0 : true;
    }();
    // (float64 beyond 48 dates -- eight slices of doubles in turn -- measured slower with the rasters than the
    //  plain pass A and the gather: 3.65 against 3.43 ms on 96 x 1024 x 4096)
    const bool split_stats_ok = split_stats_env && !(sizeof(T) == 8 && k > 48);
    const bool split_ok = split_env && search_fs_enabled() && !retain && !stream_long && (!stats || split_stats_ok) && !exact_flags &&
                          k <= (sizeof(T) == 4 ? 192 : 96) && pm_ids == nullptr && mlp == nullptr && sx == 1;
    // The threshold only says that dense waves are LIKELY; whether they are is a property of the
    // data (a low alpha on strongly filtered data fires rarely).  Above a minimum size the choice
    // is therefore made on the device from a sample (omnibus_c2_sample_kernel): both variants are
    // launched, the unfavoured one returns at once.  ND_AMD_GATE=0: decide by alpha alone.
    static const bool gate_env = [] {
        const char *e = getenv("ND_AMD_GATE");
        return e ? atoi(e) != 0 : true;
    }();
    g.gate = flag_count + 2;            // word 2 of shard 0's counter line (zeroed with the counters)
    g.gate_n = 0;
    g.gate_mode = 0;
    {
        const size_t per = (size_t)k * 4 * sizeof(T);
        // (the streaming searches of longer series write the few pixels they hand over as well: pass B
        // would otherwise gather 4 x k isolated values per pixel from planes megabytes apart)
        size_t cap = (retain || stream_long || split_ok) ? (workspace_bytes - w.off_dump) / per / kShards : 0;   // per shard
        g.seg = w.seg;
        if (cap > g.seg) cap = g.seg;
        if (split_ok) cap &= ~(size_t)63;            // (the blocked layout of the time-split pass A: whole blocks of 64 entries)
        g.dump = reinterpret_cast<T *>(ws + w.off_dump);
        g.dump_cap = (uint32_t)cap;
    }

    constexpr int VPPT = 16 / sizeof(T);   // pixels per 16-byte load
    const size_t es = sizeof(T);
    const bool aligned = (sx == 1) && (((uintptr_t)c11 | (uintptr_t)c12re | (uintptr_t)c12im |
                                        (uintptr_t)c22) & 15) == 0 &&
                         ((sy * (int64_t)es) % 16 == 0) && ((st * (int64_t)es) % 16 == 0);
    // pixel-major inputs are walked as one flat row of pixels as well
    const bool flat = ((sx == 1) && (sy == nx)) || pm_ids != nullptr;
    g.nx = flat ? npix : nx;
    g.nrows = flat ? 1 : ny;
    unsigned long long *const hand_ws = reinterpret_cast<unsigned long long *>(ws + w.off_hand);
    const int ppt = retain ? 1 : (aligned ? VPPT : 1);
    g.blocks_per_row = ceil_div(g.nx, retain ? (int64_t)kRetainThreads : (int64_t)kGlobalThreads * ppt);
    const int64_t nblocks = g.blocks_per_row * g.nrows;
    if (nblocks > 0x7fffffffLL) {
        set_error("nd_amd_omnibus_c2: raster too large for one launch (%lld blocks)",
                  (long long)nblocks);
        return ND_AMD_EUNSUPPORTED;
    }
    // returns true when the sample was taken (the kernels launched next may then be gated)
    auto take_sample = [&]() -> bool {
        const int64_t sblocks_total = ceil_div(npix, (int64_t)kRetainThreads);
        if (!gate_env || !retain || sblocks_total < 4096) return false;
        OmniSampleArgs<T> sa;
        sa.c11 = g.c11;
        sa.c12r = g.c12r;
        sa.c12i = g.c12i;
        sa.c22 = g.c22;
        sa.nx = g.nx;
        sa.blocks_per_row = ceil_div(g.nx, (int64_t)kRetainThreads);
        const int64_t total = sa.blocks_per_row * g.nrows;
        // (pixel-major: a lane's dates are 4-byte loads k x 4 bytes apart from its neighbour's, 64 lines per
        //  instruction -- the sample's time is its line requests: 128 blocks, 32 768 pixels, are plenty)
        const int64_t nsb = pm_ids ? 128 : 512;
        sa.block_stride = total / nsb;
        sa.sy = sy;
        sa.sx = sx;
        sa.st = st;
        sa.m11 = pm_ids ? (int)pm_ids[0] : 1;
        sa.m12 = pm_ids ? (int)pm_ids[1] : 1;
        sa.m22 = pm_ids ? (int)pm_ids[3] : 1;
        sa.k = (int)k;
        sa.nlooks = g.nlooks;
        sa.e = g.e;
        sa.gate_out = flag_count + 2;
        g.gate_n = (uint32_t)(nsb * kRetainThreads);
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_SAMPLE, stream);
        hipLaunchKernelGGL((omnibus_c2_sample_kernel<T>), dim3((unsigned)nsb), dim3(kRetainThreads), 0,
                           stream, sa);
        return true;
    };
    // ---- the two kernels behind pass A, on the lists of the whole raster or of one row slab ----
    auto launch_dense = [&](hipStream_t sq, uint32_t *count, uint32_t *idx, uint32_t *dense_idx, T *dump,
                            uint32_t seg, uint32_t segd, uint32_t dump_cap, int64_t npix_listed) {
        OmniDenseArgs<T> d;
        d.c11 = g.c11;
        d.c12r = g.c12r;
        d.c12i = g.c12i;
        d.c22 = g.c22;
        d.nx = nx;
        d.npix = npix;
        d.sy = sy;
        d.sx = sx;
        d.st = st;
        d.k = (int)k;
        d.flat = flat ? 1 : 0;
        d.nlooks = g.nlooks;
        d.alpha = alpha;
        d.change = change;
        d.flag_count = count;
        d.dense_idx = dense_idx;
        d.segd = segd;
        d.tab = tab_dev;
        d.flag_idx = idx;
        d.seg = seg;
        d.dump = dump;
        d.dump_cap = dump_cap;
        int64_t per_shard_d = ceil_div(ceil_div(npix_listed, (int64_t)kShards), 64);
        if (per_shard_d > 32) per_shard_d = 32;     // 4096 waves: two rounds of what the chip holds
        if (per_shard_d < 1) per_shard_d = 1;
        const dim3 gridd((unsigned)(per_shard_d * kShards)), blockd(64);
        const DenseScreen scr = make_dense_screen<T>(htab, (int)k, n_looks);
        const StreamScreen<32> ssd = make_stream_screen<T, 32>(htab, scr, (int)k, n_looks);
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_DENSE, sq);
        if (k <= 8)
            hipLaunchKernelGGL((omnibus_c2_dense_kernel<T, 8>), gridd, blockd, 0, sq, d, ssd);
        else if (k <= 16)
            hipLaunchKernelGGL((omnibus_c2_dense_kernel<T, 16>), gridd, blockd, 0, sq, d, ssd);
        else if (sizeof(T) == 4 && k <= 24)
            hipLaunchKernelGGL((omnibus_c2_dense_kernel<float, 24>), gridd, blockd, 0, sq,
                               reinterpret_cast<const OmniDenseArgs<float> &>(d), ssd);
        else if (sizeof(T) == 4)
            hipLaunchKernelGGL((omnibus_c2_dense_kernel<float, 32>), gridd, blockd, 0, sq,
                               reinterpret_cast<const OmniDenseArgs<float> &>(d), ssd);
    };
    const bool low_threshold = fused || stream_long || ml_chain || (pm_ids != nullptr && alpha < fused_alpha && !exact_flags);
    auto launch_search = [&](hipStream_t sq, const uint32_t *count, const uint32_t *idx, const T *dump,
                             uint32_t seg, uint32_t dump_cap, int64_t npix_listed,
                             unsigned long long *hand, uint32_t hand_words) -> int {
        OmniSearchArgs<T> s;
        s.c11 = g.c11;
        s.c12r = g.c12r;
        s.c12i = g.c12i;
        s.c22 = g.c22;
        s.nx = nx;
        s.sy = sy;
        s.sx = sx;
        s.st = st;
        s.m11 = pm_ids ? (int)pm_ids[0] : 1;
        s.m12 = pm_ids ? (int)pm_ids[1] : 1;
        s.m22 = pm_ids ? (int)pm_ids[3] : 1;
        s.k = (int)k;
        s.nlooks = g.nlooks;
        s.alpha = alpha;
        s.change = change;
        s.flag_count = count;
        s.flag_idx = idx;
        s.seg = seg;
        s.tab = tab_dev;
        s.dump = dump;
        s.dump_cap = dump_cap;
        s.hand_bits = nullptr;
        s.hand_words = hand_words;
        s.hand_count = const_cast<uint32_t *>(count) + 3;    // word 3 of the lists' first counter line
        s.starts_max = 0;
        s.first = 0;
        const size_t scr_bytes = (size_t)(k + 1) * 4 * sizeof(double);      // screen constants
        const size_t lds_bytes = (size_t)k * 4 * 64 * sizeof(T) + scr_bytes;
        // the LDS image of 64 series may take up to 150 KB of the CU's 160 KB: for long series that is
        // one or two waves per CU, still well ahead of a dependent plane access per date and lane
        const bool use_lds = lds_bytes <= 150 * 1024;
        // the float32 screen of the sweep (FS): its per-j band, for tests over up to 192 dates
        const bool fs = search_fs_enabled() && k <= kScreenLong;
        DenseScreenLong fscr;
        memset(&fscr, 0, sizeof(fscr));
        for (int j = 0; j <= kScreenLong; ++j) {
            fscr.e[j].a = -INFINITY;
            fscr.e[j].b = INFINITY;
            if (fs && j >= 1 && j <= (int)k) fscr.e[j] = make_dense_entry<T>(htab[(size_t)j], j, n_looks);
        }
        if (use_lds && lds_bytes > 64 * 1024) {
            ND_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&omnibus_c2_search_kernel<T, 0, 64, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            ND_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&omnibus_c2_search_kernel<T, 0, 64, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        }
// the sweep kernel with or without the float32 screen
#define ND_LAUNCH_SWEEP(MODE_, PXW_, GRID_, LDS_)                                                                  \
    do {                                                                                                          \
        if (fs)                                                                                                   \
            hipLaunchKernelGGL((omnibus_c2_search_kernel<T, MODE_, PXW_, true>), GRID_, dim3(64), LDS_, sq, s, fscr);  \
        else                                                                                                      \
            hipLaunchKernelGGL((omnibus_c2_search_kernel<T, MODE_, PXW_, false>), GRID_, dim3(64), LDS_, sq, s, fscr); \
    } while (0)
        // kShards x (blocks per shard); a shard's blocks stride through its list
        int64_t per_shard = ceil_div(ceil_div(npix_listed, kShards), 64);
        if (per_shard > 64) per_shard = 64;
        if (per_shard < 1) per_shard = 1;
        const int64_t sblocks = per_shard * kShards;
        static const int mode_env = [] {
            const char *e = getenv("ND_AMD_SEARCH_MODE");     // 0 LDS image, 1 from memory, 3 chain form
            return e ? atoi(e) : -1;
        }();
        // The register form serves the series lengths of dense_search, as long as the screen can
        // decide tests at all (it cannot where omega2 leaves [0, 1], e.g. single-look data: every
        // test would cost a second walk of its segment).
        bool regs_ok = k >= 2 && k <= (sizeof(T) == 4 ? 32 : 16);
        DenseScreen scr;
        if (regs_ok) {
            scr = make_dense_screen<T>(htab, (int)k, n_looks);
            for (int j = 2; j <= (int)k; ++j)
                if (!(scr.e[j].a > -INFINITY) && !(scr.e[j].b < INFINITY)) regs_ok = false;
        }
        // Where the search was fused into pass A (low thresholds), what reaches pass B are the few
        // pixels its screen could not decide, and in that regime nearly every date of a pixel is a
        // change: two dozen short segments, i.e. two dozen rounds of the register form against ~70
        // date steps of the LDS form (measured at alpha = 0.01: 0.166 against 0.125 ms).
        // Behind the fused search: short lists (what its screen could not decide: a few thousand
        // pixels, nearly every date of them a change) one lane per segment start; long ones (the
        // candidates of sparse waves, few changes each) in the register form.  The choice is made
        // per shard, on the device.
        static const int starts_env = [] {
            const char *e = getenv("ND_AMD_SEARCH_STARTS");      // list length per shard up to which ...; 0 = off
            return e ? atoi(e) : 512;
        }();
        const bool starts_form = low_threshold && k >= 2 && k <= 193 && starts_env > 0 && mode_env < 0;
        if (starts_form) {
            s.starts_max = (uint32_t)starts_env;
            KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_SEARCH, sq);
            hipLaunchKernelGGL((omnibus_c2_search_starts_kernel<T>), dim3((unsigned)sblocks), dim3(64), 0, sq, s);
        }
        // ND_AMD_SEARCH_MODE: 0 LDS image, 1 from memory, 3 chain form (the default where it exists)
        const bool chain_form = regs_ok && (mode_env == 3 || mode_env < 0);
        if (chain_form) {
            KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_SEARCH, sq);
            s.hand_bits = hand;
            const StreamScreen<32> ss0 = make_stream_screen<T, 32>(htab, scr, (int)k, n_looks);
            const dim3 gr((unsigned)sblocks), bl(64);
            if (k <= 8)
                hipLaunchKernelGGL((omnibus_c2_search_chain_kernel<T, 8>), gr, bl, 0, sq, s, ss0);
            else if (k <= 16)
                hipLaunchKernelGGL((omnibus_c2_search_chain_kernel<T, 16>), gr, bl, 0, sq, s, ss0);
            else if (sizeof(T) == 4 && k <= 24)
                hipLaunchKernelGGL((omnibus_c2_search_chain_kernel<float, 24>), gr, bl, 0, sq,
                                   reinterpret_cast<const OmniSearchArgs<float> &>(s), ss0);
            else if (sizeof(T) == 4)
                hipLaunchKernelGGL((omnibus_c2_search_chain_kernel<float, 32>), gr, bl, 0, sq,
                                   reinterpret_cast<const OmniSearchArgs<float> &>(s), ss0);
        }
        // the exact form: every listed pixel, or (behind a register form) the marked ones
        KernelTimer timer((chain_form || starts_form) ? ND_AMD_KERNEL_OMNIBUS_EXACT : ND_AMD_KERNEL_OMNIBUS_SEARCH, sq);
        if (chain_form && k <= 65) {
            // the marked pixels one lane per segment start: their number is small, the time of
            // this step is the dependent chain of one wave
            hipLaunchKernelGGL((omnibus_c2_search_starts_kernel<T>), dim3((unsigned)sblocks), dim3(64), 0, sq, s);
            return ND_AMD_OK;
        }
        // (behind the register form there is usually nothing left: the from-memory form, whose
        // blocks reserve no LDS, and a quarter of the blocks)
        const bool behind = chain_form;
        // (behind the time-split pass A every listed series of a long stack lies in the dump, one run per
        //  pixel: read there date by date, the next date in flight, on all 64 lanes -- the image form runs 16
        //  pixels per wave at these lengths: 96 dates x 8.4 Mpx at alpha = 0.99 0.86 ms against 0.41.  Two
        //  other forms were built, verified and measured slower -- eight dates per lane at a time: 0.56 ms; a
        //  ring of LDS-DMA transfers twelve dates ahead: 0.44 - 0.47 ms -- the sweep is bound by the rate at
        //  which the CU looks up 64 different lines per wave instruction, not by latency or arithmetic:
        //  DESIGN-EXPERIMENTS.md, round 6)
        const bool from_dump = split_ok && dump_cap > 0;
        const int mode = (mode_env == 0 || mode_env == 1) ? mode_env : ((use_lds && !behind && !from_dump) ? 0 : 1);
        const int64_t xblocks = behind ? (per_shard > 16 ? 16 : per_shard) * kShards : sblocks;
        // behind the time-split pass A the dump is BLOCKED (64 series interleaved date by date): the lockstep sweep
        // is its only reader; what the dump could not hold is gathered from the planes by the from-memory form
        // (ND_AMD_SEARCH_MODE / ND_AMD_SEARCH_PXW do not apply to this path)
        if (split_ok && dump_cap > 0) {
            if (!fs) {
                set_error("nd_amd_omnibus_c2: internal: the time-split pass A needs the float32 screen's table (k <= %d)", kScreenLong);
                return ND_AMD_EINVAL;
            }
            hipLaunchKernelGGL((omnibus_c2_search_rounds_kernel<T>), dim3((unsigned)sblocks), dim3(64), (size_t)(k + 1) * sizeof(DenseScreenEntry), sq, s, fscr);
            s.first = dump_cap;
            ND_LAUNCH_SWEEP(1, 64, dim3((unsigned)sblocks), scr_bytes);
            return ND_AMD_OK;
        }
        // images beyond 48 KB (three waves per CU or fewer at 64 series per wave): 16 series per wave
        // (96 dates x 8.4 Mpx at alpha = 0.99: pass B 1.84 ms with 64, 1.35 with 32, 1.30 with 16)
        static const int pxw_env = [] {
            const char *e = getenv("ND_AMD_SEARCH_PXW");        // 64: always full waves; 32 / 16
            return e ? atoi(e) : 0;
        }();
        const int pxw = pxw_env ? pxw_env : (lds_bytes > 48 * 1024 ? 16 : 64);
        if (mode == 0 && use_lds && !behind && (pxw == 32 || pxw == 16)) {
            const size_t lds_n = (size_t)k * 4 * (size_t)pxw * sizeof(T) + scr_bytes;
            int64_t ps = ceil_div(ceil_div(npix_listed, kShards), (int64_t)pxw);
            if (ps > 256) ps = 256;
            if (ps < 1) ps = 1;
            const dim3 gn((unsigned)(ps * kShards));
#define ND_LAUNCH_PXW(P)                                                                                   \
    do {                                                                                                  \
        if (lds_n > 64 * 1024) {                                                                          \
            ND_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&omnibus_c2_search_kernel<T, 0, P, false>), \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_n));    \
            ND_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&omnibus_c2_search_kernel<T, 0, P, true>), \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_n));    \
        }                                                                                                 \
        ND_LAUNCH_SWEEP(0, P, gn, lds_n);                                                                 \
    } while (0)
            if (pxw == 32)
                ND_LAUNCH_PXW(32);
            else
                ND_LAUNCH_PXW(16);      // (8 per wave measured the same: the gather's sector traffic bounds it)
#undef ND_LAUNCH_PXW
            return ND_AMD_OK;
        }
        const dim3 gx((unsigned)xblocks);
        if (mode == 0 && use_lds)
            ND_LAUNCH_SWEEP(0, 64, gx, lds_bytes);
        else
            ND_LAUNCH_SWEEP(1, 64, gx, scr_bytes);
#undef ND_LAUNCH_SWEEP
        return ND_AMD_OK;
    };

    bool gated = false;
    if (mlp != nullptr) {
        if constexpr (std::is_same<T, float>::value) {
            if (g.dump_cap < g.seg) {
                set_error("nd_amd_omnibus_c2_ml: workspace too small for the dump of every listed pixel");
                return ND_AMD_EWORKSPACE;
            }
            if (ml_chain) {
                // (round 6) the rasters from the fused kernel itself: the multilooked series is still in its
                // registers behind the search; ND_AMD_STATS_SPLIT=1: from a pass of their own, as before
                const bool stats_here = stats && !stats_split_env0;
                if (stats && !stats_here) {
                    KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
                    const int rc = launch_ml_pass_a(g, tab, *mlp, nullptr, true, false, stream);
                    if (rc != ND_AMD_OK) return rc;
                    g.z_out = nullptr;
                    g.p_out = nullptr;
                }
                const DenseScreen scr = make_dense_screen<T>(htab, (int)k, n_looks);
                const StreamScreen<32> ss0 = make_stream_screen<T, 32>(htab, scr, (int)k, n_looks);
                KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_FUSED, stream);
                const int rc = launch_ml_pass_a(g, tab, *mlp, &ss0, stats_here, true, stream);
                if (rc != ND_AMD_OK) return rc;
            } else {
                KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
                const int rc = launch_ml_pass_a(g, tab, *mlp, nullptr, stats, true, stream);
                if (rc != ND_AMD_OK) return rc;
            }
        } else {
            set_error("nd_amd_omnibus_c2_ml: float32 only");
            return ND_AMD_EUNSUPPORTED;
        }
    } else if (pm_ids != nullptr && k > (sizeof(T) == 4 ? 24 : 12)) {       // (the pixel-major register forms end there)
        // beyond the register-retaining sizes: LDS images folded in place, in the sparse regime; at low
        // thresholds (nearly every pixel changes) the caller transposes and takes the planar streaming search
        constexpr int VE = 16 / (int)sizeof(T);
        OmniPmDmaArgs<T> dm;
        for (int vi = 0; vi < 4; ++vi) dm.ids[vi] = (int)pm_ids[vi];
        dm.c12_joint = (pm_ids[1] == 2 && pm_ids[2] == 2 &&
                        static_cast<const T *>(c12im) == static_cast<const T *>(c12re) + 1) ? 1 : 0;
        const bool dma_ok = (k % VE) == 0 &&
                            (((uintptr_t)c11 | (uintptr_t)c22 | (uintptr_t)c12re) & 15) == 0 &&
                            (dm.c12_joint || ((uintptr_t)c12im & 15) == 0);
        const int64_t per_px = k * (pm_ids[0] + pm_ids[3] + pm_ids[1] + (dm.c12_joint ? 0 : pm_ids[2]));   // elements
        const int pxw = 64 * per_px * (int64_t)sizeof(T) <= 48 * 1024 ? 64
                        : (32 * per_px * (int64_t)sizeof(T) <= 48 * 1024 ? 32
                           : (16 * per_px * (int64_t)sizeof(T) <= 48 * 1024 ? 16 : 0));
        const bool four_real = pm_ids[0] == 1 && pm_ids[1] == 1 && pm_ids[2] == 1 && pm_ids[3] == 1;
        const bool joint = pm_ids[0] == 1 && pm_ids[3] == 1 && dm.c12_joint;
        if (!dma_ok || pxw == 0 || !(alpha >= fused_alpha || exact_flags) || !(four_real || joint)) {
            set_error("nd_amd_omnibus_c2_pixel_major: %lld dates: beyond the register-retaining sizes only 16-byte "
                      "aligned series of a multiple of %d dates, up to 48 KB per 16 pixels, at alpha >= %g "
                      "(transpose and call nd_amd_omnibus_c2 otherwise)", (long long)k, VE, fused_alpha);
            return ND_AMD_EUNSUPPORTED;
        }
        int off = 0;
        const int order[4] = {0, 1, 2, 3};
        for (int oi = 0; oi < 4; ++oi) {
            const int vi = order[oi];
            dm.img_off[vi] = off;
            if (vi == 2 && dm.c12_joint) {
                dm.img_off[2] = dm.img_off[1];               // shares the image of C12 re
                continue;
            }
            off += pxw * (int)k * dm.ids[vi];
        }
        const size_t lds_long = (size_t)off * sizeof(T);
        g.dump_cap = 0;                          // nothing is dumped: pass B reads the series where they lie
        const dim3 gridw((unsigned)ceil_div(npix, (int64_t)pxw)), blockw(64);
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
#define ND_LAUNCH_PM_LONG(PXW_)                                                                               \
    do {                                                                                                      \
        if (stats_a && joint)                                                                                   \
            hipLaunchKernelGGL((omnibus_c2_pm_long_kernel<T, PXW_, true, true>), gridw, blockw, lds_long, stream, g, tab, dm);   \
        else if (stats_a)                                                                                       \
            hipLaunchKernelGGL((omnibus_c2_pm_long_kernel<T, PXW_, true, false>), gridw, blockw, lds_long, stream, g, tab, dm);  \
        else if (joint)                                                                                       \
            hipLaunchKernelGGL((omnibus_c2_pm_long_kernel<T, PXW_, false, true>), gridw, blockw, lds_long, stream, g, tab, dm);  \
        else                                                                                                  \
            hipLaunchKernelGGL((omnibus_c2_pm_long_kernel<T, PXW_, false, false>), gridw, blockw, lds_long, stream, g, tab, dm); \
    } while (0)
        if (pxw == 64)
            ND_LAUNCH_PM_LONG(64);
        else if (pxw == 32)
            ND_LAUNCH_PM_LONG(32);
        else
            ND_LAUNCH_PM_LONG(16);
#undef ND_LAUNCH_PM_LONG
    } else if (pm_ids != nullptr) {
        if (!flat) {
            set_error("nd_amd_omnibus_c2_pixel_major: the variables must be contiguous");
            return ND_AMD_EUNSUPPORTED;
        }
        OmniPmArgs<T> pm;
        for (int vi = 0; vi < 4; ++vi) {
            pm.ids[vi] = (int)pm_ids[vi];
            const unsigned spp = (unsigned)(k * pm_ids[vi]);
            pm.magic[vi] = (unsigned)((0x100000000ULL + spp - 1) / spp);
        }
        pm.c12_joint = (pm_ids[1] == 2 && pm_ids[2] == 2 &&
                        static_cast<const T *>(c12im) == static_cast<const T *>(c12re) + 1) ? 1 : 0;
        // LDS-DMA form: 16-byte aligned variables, k a multiple of the 16-byte vector, k <= 24
        constexpr int VE = 16 / (int)sizeof(T);
        static const int pm_form = [] {
            const char *e = getenv("ND_AMD_PM_FORM");        // 1 = always the register-staged form
            return e ? atoi(e) : 0;
        }();
        // (round 6: any series length -- the spans of 64 pixels are 16-byte pieces whatever k is; lanes whose runs
        //  are not 16-byte aligned in the image read it element by element, pm_pick.  ND_AMD_PM_ANYK=0: as before,
        //  multiples of the vector only)
        static const bool pm_anyk = [] {
            const char *e = getenv("ND_AMD_PM_ANYK");
            return e ? atoi(e) != 0 : true;
        }();
        const bool kvec = (k % VE) == 0;
        const bool dma_ok = pm_form != 1 && (kvec || pm_anyk) &&
                            (((uintptr_t)c11 | (uintptr_t)c22 | (uintptr_t)c12re) & 15) == 0 &&
                            (pm.c12_joint || ((uintptr_t)c12im & 15) == 0);
        const bool fused_pm = dma_ok && !stats && !exact_flags && k <= 32 && dense_env <= 64 && alpha < fused_alpha;
        if (fused_pm) g.dense_min = dense_env;
        if (dma_ok) {
            OmniPmDmaArgs<T> dm;
            int off = 0;
            for (int vi = 0; vi < 4; ++vi) dm.ids[vi] = (int)pm_ids[vi];
            dm.c12_joint = pm.c12_joint;
            const int order[4] = {0, 3, 1, 2};
            for (int oi = 0; oi < 4; ++oi) {
                const int vi = order[oi];
                dm.img_off[vi] = off;
                if (vi == 2 && dm.c12_joint) continue;      // shares the image of C12 re
                off += 64 * (int)k * dm.ids[vi];
            }
            const size_t lds_dma = (size_t)off * sizeof(T);
            const dim3 gridw((unsigned)ceil_div(npix, 64)), blockw(64);
            if (fused_pm) {
                // low threshold: the streaming search, its dates served from the LDS images --
                // if the sample confirms that the raster is dense
                gated = take_sample();
                const DenseScreen scr = make_dense_screen<T>(htab, (int)k, n_looks);
                const StreamScreen<32> ss0 = make_stream_screen<T, 32>(htab, scr, (int)k, n_looks);
                g.gate_mode = gated ? 1 : 0;
                KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_FUSED, stream);
                // Three forms of the search on pixel-major data, same map.  LDS images (each sector
                // of the input fetched once; 24.6 KB per wave, six waves per CU) and two from-memory
                // forms (no LDS, full occupancy, but a lane's 16-byte pieces make every sector of
                // C11 / C22 cross the L2 two to four times: 23.8 GB of traffic for 6.4 GB of data).
                // Round 2 (140 vector instructions per date): images 4.34 / 3.95 ms at alpha = 0.01 /
                // 1e-4, one piece per step 3.23 / 3.77, sector pairs 3.52 / 3.12.  Round 3 (86 per
                // date, no table in LDS): images 2.39 / 2.05 ms, from memory 3.00 / 3.35 -- with less
                // to issue per date six waves per CU keep up and the single fetch wins.
                // ND_AMD_PM_STREAM_LDS=0 selects the from-memory forms, ND_AMD_PM_STREAM_SECTOR=0/1
                // one of the two.
                static const bool pm_lds = [] {
                    const char *e = getenv("ND_AMD_PM_STREAM_LDS");
                    if (e) return atoi(e) != 0;
                    return getenv("ND_AMD_PM_STREAM_SECTOR") == nullptr;
                }();
                static const int pm_sector_env = [] {
                    const char *e = getenv("ND_AMD_PM_STREAM_SECTOR");
                    return e ? atoi(e) : -1;
                }();
                const bool pm_direct4 = pm_sector_env >= 0 ? pm_sector_env != 0 : alpha <= 1e-3;
                const int fused_form_pm = fused_form_env();
                static const int pm_direct_env = [] {
                    const char *e = getenv("ND_AMD_PM_DIRECT");        // 0: every variable through an LDS image
                    return e ? atoi(e) : 1;
                }();
                // dense_chain behind the staging, C11 / C22 straight into registers and only C12 through
                // LDS: at every threshold below the sparse regime (2.0 ms; the streaming search on full
                // LDS images: 2.05 / 2.37 ms at alpha = 1e-4 / 0.01, dense_chain on full images: 2.5 ms)
                if (fused_form_pm != 0 && k <= 24 && pm_direct_env && pm_ids[0] == 1 && pm_ids[3] == 1 && kvec) {
                    OmniPmDmaArgs<T> dd = dm;
                    dd.img_off[1] = 0;
                    dd.img_off[2] = 64 * (int)k * dd.ids[1];
                    const size_t lds_c12 = (size_t)64 * k * (dd.ids[1] + (dd.c12_joint ? 0 : dd.ids[2])) * sizeof(T);
                    if (k <= 8)
                        hipLaunchKernelGGL((omnibus_c2_pm_dma_kernel<T, 8, false, true, true>), gridw, blockw, lds_c12, stream, g, tab, dd, ss0);
                    else if (k <= 16)
                        hipLaunchKernelGGL((omnibus_c2_pm_dma_kernel<T, 16, false, true, true>), gridw, blockw, lds_c12, stream, g, tab, dd, ss0);
                    else
                        hipLaunchKernelGGL((omnibus_c2_pm_dma_kernel<T, 24, false, true, true>), gridw, blockw, lds_c12, stream, g, tab, dd, ss0);
                } else
                if ((fused_form_pm == 2 || (fused_form_pm != 0 && alpha > 0.02) || !kvec) && k <= 24) {
                    // dense_chain behind the LDS-DMA staging: the form for the thresholds in between (and, at every
                    // low threshold, for the series lengths that are not a multiple of the vector)
                    if (k <= 8)
                        hipLaunchKernelGGL((omnibus_c2_pm_dma_kernel<T, 8, false, true>), gridw, blockw, lds_dma, stream, g, tab, dm, ss0);
                    else if (k <= 16)
                        hipLaunchKernelGGL((omnibus_c2_pm_dma_kernel<T, 16, false, true>), gridw, blockw, lds_dma, stream, g, tab, dm, ss0);
                    else
                        hipLaunchKernelGGL((omnibus_c2_pm_dma_kernel<T, 24, false, true>), gridw, blockw, lds_dma, stream, g, tab, dm, ss0);
                } else
                if (pm_lds)
                    hipLaunchKernelGGL((omnibus_c2_stream_kernel<T, 2, 2>), gridw, blockw, lds_dma, stream, g, tab, dm, ss0);
                else if (pm_direct4 && (k % (2 * VE)) == 0)
                    hipLaunchKernelGGL((omnibus_c2_stream_kernel<T, 2, 4>), dim3((unsigned)ceil_div(npix, (int64_t)kRetainThreads)),
                                       dim3(kRetainThreads), 0, stream, g, tab, dm, ss0);
                else
                    hipLaunchKernelGGL((omnibus_c2_stream_kernel<T, 2, 3>), dim3((unsigned)ceil_div(npix, (int64_t)kRetainThreads)),
                                       dim3(kRetainThreads), 0, stream, g, tab, dm, ss0);
                g.gate_mode = gated ? 2 : 0;
                g.dense_min = 65;                            // the sparse form lists pixel by pixel
            }
            if (!fused_pm || gated) {
            KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
            StreamScreen<32> ss_unused;
            memset(&ss_unused, 0, sizeof(ss_unused));
#define ND_LAUNCH_DMA(KM)                                                                              \
    do {                                                                                              \
        if (stats_a)                                                                                  \
            hipLaunchKernelGGL((omnibus_c2_pm_dma_kernel<T, KM, true>), gridw, blockw, lds_dma, stream, g, tab, dm, ss_unused);  \
        else                                                                                          \
            hipLaunchKernelGGL((omnibus_c2_pm_dma_kernel<T, KM, false>), gridw, blockw, lds_dma, stream, g, tab, dm, ss_unused); \
    } while (0)
            if (k <= 8)
                ND_LAUNCH_DMA(8);
            else if (k <= 16)
                ND_LAUNCH_DMA(16);
            else
                ND_LAUNCH_DMA(24);
#undef ND_LAUNCH_DMA
            }
            g.gate_mode = 0;
        } else {
        const size_t lds = 2 * (size_t)kRetainThreads * (size_t)(k | 1) * sizeof(T);
        if (lds > 64 * 1024) {
            set_error("nd_amd_omnibus_c2_pixel_major: %lld dates of this type do not fit the staging image",
                      (long long)k);
            return ND_AMD_EUNSUPPORTED;
        }
        const dim3 grid((unsigned)nblocks), block(kRetainThreads);
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
#define ND_LAUNCH_PM(KM)                                                                              \
    do {                                                                                              \
        if (stats_a)                                                                                  \
            hipLaunchKernelGGL((omnibus_c2_retain_pm_kernel<T, KM, true>), grid, block, lds, stream, g, tab, pm);  \
        else                                                                                          \
            hipLaunchKernelGGL((omnibus_c2_retain_pm_kernel<T, KM, false>), grid, block, lds, stream, g, tab, pm); \
    } while (0)
        if (k <= 8)
            ND_LAUNCH_PM(8);
        else if (k <= 16)
            ND_LAUNCH_PM(16);
        else if (k <= 24)
            ND_LAUNCH_PM(24);
        else {
            set_error("nd_amd_omnibus_c2_pixel_major: at most 24 dates");
            return ND_AMD_EUNSUPPORTED;
        }
#undef ND_LAUNCH_PM
        }
    } else if (fused) {
        // z / P rasters asked for on top (fused_stats): the chain form evaluates them from the series it
        // retains (one read of the planes; up to round 3 a plain pass A of its own produced them in front
        // of the search: 24 x 2048 x 4096 at alpha = 0.01 1.54 ms against 0.77 without rasters).
        // ND_AMD_STATS_SPLIT=1: the separate pass as before.
        static const bool stats_split_env = [] {
            const char *e = getenv("ND_AMD_STATS_SPLIT");
            return e ? atoi(e) != 0 : false;
        }();
        const bool stats_in_chain = fused_stats && !stats_split_env;
        if (fused_stats && !stats_in_chain) {
            const int dm = g.dense_min;
            const uint32_t cap = g.dump_cap;
            g.dense_min = 65;
            g.dump_cap = 0;             // its candidate lists are not used: no series dumped for them
            {
                KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
                launch_retain<T>(g, tab, nblocks, true, stream);
            }
            ND_HIP_CHECK(hipGetLastError());
            // the search below makes its own lists
            ND_HIP_CHECK(hipMemsetAsync(flag_count, 0, kCounterBytes, stream));
            g.z_out = nullptr;
            g.p_out = nullptr;
            g.dense_min = dm;
            g.dump_cap = cap;
        }
        gated = take_sample();
        const DenseScreen scr = make_dense_screen<T>(htab, (int)k, n_looks);
        g.gate_mode = gated ? 1 : 0;
        {
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_FUSED, stream);
        // Forms of the fused kernel, same map: the streaming one wins while searches beyond three dates
        // are rare (kernel 1.40 - 1.46 ms at alpha = 0.01 against 1.57 for dense_chain, 1.18 against 1.31 at
        // 1e-4); above, dense_chain costs the same at every threshold (1.55 ms) where the streaming
        // search's deep searches take 2.5 ms at 0.05 and 4.6 at 0.2 (24 x 4096^2).
        const int fused_form = fused_form_env();
        // Where the chain form overtakes the streaming one (tools/exp_form_switch.py, float32, 4096^2): the
        // chain search costs the same at every threshold, the streaming search's deep searches grow with it --
        // 8 dates: the chain form wins everywhere (0.38 against 0.40 ms at alpha = 1e-4, 0.42 against 0.52 at
        // 0.01); 16: from 0.002; 24: from 0.007 (at the reference's default 0.01: 1.32 against 1.41 ms of kernel
        // and 6.9 against 8.3 GB of traffic; at 0.02: 1.30 against 1.69).  Series that do not fill their
        // instantiation (the chain form walks KMAX dates), 32 dates (two waves per SIMD) and float64 keep 0.02.
        const double chain_alpha = sizeof(T) == 4 ? (k == 8 ? 0.0 : (k == 16 ? 0.002 : (k == 24 ? 0.007 : 0.02))) : 0.02;
        const bool chain_form = stats_in_chain || fused_form == 2 || (fused_form < 0 && alpha > chain_alpha);
        if (chain_form) {
            launch_chain<T>(g, tab, htab, scr, n_looks, nblocks, stream, stats_in_chain);
        } else {
            const dim3 grid((unsigned)nblocks), block(kRetainThreads);
            constexpr int PF = sizeof(T) == 4 ? 6 : 4;
            OmniPmDmaArgs<T> nopm;
            memset(&nopm, 0, sizeof(nopm));
            const StreamScreen<32> ss0 = make_stream_screen<T, 32>(htab, scr, (int)k, n_looks);
            const bool buf1 = g.sx == 1 && (int64_t)g.k * g.st * (int64_t)sizeof(T) < 0x7fffffffLL && g.st >= 0;
            if (buf1)
                hipLaunchKernelGGL((omnibus_c2_stream_kernel<T, PF, 1>), grid, block, 0, stream, g, tab, nopm, ss0);
            else
                hipLaunchKernelGGL((omnibus_c2_stream_kernel<T, PF, 0>), grid, block, 0, stream, g, tab, nopm, ss0);
        }
        }
        if (gated) {
            // the sparse design, should the sample say so (dense waves it still meets go to the
            // separate dense kernel below); with the rasters in the chain form, exactly one of the two
            // kernels runs and writes them
            g.gate_mode = 2;
            KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
            launch_retain<T>(g, tab, nblocks, stats && (!fused_stats || stats_in_chain), stream);
        }
        g.gate_mode = 0;
    } else if (stream_long) {
        const uint32_t cap_keep = g.dump_cap;
        if (stats_split) {
            g.dense_min = 65;
            // (this pass lists 1 - alpha of the pixels at a low threshold; its lists are not used, so no
            // series is dumped for them: at alpha = 0.01 that was a write of the whole stack)
            g.dump_cap = 0;
            {
                KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
                if (retain)
                    launch_retain<T>(g, tab, nblocks, true, stream);
                else if (aligned)
                    launch_global<T, VPPT>(g, tab, nblocks, true, stream);
                else
                    launch_global<T, 1>(g, tab, nblocks, true, stream);
            }
            ND_HIP_CHECK(hipGetLastError());
            // its candidate lists are not used: the search below makes its own
            ND_HIP_CHECK(hipMemsetAsync(flag_count, 0, kCounterBytes, stream));
            g.z_out = nullptr;
            g.p_out = nullptr;
            g.dump_cap = cap_keep;
        }
        g.dense_min = dense_env;
        gated = take_sample();                      // only where the sparse form can retain (k <= 48)
        const DenseScreen scr = make_dense_screen<T>(htab, (int)k, n_looks);
        g.gate_mode = gated ? 1 : 0;
        {
            KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_FUSED, stream);
            // the streaming kernel walks 256-pixel blocks whatever the sparse form would use
            const int64_t bpr_keep = g.blocks_per_row;
            g.blocks_per_row = ceil_div(g.nx, (int64_t)kRetainThreads);
            const int64_t nbs = g.blocks_per_row * g.nrows;
            const dim3 grid((unsigned)nbs), block(kRetainThreads);
            constexpr int PF = sizeof(T) == 4 ? 6 : 4;
            OmniPmDmaArgs<T> nopm;
            memset(&nopm, 0, sizeof(nopm));
            const bool buf = g.sx == 1 && (int64_t)g.k * g.st * (int64_t)sizeof(T) < 0x7fffffffLL && g.st >= 0;
            const int fused_form_long = fused_form_env();
            // float64 series of 17 .. 24 dates still fit the registers (as 32 float32 dates do): the
            // chain form.  (33 .. 48 float32 dates: that instantiation spilled 2 KB per lane under the
            // 256-register cap of two waves per SIMD and was no faster than the streaming search with
            // its deep searches -- 6.9 against 6.6 ms at alpha = 0.2 on 48 x 2048 x 4096; deleted in round 4.)
            if (retain && sizeof(T) == 8 && (stats_long_chain || fused_form_long == 2 || (fused_form_long < 0 && alpha > 0.02))) {
                launch_chain<T>(g, tab, htab, scr, n_looks, nblocks, stream, stats_long_chain);
            } else if (k > kDenseMax) {
                // 129 .. 192 dates: the two-pass chain search at every low threshold (three-word masks; the constants
                // of the tests from make_dense_entry directly: DenseScreen ends at 128)
                StreamScreen<kStreamChainMax> ss3 = make_stream_screen<T, kDenseMax>(htab, scr, (int)k, n_looks).template widen<kStreamChainMax>();
                for (int j = kDenseMax + 1; j <= kStreamChainMax; ++j) {
                    DenseScreenEntry de;
                    de.re = 0;
                    de.rf = 0.f;
                    de.a = -INFINITY;
                    de.b = INFINITY;
                    if (j <= (int)k) de = make_dense_entry<T>(htab[(size_t)j], j, n_looks);
                    ss3.e[j].re = de.re;
                    ss3.e[j].rf = de.rf;
                    ss3.e[j].a = de.a;
                    ss3.e[j].b = de.b;
                    ss3.e[j].jf = (float)j;
                    ss3.e[j].cj = (sizeof(T) == 4 ? 5.9604645e-08f : 1.1102230e-16f) * 7.5f * (float)j;
                    ss3.e[j].mj = (float)j * 1.01f;
                    ss3.e[j].pad = 0.f;
                }
                if (stats_long_chain) {
                    if (buf) hipLaunchKernelGGL((omnibus_c2_stream_chain_kernel<T, PF, 1, 3, true>), grid, block, 0, stream, g, tab, ss3);
                    else hipLaunchKernelGGL((omnibus_c2_stream_chain_kernel<T, PF, 0, 3, true>), grid, block, 0, stream, g, tab, ss3);
                } else {
                    if (buf) hipLaunchKernelGGL((omnibus_c2_stream_chain_kernel<T, PF, 1, 3, false>), grid, block, 0, stream, g, tab, ss3);
                    else hipLaunchKernelGGL((omnibus_c2_stream_chain_kernel<T, PF, 0, 3, false>), grid, block, 0, stream, g, tab, ss3);
                }
            } else if (stats_long_chain || fused_form_long == 3 || (fused_form_long < 0 && alpha > 0.02)) {
                // longer series between the streaming search's thresholds and the sparse regime: the
                // chain search in two streaming passes (ND_AMD_FUSED_FORM=3 forces it, 0 the one-pass form)
#define ND_LAUNCH_SCHAIN(MODE_, MW_, NJ_)                                                                         \
    do {                                                                                                          \
        if (stats_long_chain)                                                                                     \
            hipLaunchKernelGGL((omnibus_c2_stream_chain_kernel<T, PF, MODE_, MW_, true>), grid, block, 0, stream, g, tab, \
                               make_stream_screen<T, NJ_>(htab, scr, (int)k, n_looks));                           \
        else                                                                                                      \
            hipLaunchKernelGGL((omnibus_c2_stream_chain_kernel<T, PF, MODE_, MW_, false>), grid, block, 0, stream, g, tab, \
                               make_stream_screen<T, NJ_>(htab, scr, (int)k, n_looks));                           \
    } while (0)
                if (k <= 32) {
                    if (buf) ND_LAUNCH_SCHAIN(1, 0, 32);
                    else ND_LAUNCH_SCHAIN(0, 0, 32);
                } else if (k <= 64) {
                    if (buf) ND_LAUNCH_SCHAIN(1, 1, 64);
                    else ND_LAUNCH_SCHAIN(0, 1, 64);
                } else {
                    if (buf) ND_LAUNCH_SCHAIN(1, 2, 128);
                    else ND_LAUNCH_SCHAIN(0, 2, 128);
                }
#undef ND_LAUNCH_SCHAIN
            } else
            if (k <= 32) {
                if (buf) hipLaunchKernelGGL((omnibus_c2_stream_kernel<T, PF, 1, 0>), grid, block, 0, stream, g, tab, nopm, make_stream_screen<T, 32>(htab, scr, (int)k, n_looks));
                else hipLaunchKernelGGL((omnibus_c2_stream_kernel<T, PF, 0, 0>), grid, block, 0, stream, g, tab, nopm, make_stream_screen<T, 32>(htab, scr, (int)k, n_looks));
            } else if (k <= 64) {
                if (buf) hipLaunchKernelGGL((omnibus_c2_stream_kernel<T, PF, 1, 1>), grid, block, 0, stream, g, tab, nopm, make_stream_screen<T, 64>(htab, scr, (int)k, n_looks));
                else hipLaunchKernelGGL((omnibus_c2_stream_kernel<T, PF, 0, 1>), grid, block, 0, stream, g, tab, nopm, make_stream_screen<T, 64>(htab, scr, (int)k, n_looks));
            } else {
                if (buf) hipLaunchKernelGGL((omnibus_c2_stream_kernel<T, PF, 1, 2>), grid, block, 0, stream, g, tab, nopm, make_stream_screen<T, 128>(htab, scr, (int)k, n_looks));
                else hipLaunchKernelGGL((omnibus_c2_stream_kernel<T, PF, 0, 2>), grid, block, 0, stream, g, tab, nopm, make_stream_screen<T, 128>(htab, scr, (int)k, n_looks));
            }
            g.blocks_per_row = bpr_keep;
        }
        g.dense_min = 65;                           // no register search at these lengths
        if (gated) {
            g.gate_mode = 2;
            KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
            launch_retain<T>(g, tab, nblocks, stats && !stats_split, stream);
        }
        g.gate_mode = 0;
    } else if (split_ok) {
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
        g.blocks_per_row = ceil_div(g.nx, (int64_t)64);
        const int64_t nbs = g.blocks_per_row * g.nrows;
        if (nbs > 0x7fffffffLL) {
            set_error("nd_amd_omnibus_c2: raster too large for one launch (%lld blocks)", (long long)nbs);
            return ND_AMD_EUNSUPPORTED;
        }
        // rounding band of the re-associated sums: 3 (5 k + 8) u (see the kernel), u = half an ulp of T
        const float rel = 3.f * (5.f * (float)k + 8.f) * (sizeof(T) == 4 ? 5.9604645e-08f : 1.1102230e-16f);
        const dim3 grid((unsigned)nbs);
#define ND_LAUNCH_SPLIT(TT_, KQ_, NS_)                                                                         \
    do {                                                                                                       \
        if (stats)                                                                                             \
            hipLaunchKernelGGL((omnibus_c2_split_kernel<TT_, KQ_, NS_, true>), grid, dim3(64 * NS_), 0, stream, \
                               reinterpret_cast<const OmniGlobalArgs<TT_> &>(g), tab, rel);                     \
        else                                                                                                   \
            hipLaunchKernelGGL((omnibus_c2_split_kernel<TT_, KQ_, NS_, false>), grid, dim3(64 * NS_), 0, stream, \
                               reinterpret_cast<const OmniGlobalArgs<TT_> &>(g), tab, rel);                     \
    } while (0)
        if constexpr (std::is_same<T, float>::value) {
            // 4 KQ registers per lane: four waves up to 96 dates, eight beyond
            const int ns = k <= 96 ? 4 : 8;
            const int kqc = (int)ceil_div(k, ns);
            const int kq = kqc <= 16 ? 16 : (kqc <= 20 ? 20 : 24);
            if (ns == 4) {
                if (kq <= 16) ND_LAUNCH_SPLIT(float, 16, 4);
                else if (kq == 20) ND_LAUNCH_SPLIT(float, 20, 4);
                else ND_LAUNCH_SPLIT(float, 24, 4);
            } else {
                if (kq == 16) ND_LAUNCH_SPLIT(float, 16, 8);
                else if (kq == 20) ND_LAUNCH_SPLIT(float, 20, 8);
                else ND_LAUNCH_SPLIT(float, 24, 8);
            }
        } else {
            // float64 (25 .. 96 dates): 8 KQ registers per lane -- 8 or 12 dates per wave
            if (k <= 32) ND_LAUNCH_SPLIT(double, 8, 4);
            else if (k <= 48) ND_LAUNCH_SPLIT(double, 12, 4);
            else if (k <= 64) ND_LAUNCH_SPLIT(double, 8, 8);
            else ND_LAUNCH_SPLIT(double, 12, 8);
        }
#undef ND_LAUNCH_SPLIT
    } else {
        KernelTimer timer(ND_AMD_KERNEL_OMNIBUS_GLOBAL, stream);
        if (retain)
            launch_retain<T>(g, tab, nblocks, stats || exact_flags, stream);
        else if (aligned)
            launch_global<T, VPPT>(g, tab, nblocks, stats || exact_flags, stream);
        else
            launch_global<T, 1>(g, tab, nblocks, stats || exact_flags, stream);
    }
    ND_HIP_CHECK(hipGetLastError());

    // ---- dense waves (none in the sparse regime: every block then leaves at once) ----
    if (retain && g.dense_min <= 64 && (!fused || gated) && pm_ids == nullptr)
        launch_dense(stream, flag_count, flag_idx, g.dense_idx, g.dump, g.seg, g.segd, g.dump_cap, npix);
    ND_HIP_CHECK(hipGetLastError());

    // ---- pass B ----
    {
        const int rc = launch_search(stream, flag_count, flag_idx, g.dump, g.seg, g.dump_cap, npix,
                                     hand_ws, g.seg / 64 + 1);
        if (rc != ND_AMD_OK) return rc;
    }
    ND_HIP_CHECK(hipGetLastError());
    return ND_AMD_OK;
}

}  // namespace nd_amd

using namespace nd_amd;

extern "C" size_t nd_amd_omnibus_c2_workspace_bytes(int dtype, int64_t ny, int64_t nx, int64_t k,
                                                    size_t *min_bytes)
{
    if (ny < 0 || nx < 0 || k < 0) return 0;
    const OmniWorkspace w = omni_layout(ny * nx, ny, k, dtype == ND_AMD_F64 ? 8 : 4);
    if (min_bytes) *min_bytes = w.min_total;
    return w.recommended;
}

extern "C" int nd_amd_omnibus_c2(const void *c11, const void *c12re, const void *c12im,
                                 const void *c22, int dtype, int64_t ny, int64_t nx, int64_t k,
                                 int64_t stride_y, int64_t stride_x, int64_t stride_t,
                                 uint32_t n_looks, double alpha, uint8_t *change, void *z_out,
                                 void *p_out, void *workspace, size_t workspace_bytes,
                                 void *hip_stream)
{
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_omnibus_c2: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (ny < 0 || nx < 0 || k < 0) {
        set_error("nd_amd_omnibus_c2: negative shape (%lld, %lld, %lld)", (long long)ny,
                  (long long)nx, (long long)k);
        return ND_AMD_EINVAL;
    }
    if (ny == 0 || nx == 0 || k == 0) return ND_AMD_OK;   // nothing to write
    if (!c11 || !c12re || !c12im || !c22 || !change) {
        set_error("nd_amd_omnibus_c2: null data pointer");
        return ND_AMD_EINVAL;
    }
    if (n_looks == 0) {
        set_error("nd_amd_omnibus_c2: n_looks must be >= 1");
        return ND_AMD_EINVAL;
    }
    if (ny * nx >= 0xffffffffLL || k > 0x3fffffff) {
        set_error("nd_amd_omnibus_c2: raster of %lld pixels exceeds the 32-bit pixel index",
                  (long long)(ny * nx));
        return ND_AMD_EUNSUPPORTED;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (dtype == ND_AMD_F32)
        return omnibus_c2_impl<float>(c11, c12re, c12im, c22, ny, nx, k, stride_y, stride_x,
                                      stride_t, n_looks, alpha, change, z_out, p_out, workspace,
                                      workspace_bytes, stream);
    return omnibus_c2_impl<double>(c11, c12re, c12im, c22, ny, nx, k, stride_y, stride_x,
                                   stride_t, n_looks, alpha, change, z_out, p_out, workspace,
                                   workspace_bytes, stream);
}

extern "C" size_t nd_amd_omnibus_c2_ml_workspace_bytes(int dtype, int64_t ny, int64_t nx, int64_t k, int ml)
{
    OmniMlPlan p;
    if (ny <= 0 || nx <= 0 || k <= 0 || !omni_ml_plan(ny, nx, k, nx, 1, ny * nx, ml, dtype, &p)) return 0;
    return omni_layout(ny * nx, ny, k, 4, p.seg).min_total;
}

extern "C" int nd_amd_omnibus_c2_ml(const void *c11, const void *c12re, const void *c12im, const void *c22,
                                    int dtype, int64_t ny, int64_t nx, int64_t k, int64_t stride_y,
                                    int64_t stride_x, int64_t stride_t, int ml, double alpha,
                                    uint8_t *change, void *z_out, void *p_out, void *workspace,
                                    size_t workspace_bytes, void *hip_stream)
{
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_omnibus_c2_ml: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (ny < 0 || nx < 0 || k < 0 || ml < 1) {
        set_error("nd_amd_omnibus_c2_ml: bad shape or window (%lld, %lld, %lld; ml = %d)", (long long)ny,
                  (long long)nx, (long long)k, ml);
        return ND_AMD_EINVAL;
    }
    if (ny == 0 || nx == 0 || k == 0) return ND_AMD_OK;
    if (!c11 || !c12re || !c12im || !c22 || !change) {
        set_error("nd_amd_omnibus_c2_ml: null data pointer");
        return ND_AMD_EINVAL;
    }
    OmniMlPlan p;
    if (ny * nx >= 0xffffffffLL ||
        !omni_ml_plan(ny, nx, k, stride_y, stride_x, stride_t, ml, dtype, &p)) {
        set_error("nd_amd_omnibus_c2_ml: fused multilooking covers float32, x-contiguous planes, ml = 3 or 5, "
                  "2 <= k <= 24 (multilook with nd_amd_correlate and call nd_amd_omnibus_c2 otherwise)");
        return ND_AMD_EUNSUPPORTED;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    return omnibus_c2_impl<float>(c11, c12re, c12im, c22, ny, nx, k, stride_y, stride_x, stride_t,
                                  (uint32_t)(ml * ml), alpha, change, z_out, p_out, workspace,
                                  workspace_bytes, stream, nullptr, &p);
}

extern "C" int nd_amd_omnibus_c2_pixel_major(const void *c11, const void *c12re, const void *c12im,
                                             const void *c22, int dtype, int64_t ny, int64_t nx,
                                             int64_t k, const int64_t date_stride[4],
                                             uint32_t n_looks, double alpha, uint8_t *change,
                                             void *z_out, void *p_out, void *workspace,
                                             size_t workspace_bytes, void *hip_stream)
{
    if (dtype != ND_AMD_F32 && dtype != ND_AMD_F64) {
        set_error("nd_amd_omnibus_c2_pixel_major: dtype must be ND_AMD_F32 or ND_AMD_F64, got %d", dtype);
        return ND_AMD_EINVAL;
    }
    if (ny < 0 || nx < 0 || k < 0 || !date_stride) {
        set_error("nd_amd_omnibus_c2_pixel_major: bad shape");
        return ND_AMD_EINVAL;
    }
    for (int vi = 0; vi < 4; ++vi)
        if (date_stride[vi] != 1 && date_stride[vi] != 2) {
            set_error("nd_amd_omnibus_c2_pixel_major: date strides must be 1 or 2");
            return ND_AMD_EINVAL;
        }
    if (date_stride[1] != date_stride[2]) {
        set_error("nd_amd_omnibus_c2_pixel_major: the two halves of C12 must share their stride");
        return ND_AMD_EINVAL;
    }
    if (ny == 0 || nx == 0 || k == 0) return ND_AMD_OK;
    if (!c11 || !c12re || !c12im || !c22 || !change) {
        set_error("nd_amd_omnibus_c2_pixel_major: null data pointer");
        return ND_AMD_EINVAL;
    }
    if (n_looks == 0) {
        set_error("nd_amd_omnibus_c2_pixel_major: n_looks must be >= 1");
        return ND_AMD_EINVAL;
    }
    if (ny * nx >= 0xffffffffLL || k > 192) {
        set_error("nd_amd_omnibus_c2_pixel_major: at most 192 dates and 2^32 - 1 pixels");
        return ND_AMD_EUNSUPPORTED;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    // unit-stride geometry of the layout: (y, x, t) -> (y * nx + x) * k + t
    if (dtype == ND_AMD_F32)
        return omnibus_c2_impl<float>(c11, c12re, c12im, c22, ny, nx, k, nx * k, k, 1, n_looks, alpha,
                                      change, z_out, p_out, workspace, workspace_bytes, stream,
                                      date_stride);
    return omnibus_c2_impl<double>(c11, c12re, c12im, c22, ny, nx, k, nx * k, k, 1, n_looks, alpha,
                                   change, z_out, p_out, workspace, workspace_bytes, stream,
                                   date_stride);
}
