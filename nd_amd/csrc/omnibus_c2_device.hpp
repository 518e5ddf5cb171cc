// nd_amd/csrc/omnibus_c2_device.hpp -- device-side pieces of the dual-pol omnibus test shared by
// omnibus.hip and omnibus_ml.hip: the reference's running state (nd/_change.pyx:53-77), pass A's
// argument block, the wave-level zero-fill of the change map and the two-pass search on a series
// held in registers (dense_chain).  Moved here verbatim from omnibus.hip (round 4) so that the
// multilooking front end can live in a translation unit of its own.
#pragma once
#include <type_traits>

#include "omnibus_common.hpp"

namespace nd_amd {

// ---- the reference's running state (nd/_change.pyx:53-69) -------------------------------
template <typename T>
struct Accum {
    T s11, s12r, s12i, s22;
    double prod;
    __device__ __forceinline__ void reset()
    {
        s11 = 0;
        s12r = 0;
        s12i = 0;
        s22 = 0;
        prod = 1.0;
    }
    __device__ __forceinline__ void step(T a, T b, T c, T d)
    {
        const T det = (a * d) - ((b * b) + (c * c));
        prod = prod * (double)det;
        s11 = s11 + a;
        s12r = s12r + b;
        s12i = s12i + c;
        s22 = s22 + d;
    }
};

// z = -2 rho ln Q over j matrices (nd/_change.pyx:72-76)
template <typename T>
__device__ __forceinline__ T z_stat(const Accum<T> &A, int j, double nlooks, const OmniTabEntry &e)
{
    const T det_of_sum = (A.s11 * A.s22) - ((A.s12r * A.s12r) + (A.s12i * A.s12i));
    const double logQ =
        nlooks * ((e.pklogk + log(A.prod)) - ((double)j * log((double)det_of_sum)));
    return (T)(e.m2rho * logQ);
}

// Screen statistic: z from approx_ln.  |z_approx - z| <= |m2rho| n (k+1) 1e-7; the host widens
// zlo_a / zhi_a by ten times that, so z_approx < zlo_a implies z < zlo (omni tables).
template <typename T>
__device__ __forceinline__ double z_approx(const Accum<T> &A, int j, double nlooks, double m2rho,
                                           double pklogk)
{
    const T det_of_sum = (A.s11 * A.s22) - ((A.s12r * A.s12r) + (A.s12i * A.s12i));
    const double logQ = nlooks * ((pklogk + approx_ln(A.prod)) -
                                  ((double)j * approx_ln((double)det_of_sum)));
    return m2rho * logQ;
}
template <typename T>
__device__ __forceinline__ double z_approx(const Accum<T> &A, int j, double nlooks,
                                           const OmniTabEntry &e)
{
    return z_approx<T>(A, j, nlooks, e.m2rho, e.pklogk);
}

// The input planes are read exactly once and the change map is written once: both bypass the
// cache hierarchy's retention ("nt" = aux bit 1 on gfx950 buffer ops, __builtin_nontemporal_* on
// plain accesses).  Measured on pass A: 1.30 -> 1.14 ms (profiles/r01_probe_bandwidth.txt).
constexpr int kNtAux = 2;

__device__ __forceinline__ void store_zero16_nt(uint4 *p)
{
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 z = {0u, 0u, 0u, 0u};
    __builtin_nontemporal_store(z, reinterpret_cast<u4 *>(p));
}

// raw buffer load of one element: descriptor (SGPRs) + lane byte offset + scalar byte offset
template <typename T>
__device__ __forceinline__ T buffer_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ float buffer_load<float>(__amdgpu_buffer_rsrc_t rsrc, unsigned voff,
                                                    unsigned soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, kNtAux));
}
template <>
__device__ __forceinline__ double buffer_load<double>(__amdgpu_buffer_rsrc_t rsrc, unsigned voff,
                                                      unsigned soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, kNtAux));
}

template <typename T, int N>
struct alignas(sizeof(T) * N) Pack {
    T v[N];
};

// =========================================================================================
// pass A
// =========================================================================================
template <typename T>
struct OmniGlobalArgs {
    const T *c11, *c12r, *c12i, *c22;
    int64_t nx, nrows;        // pixels per row, rows (flattened to one row when planes are contiguous)
    int64_t sy, sx, st;       // element strides
    int64_t blocks_per_row;
    int k;
    int write_tab;            // block 0 copies `tab` into tab_dev
    double nlooks, alpha;
    OmniTabEntry e;           // constants of the test over all k matrices
    uint8_t *change;
    T *z_out, *p_out;
    // The list of pixels whose global test can fire is kept in kShards independent segments
    // (shard = block index mod kShards), each with its own counter on its own 128-byte line:
    // one counter would serialise ~2e5 wave-level atomics per launch at ~88 per microsecond.
    uint32_t *flag_count;     // [kShards] counters, kCounterStride words apart
    uint32_t *flag_idx;       // [kShards][seg] pixel indices
    uint32_t seg;             // list entries per shard
    OmniTabEntry *tab_dev;
    T *dump;                  // [kShards][dump_cap][date][4] series of the first dump_cap pixels of a shard
    uint32_t dump_cap;
    // Waves in which at least `dense_min` pixels are listed are not listed pixel by pixel: one
    // entry (the pixel index of lane 0) goes to the dense list and omnibus_c2_dense_kernel searches
    // all 64 pixels from registers.  Counter: word 1 of the shard's counter line.
    uint32_t *dense_idx;      // [kShards][segd]
    uint32_t segd;
    int dense_min;            // 65 = never
    // Data-driven choice between the sparse design (this pass + pass B) and the fused search, made
    // on the device: omnibus_c2_sample_kernel counts the candidates among `gate_n` sampled pixels
    // into *gate; both variants are launched and the one the count does not favour returns at
    // once.  gate_mode 0: no gate; 1: run only if the sample is dense; 2: only if it is sparse.
    const uint32_t *gate;
    uint32_t gate_n;
    int gate_mode;
};

// dense <=> at least 1/8 of the sampled pixels pass the global screen (measured break-even of
// pass A + pass B against the fused kernel: ~10 % candidates)
template <typename T>
__device__ __forceinline__ bool omni_gate_skip(const OmniGlobalArgs<T> &g)
{
    if (g.gate_mode == 0) return false;
    const uint32_t hits = __builtin_nontemporal_load(g.gate);
    const bool dense = hits * 8u >= g.gate_n;
    return (g.gate_mode == 1) != dense;
}

constexpr int kGlobalThreads = 256;
#ifndef ND_RETAIN_THREADS
#define ND_RETAIN_THREADS 256
#endif
constexpr int kRetainThreads = ND_RETAIN_THREADS;   // block size of the register-retaining pass A
constexpr int kShards = 128;
constexpr int kCounterStride = 32;   // uint32 words between shard counters (128 B)
#ifndef ND_TIME_CHUNK
#define ND_TIME_CHUNK 4
#endif
constexpr int kTimeChunk = ND_TIME_CHUNK;


__device__ __forceinline__ void zero_fill_span(uint8_t *ob, const int nb, const int lane)
{
    int head = (int)((16 - ((uintptr_t)ob & 15)) & 15);
    if (head > nb) head = nb;
    if (lane < head) ob[lane] = 0;
    const int nvec = (nb - head) >> 4;
    uint4 *vz = reinterpret_cast<uint4 *>(ob + head);
    for (int i = lane; i < nvec; i += 64) store_zero16_nt(vz + i);
    const int tail0 = head + (nvec << 4);
    if (tail0 + lane < nb) ob[tail0 + lane] = 0;
}

typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
typedef __attribute__((address_space(1))) const unsigned char glb_u8_t;

// (series of up to 96 registers: three waves per SIMD; beyond -- 32 float32 / 16 float64 dates and
// more -- two: under the cap of three the search spilled 150 - 230 bytes per lane)
constexpr int chain_waves(const int kmax, const size_t elem) { return kmax * (int)elem <= 96 ? 3 : 2; }
constexpr int chain_nj(const int kmax) { return kmax > 32 ? 64 : 32; }

// -----------------------------------------------------------------------------------------
// The same search in two passes over the registers, linear in k (round 3).
//
// dense_search above walks a triangle: one row per segment start, each row re-adding its dates.
// But single_pixel_change_detection consumes every date ONCE: within a segment the marginal tests
// j = 2, 3, ... extend one running sum date by date, and where one fires the next segment starts at
// that very date (nd/_change.pyx:247-256).  The only thing that looks ahead is the global test of
// ts[l:], asked at each segment start.  So:
//   pass 1 (dates last to first, as in omnibus_c2_stream_kernel): the global test of EVERY start
//           from suffix sums in double, with the rounding band of the reference's forward float
//           sums; two bits per date (fires / cannot fire).
//   pass 2 (dates first to last, date index wave-uniform, every lane busy at every date): each lane
//           carries the reference's own running state of its CURRENT segment -- the four sums in
//           `floating`, started as 0 + a_l, and the double product of the determinants -- adds date
//           t, and decides the marginal test over its j = t - l + 1 dates (bit-identical
//           determinants, tight band; the constants of the lane's own j come from an LDS table).
//           Where it fires: change at t, and if the global test of ts[t:] (pass 1) fires too the
//           state restarts as date t alone; if that test cannot fire the lane is finished; an
//           undecided test of either kind hands the pixel to the exact pass.  At the last date the
//           marginal test IS the global test of the segment, which fired.
// ~105 vector instructions per date in all, whatever the threshold: no rows, no deep searches,
// no divergence.
// -----------------------------------------------------------------------------------------
// MT: unsigned for KMAX <= 32, unsigned long long up to 64 dates
template <typename T, int KMAX, int NJ, typename MT>
__device__ __forceinline__ void dense_chain(const T (&v)[KMAX][4], const int k, const bool active,
                                            const StreamScreen<NJ> &ss, const StreamEntry *tab_lds,
                                            MT &mask_out, bool &handoff_out, bool &cand_out)
{
    static_assert(KMAX <= (int)(8 * sizeof(MT)) && KMAX <= NJ, "mask width");
    // the instantiation serves kmin <= k <= KMAX (8, 16, 24, 32 dates; 48 takes 33 .. 48)
    constexpr int kmin = KMAX == 8 ? 2 : (KMAX == 48 ? 33 : KMAX - 7);
    const T dlo = (T)ss.dlo, dhi = (T)ss.dhi;
    MT gF = 0, gC = 0;
    bool bad = false, dead = false;
    {
        double S11 = 0.0, S12r = 0.0, S12i = 0.0, S22 = 0.0, PP = 1.0;
        int emin = 1, emax = 1;
#pragma unroll
        for (int t = KMAX - 1; t >= 0; --t) {
            // branch-free: elements beyond k hold a copy of a valid date and are masked out
            const bool live = (t < kmin) || (t < k);
            const T a = v[t][0], b = v[t][1], c = v[t][2], d = v[t][3];
            const T det = (a * d) - ((b * b) + (c * c));
            bad = bad | (live & !((a > (T)0) & (det > dlo) & (det < dhi)));
            dead = dead | (live & !((det > (T)0) | (det < (T)0)));
            PP = PP * (live ? (double)det : 1.0);
            S11 += live ? (double)a : 0.0;
            S12r += live ? (double)b : 0.0;
            S12i += live ? (double)c : 0.0;
            S22 += live ? (double)d : 0.0;
            const int jr = k - t;
            const int jj = jr > 0 ? jr : 0;
            const StreamEntry e = ss.e[jj];                  // wave-uniform: one scalar load
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
            const int E = (eP - e.re) - __mul24(jj, es);
            const float x = (float)E + __builtin_fmaf(-e.jf, ms, mP - e.rf);
            const float qq = (float)pp * __builtin_amdgcn_rcpf(df);
            const float rel = e.cj * qq;                     // 1.46 * 5 n u * s11 s22 / det
            const float m2 = e.mj * rel;
            bad = bad | (live & !(okd & (rel < 0.01f)));
            mask_push(gF, x + m2 < e.a);
            mask_push(gC, x - m2 > e.b);
            __builtin_amdgcn_sched_barrier(0);               // one date at a time (registers)
        }
        bad = bad | (emax - emin > 900);
    }
    // KMAX pushes: the bit of date t sits at position t
    MT gI = (MT) ~(gF | gC);
    mask_keep_low(gF, k - 1);
    mask_keep_low(gI, k - 1);
    if (dead) {                                 // a NaN or zero determinant: no change anywhere, no exact pass
        bad = false;
        gF = 0;
        gI = 0;
    }
    cand_out = active && (bad || ((gF | gI) & (MT)1));
    bool handoff = active && (bad || (gI & (MT)1));
    bool done = !active || bad || !(gF & (MT)1) || (gI & (MT)1);
    MT mask = 0;
    // the running state of the segment that starts at date 0 (0 + a_0 = a_0)
    T s11 = v[0][0], s12r = v[0][1], s12i = v[0][2], s22 = v[0][3];
    double PP = (double)((v[0][0] * v[0][3]) - ((v[0][1] * v[0][1]) + (v[0][2] * v[0][2])));
    int j = 1;
#pragma unroll
    for (int t = 1; t < KMAX; ++t) {
        const bool live = (t < kmin) || (t < k);
        const bool last = (t == k - 1);
        T a = v[t][0], b = v[t][1];
        const T c = v[t][2], d = v[t][3];
        // (opaque to the optimiser: it would otherwise keep the 24 determinants of pass 1, and their
        // conversions to double, alive in 72 registers instead of recomputing them here; for the long
        // series also b^2 + c^2, one register per date)
        asm volatile("" : "+v"(a));
        if (KMAX * sizeof(T) > 96) asm volatile("" : "+v"(b));
        const T det = (a * d) - ((b * b) + (c * c));
        s11 = s11 + a;
        s12r = s12r + b;
        s12i = s12i + c;
        s22 = s22 + d;
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
        const bool fires = last | (oks & (x < ca));
        const bool cant = !last & oks & (x > cb);
        const bool act = !done & live;
        const bool und = act & !(fires | cant);
        const bool f = act & fires;
        handoff = handoff | und;
        mask |= f ? ((MT)1 << t) : (MT)0;                    // :252
        // the segment that starts at t (:255): its global test was decided in pass 1
        const bool gi = (gI >> t) & (MT)1, gf = (gF >> t) & (MT)1;
        handoff = handoff | (f & !last & gi);
        done = done | und | (f & (last | gi | !gf));         // :256, :241-242
        s11 = f ? a : s11;
        s12r = f ? b : s12r;
        s12i = f ? c : s12i;
        s22 = f ? d : s22;
        PP = f ? (double)det : PP;
        j = f ? 1 : j;
        __builtin_amdgcn_sched_barrier(0);
    }
    mask_out = mask;
    handoff_out = handoff;
}

// ---- OmnibusTest(ml=w): multilooking fused into pass A (omnibus_ml.hip) -------------------------
constexpr int kMlTileRows = 12;          // rows of a strip = waves of a block
struct OmniMlPlan {
    int ml;
    int64_t ny, nx;
    int strips, xsegs, segw;             // row strips, segments per strip, output columns per segment
    int64_t nblocks;
    uint32_t seg;                        // list (and dump) entries per shard
};
// false: this (shape, strides, window, dtype) is not covered -- the caller multilooks separately
bool omni_ml_plan(int64_t ny, int64_t nx, int64_t k, int64_t sy, int64_t sx, int64_t st, int ml, int dtype,
                  OmniMlPlan *p);
// ss != null: the search fused in (dense_chain); else the sparse design's pass A (stats: with the z / P
// rasters; list = false: rasters only, no candidate list and no zero-fill)
// -> ND_AMD_OK or ND_AMD_EHIP (the dynamic-LDS opt-in of the current device failed)
int launch_ml_pass_a(const OmniGlobalArgs<float> &g, const OmniTab &tab, const OmniMlPlan &p,
                     const StreamScreen<32> *ss, bool stats, bool list, hipStream_t stream);

}  // namespace nd_amd
