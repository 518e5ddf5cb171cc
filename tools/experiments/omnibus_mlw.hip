// tools/experiments/omnibus_mlw.hip -- OmnibusTest(ml=w), multilooking fused into pass A: the WAVE form.
//
// STATUS (round 5): an experiment, NOT part of the library.  Built, verified (tests/test_omnibus_ml_gpu.py: 15
// passed with this form selected; 11 485 fuzz cases, 0 failures) and measured SLOWER than the block form of
// nd_amd/csrc/omnibus_ml.hip: 24 x 4096^2, alpha = 0.99, one box: 2.69 ms (3 x 3) / 4.11 ms (5 x 5) against
// 2.04 / 3.21 ms; 4 waves x 3 slots 2.30 ms; two waves per SIMD 2.67 ms (profiles/r05_ml_experiments.txt,
// DESIGN.md section 5 K1m "Round 5").  Kept as the record of what was tried.  To build it again: copy it to
// nd_amd/csrc/, and restore the three hooks it needs --
//   omnibus_c2_device.hpp, struct OmniMlPlan:   int w_strips, w_xsegs, w_segw;
//   omnibus_ml.hip, omni_ml_plan (behind p->seg):
//       const int W = mlw_waves_per_block();
//       p->w_strips = (int)ceil_div(ny, (int64_t)(2 * W));
//       const int64_t tiles32 = ceil_div(nx, 32);
//       int xs = 1;
//       while ((int64_t)p->w_strips * W * xs < 4 * 3072 && tiles32 / (xs * 2) >= 8) xs *= 2;
//       p->w_segw = (int)(ceil_div(tiles32, xs) * 32);
//       p->w_xsegs = (int)ceil_div(nx, (int64_t)p->w_segw);
//       const int64_t range = (int64_t)p->w_strips * W * p->w_xsegs * (p->w_segw / 32 + 2);
//       const uint32_t wseg = (uint32_t)(ceil_div(range, (int64_t)kShards) * 64 + 64);
//       if (wseg > p->seg) p->seg = wseg;
//   omnibus_ml.hip, launch_ml_pass_a (behind a.list):   if (a.x4) return launch_mlw_pass_a(g, tab, p, a, ss, stats, stream);
//   omnibus_ml_common.hpp:   int mlw_waves_per_block();  int launch_mlw_pass_a(const OmniGlobalArgs<float> &, const OmniTab &,
//       const OmniMlPlan &, const OmniMlArgs &, const StreamScreen<32> *, bool, hipStream_t);
//
// Reference: nd/change.py:61-64 (BoxcarFilter(w=ml) in front of nd._change.change_detection, n = ml**2),
// nd/filters.py:256-267, 294-298 (scipy.ndimage.convolve, ones((w, w)) / w**2, mode 'reflect').  Same
// arithmetic as omnibus_ml.hip (the header there has the details): per output
//     double tmp = 0; for (dy, dx) in row-major window order: tmp += (1 / w^2) * (double)x[y+dy][x+dx]
// product and sum rounded separately, out = (float)tmp; the multilooked value of a (date, variable)
// exists only in registers, the planes are read once.
//
// Why a second form.  The block form (omnibus_ml.hip: twelve waves share a strip of twelve rows and meet at
// one barrier per step of 8 planes) spends 40 % of a resident wave's time waiting: between two barriers
// all twelve waves issue their conversions and additions at the same time (the vector ALUs are the limit
// while that lasts) and then all of them sit in the step's latency chain -- results to LDS, barrier,
// results back, carried columns, transfers, first reads of the next planes -- with the ALUs idle
// (profiles/r04_ml_pmc.txt: VALU busy 43 %).  Here NOTHING is shared between waves, so there is nothing to
// meet for:
//
//   * A wave owns a tile of 2 rows x 32 columns of pixels (lane = 32 row + column) and walks a segment of
//     its two rows along x; its lanes retain the multilooked series of their pixels like
//     omnibus_c2_retain_kernel.
//   * Per step of 8 planes (2 dates x 4 variables) the wave stages ITS OWN 2 + 2h rows of the 32 new columns
//     by LDS-DMA (`global_load_lds_dwordx4`: 8 rows x 128 bytes per instruction, every piece an aligned
//     line) into a wave-private slot and waits for it by count (`s_waitcnt vmcnt(n)`): no barrier.  The 2h
//     columns shared with the tile to the left are carried in LDS, as in the block form.
//   * The same lanes form the window sums (lane = plane x column quad: a patch of 4 columns x 2 rows of one
//     plane, 16-byte LDS reads, every element converted once per patch) and hand the 8 results to the
//     lanes that own the pixels through 2 KB of wave-private LDS -- LDS operations of one wave complete in
//     order, so the exchange needs no synchronisation either.
//   * Waves drift apart: while one waits for LDS or for its transfers, the other two of its SIMD add.
//
// Price: the rows above and below a wave's pair are staged by the neighbouring waves as well (2 x the
// bytes through L2 -> LDS for 3 x 3, 3 x for 5 x 5).  The waves of a block own adjacent row pairs and walk
// at the same pace, and vertically adjacent blocks are dispatched next to each other ON THE SAME XCD
// (block -> strip map below), so the second request of a line is an L2 hit: the traffic to memory stays
// that of one read.
//
// Covered: what omni_ml_plan covers, 16-byte aligned planes and rows (otherwise the block form, which has
// a per-element staging path for every tile; here only the tiles that reach over the right edge of the
// raster are staged per element).
#include "omnibus_ml_common.hpp"

namespace nd_amd {

#ifndef ND_MLW_WAVES
#define ND_MLW_WAVES 6               // waves of a block (independent of each other)
#endif
#ifndef ND_MLW_SLOTS3
#define ND_MLW_SLOTS3 2              // staging slots, 3 x 3 window
#endif
#ifndef ND_MLW_SLOTS5
#define ND_MLW_SLOTS5 2
#endif
#ifndef ND_MLW_BLOCKS3
#define ND_MLW_BLOCKS3 3             // waves per SIMD the registers are budgeted for (13.3 KB of LDS per wave at 24
                                     // dates: two blocks of six waves per CU)
#endif
#ifndef ND_MLW_BLOCKS5
#define ND_MLW_BLOCKS5 2             // (5 x 5: 23.5 KB per wave, of which 9.2 KB carried columns: one block of six)
#endif

template <int K>
struct MlwGeom {
    static constexpr int HALO = K / 2;
    static constexpr int TW = 32, TR = 2, G = 8;
    static constexpr int ROWS = TR + 2 * HALO;                    // staged rows of a plane
    static constexpr int PSZ = ROWS * TW;                         // floats per staged plane, pitch 32
    static constexpr int SLOT = G * PSZ;
    static constexpr int RES = G * TR * TW;                       // the exchange area: [row][plane][32]
    static constexpr int NCAR = G * ROWS * 2 * HALO;              // carried elements per step
    static constexpr int NSLOT = K == 3 ? ND_MLW_SLOTS3 : ND_MLW_SLOTS5;
    static constexpr int NX4 = K == 3 ? 4 : 8;                    // 16-byte transfers per step
    static constexpr int NEL = 4 * ROWS;                          // 4-byte transfers per step (edge tiles)
    static constexpr int BLOCKS = K == 3 ? ND_MLW_BLOCKS3 : ND_MLW_BLOCKS5;
    static constexpr int wave_floats(int kmax) { return NSLOT * SLOT + RES + (kmax / 2) * NCAR; }
};
constexpr int kMlwTabBytes = (33 * (int)sizeof(StreamEntry) + 15) / 16 * 16;

// ---- staging, 16-byte form ----------------------------------------------------------------------
// Image of a step in LDS: [plane ip = 2 variable + date parity][row][32 columns], 128 bytes per row.
// A variable's two planes are ROWS x 2 rows = 8 (3 x 3) or 12 (5 x 5) consecutive image rows, and one
// transfer moves 8 rows (lane -> row 8 q + (lane >> 3), bytes 16 (lane & 7) .. + 15 of it): one transfer
// per variable for 3 x 3; for 5 x 5 a whole one and a half one (lanes 0 .. 31).  The lanes' memory offsets
// relative to the variable's base are the same for all four variables (`vo`, `vo2`); the date pair and
// the tile's column go into the scalar base.  One M0 (the LDS byte address the transfers are relative to)
// per group, the rows reached through the instruction's immediate offset -- which moves the MEMORY
// address as well (tools/probe_ldsdma.hip), so the scalar bases carry minus the immediate.
// One block of assembly per group, five wait states in front (a scalar operand may have been written by
// v_readfirstlane just before; the compiler's hazard recogniser does not look into inline assembly).
__device__ __forceinline__ void mlw_dma3(const unsigned m0, const int vo, const char *b0, const char *b1,
                                         const char *b2, const char *b3)
{
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:0\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %4 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %5 offset:3072"
                 :
                 : "s"(__builtin_amdgcn_readfirstlane((int)m0)), "v"(vo), "s"(b0), "s"(b1 - 1024), "s"(b2 - 2048),
                   "s"(b3 - 3072)
                 : "memory");
}
// two variables of the 5 x 5 form: image rows 0 .. 7 and 12 .. 19 whole, then 8 .. 11 and 20 .. 23 from the
// lower half of the wave
__device__ __forceinline__ void mlw_dma5(const unsigned m0, const int vo, const int vo2, const char *b0,
                                         const char *b1)
{
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %4 offset:0\n\t"
                 "global_load_lds_dwordx4 %2, %5 offset:1536\n\t"
                 "s_mov_b32 %0, exec_hi\n\t"
                 "s_mov_b32 exec_hi, 0\n\t"
                 "global_load_lds_dwordx4 %3, %6 offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, %7 offset:2560\n\t"
                 "s_mov_b32 exec_hi, %0"
                 : "=&s"(keep)
                 : "s"(__builtin_amdgcn_readfirstlane((int)m0)), "v"(vo), "v"(vo2), "s"(b0), "s"(b1 - 1536),
                   "s"(b0 - 1024), "s"(b1 - 2560)
                 : "memory");
}

// ---- staging, per element (tiles that reach over the right edge of the raster) --------------------
// One 4-byte transfer per pair of image rows (lane -> row 2 j + (lane >> 5), column lane & 31), the border
// rule applied per lane, M0 per transfer: slow (a write of M0 waits for the transfers in front of it to
// leave the queue), and rare: one tile per strip.  Not inlined: its address arithmetic stays out of the
// walk's registers.
template <int K>
__device__ __attribute__((noinline)) void mlw_stage_edge(const float *c11, const float *c12r, const float *c12i,
                                                          const float *c22, const int64_t st, const int64_t sy,
                                                          const int ny, const int nx, const int k, const int s,
                                                          const int Xi, const int yw, const unsigned lds_slot)
{
    typedef MlwGeom<K> M;
    const int lane = (int)__lane_id();
    const int col = ml_reflect(Xi + (lane & 31), nx);
    for (int j = 0; j < M::NEL; ++j) {
        const int rr = 2 * j + (lane >> 5);
        const int ip = rr / M::ROWS, r = rr - ip * M::ROWS;
        int t = 2 * s + (ip & 1);
        t = t < k ? t : k - 1;
        const int var = ip >> 1;
        const float *base = var == 0 ? c11 : (var == 1 ? c12r : (var == 2 ? c12i : c22));
        const float *p = base + (int64_t)t * st + (int64_t)ml_reflect(yw - M::HALO + r, ny) * sy + col;
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "global_load_lds_dword %1, off"
                     :
                     : "s"(__builtin_amdgcn_readfirstlane((int)(lds_slot + 256u * (unsigned)j))), "v"(p)
                     : "memory");
    }
}

// at most n of this wave's vector memory operations outstanding (n wave-uniform; loads return in order,
// so the newest n are the ones that may still be in flight)
__device__ __forceinline__ void mlw_wait_vm(const int n)
{
    if (n >= 16)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 8)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 4)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int K, int KMAX, bool STATS, bool CHAIN>
__global__ void __launch_bounds__(64 * ND_MLW_WAVES, MlwGeom<K>::BLOCKS)
omnibus_c2_mlw_kernel(const OmniGlobalArgs<float> g, const OmniTab tab, const OmniMlArgs ml,
                      const StreamScreen<32> ss)
{
    typedef MlwGeom<K> M;
    constexpr int HALO = M::HALO, ROWS = M::ROWS, PSZ = M::PSZ, NSLOT = M::NSLOT;
    constexpr int NSTEP = KMAX / 2;                 // steps per tile (2 dates x 4 variables each)
    constexpr int PF = NSLOT - 1;                   // steps of transfers in flight
    extern __shared__ __align__(16) unsigned char nd_smem_mlw[];
    StreamEntry *tab_lds = reinterpret_cast<StreamEntry *>(nd_smem_mlw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *slots = reinterpret_cast<float *>(nd_smem_mlw + kMlwTabBytes) + wave * M::wave_floats(KMAX);
    float *res = slots + NSLOT * M::SLOT;                                 // [2 rows][8 planes][32]
    float *carry = res + M::RES;                                          // [NSTEP][8][ROWS][2h]

    const int k = g.k;
    const int nx = (int)ml.nx, ny = (int)ml.ny;
    // block -> (strip, segment): the XCD a block runs on is blockIdx.x % 8 (round-robin dispatch), and
    // consecutive blocks of one XCD take vertically adjacent strips of the same segment -- the rows two
    // strips share are then asked for twice within microseconds through the same L2
    const int xcd = blockIdx.x & 7, bj = blockIdx.x >> 3;
    const int spx = ml.spx;                         // strips per XCD (host: ceil(strips / 8))
    const int strip = xcd * spx + bj % spx, xseg = bj / spx;
    if (CHAIN && tid <= 32) tab_lds[tid] = ss.e[tid];
    if (g.write_tab && blockIdx.x == 0) {
        for (int j = tid; j <= k; j += 64 * ND_MLW_WAVES) g.tab_dev[j] = tab.e[j];
    }
    if (CHAIN) __syncthreads();                     // the only barrier: the screen's table, once
    const int yw = (strip * ND_MLW_WAVES + wave) * 2;                      // the wave's first row
    if (strip >= ml.nstrips || yw >= ny) return;

    // The wave loads the columns [Xs, Xs + 32 ntiles) and owns the OUTPUT columns [Xs - h, Xe - h): every
    // tile's outputs end h columns before its last new column; the last segment of a strip runs on to
    // the right edge.
    const bool last_seg = xseg + 1 == ml.xsegs;
    const int Xs = xseg * ml.segw;
    const int out_lo = xseg == 0 ? 0 : Xs - HALO;
    const int out_hi = last_seg ? nx : Xs + ml.segw - HALO;
    const int ntiles = last_seg ? (nx - Xs + HALO + 31) / 32 : ml.segw / 32;
    const int nstep_k = (k + 1) >> 1;               // steps that hold dates of the series
    const int total_steps = ntiles * nstep_k;
    // the first tile that is staged per element (its new columns reach over the right edge)
    const int edge_tile = last_seg ? (nx - Xs) / 32 : ntiles;
    const int edge_S = edge_tile * nstep_k;

    // ---- staging roles ----
    const unsigned lds0 = (unsigned)(uintptr_t)(ml_lds_f32 *)slots;       // LDS byte address of the slots
    const int64_t st4 = g.st * 4, sy4 = g.sy * 4;
    // lane -> image row (lane >> 3) of a variable's 2 ROWS rows (date parity 0: rows 0 .. ROWS - 1, parity 1
    // behind them): memory offset relative to the variable's base, dates 2 s / 2 s + 1 -- and, for the last
    // step of an odd series, with the second date repeating the first (`_even`)
    int vo, vo_even, vo2 = 0, vo2_even = 0;
    {
        const int rr = lane >> 3;
        const int par = rr >= ROWS ? 1 : 0, r = rr - par * ROWS;
        vo_even = ml_reflect(yw - HALO + r, ny) * (int)sy4 + (lane & 7) * 16;
        vo = vo_even + par * (int)st4;
        if (K == 5) {
            // the half transfer: image rows 8 .. 11 = parity 1, window rows 2 .. 5 (lanes 0 .. 31)
            const int r2 = 2 + (rr & 3);
            vo2_even = ml_reflect(yw - HALO + r2, ny) * (int)sy4 + (lane & 7) * 16;
            vo2 = vo2_even + (int)st4;
        }
    }
    const char *const vb[4] = {reinterpret_cast<const char *>(g.c11), reinterpret_cast<const char *>(g.c12r),
                               reinterpret_cast<const char *>(g.c12i), reinterpret_cast<const char *>(g.c22)};

    // returns the number of transfers issued
    auto stage = [&](const int s, const int Xi, const int slot, const bool edge) -> int {
        const unsigned m0b = lds0 + 4u * (unsigned)(slot * M::SLOT);
        if (edge) {
            mlw_stage_edge<K>(g.c11, g.c12r, g.c12i, g.c22, g.st, g.sy, ny, nx, k, s, Xi, yw, m0b);
            return M::NEL;
        }
        // dates 2 s, 2 s + 1 (a date beyond the series repeats the last one)
        const int64_t so = (int64_t)(2 * s) * st4 + (int64_t)Xi * 4;
        const bool rep = 2 * s + 1 >= k;
        if (K == 3) {
            mlw_dma3(m0b, rep ? vo_even : vo, vb[0] + so, vb[1] + so, vb[2] + so, vb[3] + so);
        } else {
            mlw_dma5(m0b, rep ? vo_even : vo, rep ? vo2_even : vo2, vb[0] + so, vb[1] + so);
            mlw_dma5(m0b + 3072u, rep ? vo_even : vo, rep ? vo2_even : vo2, vb[2] + so, vb[3] + so);
        }
        return M::NX4;
    };

    // ---- prologue: the carried columns of the segment's first tile, straight from memory ----
    {
        const int total = nstep_k * M::NCAR;
        for (int e = lane; e < total; e += 64) {
            const int c = e % (2 * HALO);
            const int r = (e / (2 * HALO)) % ROWS;
            const int ip = (e / (2 * HALO * ROWS)) & 7;
            const int s = e / M::NCAR;
            int t = 2 * s + (ip & 1);
            t = t < k ? t : k - 1;
            const int xm = ml_reflect(Xs - 2 * HALO + c, nx);
            const int ym = ml_reflect(yw - HALO + r, ny);
            const float *base = reinterpret_cast<const float *>(vb[ip >> 1]);
            carry[e] = base[(int64_t)t * g.st + (int64_t)ym * g.sy + xm];
        }
    }
    {
        int sp = 0, Xp = Xs;
        for (int j = 0; j < PF; ++j) {
            if (j < total_steps) stage(sp, Xp, j, j >= edge_S);
            if (++sp == nstep_k) {
                sp = 0;
                Xp += 32;
            }
        }
    }

    // compute role: lane = plane x column quad; a patch of 4 columns x 2 rows of one plane
    const int cpl = lane >> 3, cpx = lane & 7;
    const int rd_main = cpl * PSZ + 4 * cpx;                              // floats into a slot
    const int rd_c = cpl * ROWS * 2 * HALO;                               // floats into a step's carry
    const int wr_off = cpl * 32 + 4 * cpx;                                // floats into the exchange area
    // owner role: lane = 32 row + column
    const int orow = lane >> 5, ocol = lane & 31;
    const int rd_res = orow * (8 * 32) + ocol;
    const double wt = ml.wt;

    int S = 0;                       // step counter of the wave
    int slot_i = 0;                  // slot of step S
    float v[KMAX][4];

    for (int i = 0; i < ntiles; ++i) {
        const int Xi = Xs + 32 * i;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s < nstep_k) {
                float *cur = slots + slot_i * M::SLOT;
                // ---- A: the transfers of step S + PF, into the slot step S - 1 has been read out of ----
                int after = 0;                       // transfers issued behind those of step S
                if constexpr (PF >= 1) {
                    if (S + PF < total_steps) {
                        int s2 = s + PF, X2 = Xi;
                        while (s2 >= nstep_k) {
                            s2 -= nstep_k;
                            X2 += 32;
                        }
                        const int slot_p = slot_i == 0 ? NSLOT - 1 : slot_i - 1;
                        stage(s2, X2, slot_p, S + PF >= edge_S);
                    }
                    // steps S + 1 .. S + PF in flight behind step S (whole steps of the 16-byte form:
                    // counted; as soon as a per-element step is among them: everything)
                    const int left = total_steps - 1 - S;
                    after = (S + PF >= edge_S) ? 0 : (left < PF ? left : PF) * M::NX4;
                } else {
                    stage(s, Xi, slot_i, S >= edge_S);
                }
                mlw_wait_vm(after);
                // ---- B: window sums of this lane's patch ----
                {
                    double acc[2][4];
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) acc[oy][ii] = 0.0;
                    const float *P = cur + rd_main;
                    // first piece of a row: the 2h columns in front of the patch's own four
                    const float *F = cpx ? P - 2 * HALO : carry + s * M::NCAR + rd_c;
                    const int fstride = cpx ? 32 : 2 * HALO;
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) {
                        float wv[4 + 2 * HALO];
                        if (HALO == 1) {
                            const float2 f = *reinterpret_cast<const float2 *>(F + r * fstride);
                            wv[0] = f.x;
                            wv[1] = f.y;
                        } else {
                            const float4 f = *reinterpret_cast<const float4 *>(F + r * fstride);
                            wv[0] = f.x;
                            wv[1] = f.y;
                            wv[2] = f.z;
                            wv[3] = f.w;
                        }
                        const float4 q = *reinterpret_cast<const float4 *>(P + r * 32);
                        wv[2 * HALO + 0] = q.x;
                        wv[2 * HALO + 1] = q.y;
                        wv[2 * HALO + 2] = q.z;
                        wv[2 * HALO + 3] = q.w;
                        double d[4 + 2 * HALO];
#pragma unroll
                        for (int cc = 0; cc < 4 + 2 * HALO; ++cc) d[cc] = wt * (double)wv[cc];
#pragma unroll
                        for (int oy = 0; oy < 2; ++oy) {
                            const int dy = r - oy;
                            if (dy >= 0 && dy <= 2 * HALO) {
#pragma unroll
                                for (int dx = 0; dx <= 2 * HALO; ++dx)
#pragma unroll
                                    for (int ii = 0; ii < 4; ++ii) acc[oy][ii] = acc[oy][ii] + d[dx + ii];
                            }
                        }
                    }
                    // the last 2h new columns of this step's planes, for the next tile: behind the reads of
                    // the old ones above (LDS operations of a wave execute in order)
#pragma unroll
                    for (int e0 = 0; e0 < M::NCAR; e0 += 64) {
                        const int e = e0 + lane;
                        if (M::NCAR % 64 == 0 || e < M::NCAR) {
                            const int c = e % (2 * HALO), rq = e / (2 * HALO);          // rq = plane * ROWS + row
                            carry[s * M::NCAR + e] = cur[rq * 32 + 32 - 2 * HALO + c];
                        }
                    }
                    float *W = res + wr_off;
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy) {
                        const float4 o = make_float4((float)acc[oy][0], (float)acc[oy][1], (float)acc[oy][2],
                                                     (float)acc[oy][3]);
                        *reinterpret_cast<float4 *>(W + oy * (8 * 32)) = o;
                    }
                }
                // ---- C: this lane's pixel: its 8 values of the step (written by eight other lanes of the
                //      wave just above; in order, no wait).  They land in the retained registers while
                //      the next step's sums are formed ----
                {
                    const float *R = res + rd_res;
#pragma unroll
                    for (int ip = 0; ip < 8; ++ip) {
                        const int t = 2 * s + (ip & 1);
                        if (t < KMAX) v[t][ip >> 1] = R[ip * 32];
                    }
                }
                S += 1;
                slot_i = slot_i + 1 == NSLOT ? 0 : slot_i + 1;
            } else {
                // dates beyond the series: a copy of a valid date (dense_chain masks them out)
#pragma unroll
                for (int pl = 0; pl < 8; ++pl) {
                    const int t = 2 * s + (pl >> 2);
                    if (t < KMAX) v[t][pl & 3] = v[0][pl & 3];
                }
            }
        }

        // ================= the series of this tile's pixels is complete =================
        const int y = yw + orow;
        const int x = Xi - HALO + ocol;
        const bool in = (y < ny) && (x >= out_lo) && (x < out_hi);
        // valid span of a row of the tile: lanes lo .. lo + wnp - 1 of each half of the wave
        int xlo = Xi - HALO, xhi = Xi - HALO + 32;
        xlo = xlo < out_lo ? out_lo : xlo;
        xhi = xhi > out_hi ? out_hi : xhi;
        const int wnp = xhi > xlo ? xhi - xlo : 0;
        const int lo = xlo - (Xi - HALO);
        const int nrow = yw + 1 < ny ? 2 : 1;                    // rows of the pair inside the raster
        // (a number no other wave of the launch has)
        const int64_t wid = ((int64_t)(strip * ND_MLW_WAVES + wave)) * (int64_t)(ml.xsegs * (ml.segw / 32 + 2)) +
                            (int64_t)xseg * (ml.segw / 32 + 2) + i;
        const unsigned shard = (unsigned)(wid % kShards);
        const uint32_t pix = (uint32_t)((int64_t)y * nx + x);

        bool flag;
        bool dense = false;
        if (CHAIN) {
            unsigned mask;
            bool handoff, cand;
            int ks = g.k;
            asm volatile("" : "+s"(ks));
            dense_chain<float, KMAX, 32>(v, ks, in, ss, tab_lds, mask, handoff, cand);
            dense = true;
            if (handoff) mask = 0u;                              // pass B writes that pixel's changes
            if (wnp > 0) {
                if ((k & 3) == 0) {
                    // (the exchange area is the wave's own and free between two steps)
                    for (int rr = 0; rr < nrow; ++rr)
                        ml_store_change_rows(g.change + ((int64_t)(yw + rr) * nx + xlo) * (int64_t)k,
                                             reinterpret_cast<uint32_t *>(res), k, mask, lane, 32 * rr + lo, wnp);
                } else if (in) {
                    uint8_t *rr = g.change + (int64_t)pix * k;
                    for (int t = 0; t < k; ++t) rr[t] = (uint8_t)((mask >> t) & 1u);
                }
            }
            flag = handoff;
        } else {
            Accum<float> A;
            A.reset();
#pragma unroll
            for (int t = 0; t < KMAX; ++t)
                if (t < k) A.step(v[t][0], v[t][1], v[t][2], v[t][3]);
            if (STATS) {
                const float z = z_stat<float>(A, k, g.nlooks, g.e);
                double zd[1] = {(double)z}, P1[1], P2[1];
                chisq_pair<1>(zd, 4 * (k - 1), g.e.lgam, P1, P2);
                const float P = combine_P<float>(P1[0], P2[0], g.e.omega2);
                flag = in && ((double)P > g.alpha) && ml.list;
                if (in) {
                    if (g.z_out) g.z_out[pix] = z;
                    if (g.p_out) g.p_out[pix] = P;
                }
            } else {
                flag = in && (z_approx<float>(A, k, g.nlooks, g.e) >= g.e.zlo_a);
            }
        }

        // ---- list + dump (the multilooked series exists nowhere else: the dump holds every listed pixel) ----
        const unsigned long long m = __ballot(flag);
        if (m != 0ull) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(g.flag_count + shard * kCounterStride, (unsigned)__popcll(m));
            base = __shfl(base, 0);
            if (flag) {
                const unsigned slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                g.flag_idx[(size_t)shard * g.seg + slot] = pix;
                float *dd = g.dump + ((int64_t)shard * g.dump_cap + slot) * (int64_t)(4 * k);
#pragma unroll
                for (int t = 0; t < KMAX; ++t) {
                    if (t < k && slot < g.dump_cap) {        // (capacity = list length: omni_ml_plan)
                        Pack<float, 4> q;
                        q.v[0] = v[t][0];
                        q.v[1] = v[t][1];
                        q.v[2] = v[t][2];
                        q.v[3] = v[t][3];
                        *reinterpret_cast<Pack<float, 4> *>(dd + 4 * t) = q;
                    }
                }
            }
        }
        // ---- a sparse wave zero-fills its own slices of the change map (np.zeros, nd/_change.pyx:275) ----
        if (!dense && wnp > 0 && ml.list) {
            for (int rr = 0; rr < nrow; ++rr)
                zero_fill_span(g.change + ((int64_t)(yw + rr) * nx + xlo) * (int64_t)k, wnp * k, lane);
        }
    }
}

// -----------------------------------------------------------------------------------------
// host side
// -----------------------------------------------------------------------------------------
template <int K, int KMAX>
static int launch_mlw_k(const OmniGlobalArgs<float> &g, const OmniTab &tab, const OmniMlArgs &a,
                        const StreamScreen<32> *ss, bool stats, int64_t nblocks, hipStream_t stream)
{
    typedef MlwGeom<K> M;
    const size_t lds = (size_t)kMlwTabBytes + (size_t)ND_MLW_WAVES * M::wave_floats(KMAX) * sizeof(float);
    const dim3 grid((unsigned)nblocks), block(64 * ND_MLW_WAVES);
    StreamScreen<32> none;
    if (!ss) memset(&none, 0, sizeof(none));
#define ND_MLW_LAUNCH(STATS_, CHAIN_)                                                                          \
    do {                                                                                                       \
        if (lds > 64 * 1024)                                                                                   \
            ND_HIP_CHECK(hipFuncSetAttribute(                                                                  \
                reinterpret_cast<const void *>(&omnibus_c2_mlw_kernel<K, KMAX, STATS_, CHAIN_>),               \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                        \
        hipLaunchKernelGGL((omnibus_c2_mlw_kernel<K, KMAX, STATS_, CHAIN_>), grid, block, lds, stream, g, tab, a, \
                           ss ? *ss : none);                                                                   \
    } while (0)
    if (ss)
        ND_MLW_LAUNCH(false, true);
    else if (stats)
        ND_MLW_LAUNCH(true, false);
    else
        ND_MLW_LAUNCH(false, false);
#undef ND_MLW_LAUNCH
    return ND_AMD_OK;
}

int launch_mlw_pass_a(const OmniGlobalArgs<float> &g, const OmniTab &tab, const OmniMlPlan &p, const OmniMlArgs &a0,
                      const StreamScreen<32> *ss, bool stats, hipStream_t stream)
{
    OmniMlArgs a = a0;
    a.segw = p.w_segw;
    a.xsegs = p.w_xsegs;
    a.nstrips = p.w_strips;
    a.spx = (p.w_strips + 7) / 8;         // strips per XCD (the kernel's block -> strip map)
    a.trace = nullptr;
    const int64_t nblocks = (int64_t)8 * a.spx * p.w_xsegs;
    const int k = g.k;
#define ND_MLW_K(KK)                                                                      \
    do {                                                                                  \
        if (k <= 8)                                                                       \
            return launch_mlw_k<KK, 8>(g, tab, a, ss, stats, nblocks, stream);            \
        else if (k <= 16)                                                                 \
            return launch_mlw_k<KK, 16>(g, tab, a, ss, stats, nblocks, stream);           \
        else                                                                              \
            return launch_mlw_k<KK, 24>(g, tab, a, ss, stats, nblocks, stream);           \
    } while (0)
    if (p.ml == 3)
        ND_MLW_K(3);
    else
        ND_MLW_K(5);
#undef ND_MLW_K
}

int mlw_waves_per_block() { return ND_MLW_WAVES; }

}  // namespace nd_amd
