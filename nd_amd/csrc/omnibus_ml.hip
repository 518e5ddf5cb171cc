// nd_amd/csrc/omnibus_ml.hip -- OmnibusTest(ml=w): spatial multilooking fused into pass A.
//
// Reference: nd/change.py:61-64 -- `ds_m = BoxcarFilter(w=ml).apply(ds_m); n = ml ** 2` in front of
// nd._change.change_detection; BoxcarFilter is scipy.ndimage.convolve with the kernel
// ones((w, w)) / w**2 and mode 'reflect' (nd/filters.py:256-267, 294-298), i.e. per output
//     double tmp = 0; for (dy, dx) in row-major window order: tmp += (1 / w^2) * (double)x[y+dy][x+dx]
//     out = (float)tmp
// product and sum rounded separately.  The separate path runs that filter over the 4 k planes
// (read + write of the stack) and then the test (another read): three trips over the stack.  Here
// the planes are read ONCE: the multilooked value of a (date, variable) exists only in registers.
//
// Shape of the kernel.  The change-point search needs the multilooked SERIES of a pixel (4 k values)
// -- in pass B for the listed pixels (dump), or at once for the fused search (dense_chain) -- so a
// thread owns one pixel and retains its series in registers, exactly like omnibus_c2_retain_kernel /
// omnibus_c2_chain_kernel; the window sums need the neighbours, which live in other lanes and other
// waves, so the planes pass through LDS:
//
//   * A block of 768 threads owns a strip of 12 rows and WALKS a segment of it along x in tiles of
//     64 columns.  Wave w owns tile row w, lane j the pixel in column j of the tile.
//   * A tile is consumed in steps of 8 planes (2 dates x 4 variables).  The planes of a step are
//     staged by LDS-DMA (memory -> LDS, no registers): 12 + 2h rows of the 64 NEW columns of the tile,
//     as `buffer_load_dwordx4 ... lds` -- one instruction moves 4 rows x 256 bytes, each piece
//     aligned and whole (32 instructions per step for the whole block; the first form moved one row
//     per `buffer_load_dword ... lds`: 112 instructions per step at ~29 cycles each through the
//     texture addresser -- half of the step's time, measured with s_memtime stamps).  No column is
//     fetched twice: the 2h columns a tile shares with its left neighbour are carried over in LDS
//     (a tile's outputs are the 64 columns that end h columns before its last new column).  Only the
//     h rows above and below a strip are fetched by two blocks.  Tiles that reach over the right
//     edge of the raster take a per-element form with the border rule applied per lane.
//   * Three staging slots: while step S is computed the transfers of S + 1 and S + 2 are in flight.
//     The transfers are issued from inline assembly and waited for by count: the compiler's
//     wait-count pass would otherwise put `s_waitcnt vmcnt(0)` in front of every LDS read that
//     follows a transfer (it cannot prove dynamic LDS disjoint), i.e. wait for the planes of two
//     steps ahead before the current step's first read.
//   * Compute role: a thread forms the window sums of a patch of 4 columns x 2 rows of ONE plane,
//     walking the patch's 2 + 2h input rows once (16-byte LDS reads, bank-conflict free by
//     construction of the lane -> (plane, patch) map); every element is converted to double and
//     multiplied by 1 / w^2 once per patch, every output receives its w^2 terms in scipy's order.
//     The 8 results (float) go to a result area in LDS; after a barrier every thread picks up the 8
//     values of ITS pixel: v[2 s + ...][0..3].
//   * After the last step of a tile the series is complete and the kernel continues like the plain
//     forms: fold + screen + list + dump + zero-fill (sparse regime) or dense_chain (fused search).
//
// Traffic: 4 k planes x (12 + 2h) / 12 rows, nothing else.  Work: w^2 dependent double additions per
// value (scipy's order leaves no sharing between neighbouring windows) -- vector issue, not memory,
// is what bounds the kernel.
#include "omnibus_ml_common.hpp"

namespace nd_amd {

// Steps the L2 prefetch of the idle waves (8 .. 11) runs ahead of the transfers.  0 = none, the default:
// measured in round 5 (24 x 4096^2, 3 x 3 / 5 x 5 window, one box): none 2.03 / 3.13 ms, one step ahead
// 2.14 / 3.27, two 2.15 / 3.22, three 2.14 / 3.34, five 2.23 / 3.34, eight 2.31 / 3.36 -- the more requests
// in flight the slower: the walk is not waiting for memory (a staging-only probe of the same walk,
// tools/probe_tiles.hip, moves the 3 x 3 kernel's bytes in 1.24 ms).
#ifndef ND_ML_PREFETCH
#define ND_ML_PREFETCH 0
#endif
#ifdef ND_ML_TRACE
#define ND_ML_TRC_WORDS (12 * 16 * 5)
#else
#define ND_ML_TRC_WORDS 0
#endif

template <int K>
struct MlGeom {
    static constexpr int HALO = K / 2;
    static constexpr int W = 64, HT = kMlTileRows, NT = 64 * HT, NWAVE = HT, G = 8;
    static constexpr int ROWS = HT + 2 * HALO;                    // staged rows of a plane
    static constexpr int PROWS = 16;                              // rows of a plane's LDS image (4 per transfer)
    static constexpr int PSZ = PROWS * 64;                        // floats per staged plane, pitch 64
    static constexpr int SLOT = G * PSZ;
    static constexpr int RES = G * NT;
    static constexpr int NCAR = G * ROWS * 2 * HALO;              // carried elements per step
    static constexpr int NSLOT = K == 3 ? 3 : 2;                  // NSLOT - 1 steps of transfers in flight (LDS budget)
    static_assert(ROWS <= PROWS, "window too tall for the plane image");
};

// all LDS operations of this wave done, then the workgroup barrier (no wait for the LDS-DMA transfers
// in flight: those are waited for explicitly, by count)
__device__ __forceinline__ void ml_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// wait until at most n of this wave's transfers are outstanding (n wave-uniform: what one step of
// four 16-byte or sixteen 4-byte transfers can leave)
__device__ __forceinline__ void ml_wait_vm(const int n)
{
#ifdef ND_ML_WAIT0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
#endif
    if (n >= 16)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 4)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// LDS-DMA from inline assembly (see the header): memory -> LDS at `lds_addr` + (4 | 16) * lane.
// (Wait counts the compiler emits for its own loads stay safe: transfers it does not know about only
// make `vmcnt(n)` wait for more than it had to.  One wait state between a write of M0 and the
// transfer that reads it: the s_nop.  M0 is reserved: the compiler keeps nothing in it on gfx9.)
typedef int ml_v4i __attribute__((ext_vector_type(4)));
// A wave's transfers of one step, issued as ONE block of assembly: M0 (the LDS byte address the
// transfers are relative to), then four transfers that reach the rows of the wave's plane image
// through the instruction's 12-bit offset.
//  * M0 once per step: a write of M0 in front of EVERY transfer made the transfers of a wave run one
//    at a time (~200 - 500 cycles each, s_memtime stamps; 3 300 of a step's 6 600 cycles).
//  * The immediate offset moves the memory address as well as the LDS address
//    (tools/probe_ldsdma.hip): the descriptor starts 4096 bytes early and every scalar offset carries
//    4096 minus the immediate.
//  * One block, operands in registers of their own, five wait states in front: what the compiler's
//    hazard recogniser guarantees for its own VMEM instructions -- no VALU write (v_readlane restoring
//    a spilled SGPR, v_readfirstlane) of a scalar operand within five wait states, no reuse of an
//    address register between two transfers -- it cannot guarantee for inline assembly, which it does
//    not look into.  (Found the hard way: with one statement per transfer, results changed with
//    register allocation.)
#ifdef ND_ML_NT
#define ND_ML_POL " nt"
#else
#define ND_ML_POL ""
#endif
__device__ __forceinline__ void ml_dma16x4(const ml_v4i rsrc, const unsigned m0, const int a0, const int a1,
                                           const int a2, const int a3, const int so)
{
    const int o0 = __builtin_amdgcn_readfirstlane(so + 4096), o1 = __builtin_amdgcn_readfirstlane(so + 3072),
              o2 = __builtin_amdgcn_readfirstlane(so + 2048), o3 = __builtin_amdgcn_readfirstlane(so + 1024);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %5, %6 offen offset:0" ND_ML_POL " lds\n\t"
                 "buffer_load_dwordx4 %2, %5, %7 offen offset:1024" ND_ML_POL " lds\n\t"
                 "buffer_load_dwordx4 %3, %5, %8 offen offset:2048" ND_ML_POL " lds\n\t"
                 "buffer_load_dwordx4 %4, %5, %9 offen offset:3072" ND_ML_POL " lds"
                 :
                 : "s"(__builtin_amdgcn_readfirstlane((int)m0)), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(rsrc),
                   "s"(o0), "s"(o1), "s"(o2), "s"(o3)
                 : "memory");
}
// four rows of the per-element form: rows R .. R + 3 of the plane image, one address register
template <int R>
__device__ __forceinline__ void ml_dma4x4(const ml_v4i rsrc, const unsigned m0, const int voff, const int s0,
                                          const int s1, const int s2, const int s3)
{
    const int o0 = __builtin_amdgcn_readfirstlane(s0 + 4096 - 256 * R),
              o1 = __builtin_amdgcn_readfirstlane(s1 + 4096 - 256 * (R + 1)),
              o2 = __builtin_amdgcn_readfirstlane(s2 + 4096 - 256 * (R + 2)),
              o3 = __builtin_amdgcn_readfirstlane(s3 + 4096 - 256 * (R + 3));
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "buffer_load_dword %1, %2, %3 offen offset:%7" ND_ML_POL " lds\n\t"
                 "buffer_load_dword %1, %2, %4 offen offset:%8" ND_ML_POL " lds\n\t"
                 "buffer_load_dword %1, %2, %5 offen offset:%9" ND_ML_POL " lds\n\t"
                 "buffer_load_dword %1, %2, %6 offen offset:%10" ND_ML_POL " lds"
                 :
                 : "s"(__builtin_amdgcn_readfirstlane((int)m0)), "v"(voff), "s"(rsrc), "s"(o0), "s"(o1), "s"(o2),
                   "s"(o3), "n"(256 * R), "n"(256 * (R + 1)), "n"(256 * (R + 2)), "n"(256 * (R + 3))
                 : "memory");
}
__device__ __forceinline__ ml_v4i ml_make_rsrc(const float *p)
{
    const uint64_t a = (uint64_t)(uintptr_t)p;
    ml_v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));      // stride 0
    r.z = 0x7fffffff;                                                                   // bytes addressable
    r.w = 0x00020000;
    return r;
}

template <int K, int KMAX, bool STATS, bool CHAIN>
__global__ void __launch_bounds__(64 * kMlTileRows)
omnibus_c2_ml_kernel(const OmniGlobalArgs<float> g, const OmniTab tab, const OmniMlArgs ml,
                     const StreamScreen<32> ss)
{
    typedef MlGeom<K> M;
    constexpr int HALO = M::HALO, ROWS = M::ROWS, PSZ = M::PSZ, NT = M::NT;
    constexpr int NSTEP = KMAX / 2;                 // steps per tile (2 dates x 4 variables each)
    extern __shared__ __align__(16) unsigned char nd_smem_ml[];
    float *res = reinterpret_cast<float *>(nd_smem_ml);                   // [2][wave][8][64]
    float *slots = res + 2 * M::RES;                                          // [3][8][16][64]
    float *carry = slots + M::NSLOT * M::SLOT;                            // [NSTEP][8][ROWS][2h]
    StreamEntry *tab_lds = reinterpret_cast<StreamEntry *>(carry + NSTEP * M::NCAR);
    int *rowtab = reinterpret_cast<int *>(tab_lds + 33);                  // byte offsets of the 16 image rows


    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = g.k;
    const int nx = (int)ml.nx, ny = (int)ml.ny;
    const int64_t b = blockIdx.x;
    const int strip = (int)(b / ml.xsegs), xseg = (int)(b - (int64_t)strip * ml.xsegs);
    const int y0 = strip * M::HT;
    // The block loads the columns [Xs, Xs + 64 ntiles) and owns the OUTPUT columns [Xs - h, Xe - h):
    // every tile's outputs end h columns before its last new column, so a segment needs no tile
    // beyond its own columns -- only the last segment of a strip, which owns the columns up to the
    // right edge, runs one tile further.
    const bool last_seg = xseg + 1 == ml.xsegs;
    const int Xs = xseg * ml.segw;
    const int out_lo = xseg == 0 ? 0 : Xs - HALO;
    const int out_hi = last_seg ? nx : Xs + ml.segw - HALO;
    const int ntiles = last_seg ? (nx - Xs + HALO + 63) / 64 : ml.segw / 64;
    const int nstep_k = (k + 1) >> 1;               // steps that hold dates of the series

    if (CHAIN && tid <= 32) tab_lds[tid] = ss.e[tid];
    if (tid < M::PROWS) rowtab[tid] = ml_reflect(y0 - HALO + (tid < ROWS ? tid : ROWS - 1), ny) * (int)g.sy * 4;
    if (g.write_tab && b == 0) {
        for (int j = tid; j <= k; j += NT) g.tab_dev[j] = tab.e[j];
    }

    // ---- staging ----
    const float *vp[4] = {g.c11, g.c12r, g.c12i, g.c22};
    const unsigned lds0 = (unsigned)(uintptr_t)(ml_lds_f32 *)slots;       // LDS byte address of the slots
#if ND_ML_PREFETCH > 0
    // 256 bytes behind the table (and the trace area of diagnostic builds): where the prefetch reads land
    const unsigned pf_dump = lds0 + (unsigned)(reinterpret_cast<unsigned char *>(rowtab + 16 + ND_ML_TRC_WORDS) -
                                               reinterpret_cast<unsigned char *>(slots));
#endif
    const int sstep = (int)g.st * 4;                             // bytes between dates (host: k * st * 4 < 2^31)
    // Waves 0 .. 7 stage: wave w the plane w of the step (variable w & 3, date w >> 2), whose LDS image
    // (16 rows x 256 bytes) is within reach of the transfers' 12-bit offset from one value of M0.
    // (a) 16-byte form: four transfers of 4 rows x 64 columns; lane -> row 4 q + (lane >> 4), columns
    //     4 (lane & 15) .. + 3 of the tile.
    // (b) 4-byte form (tiles over the right edge, unaligned planes): one transfer per row, the border
    //     rule applied per lane.
    const bool stager = wave < 8;
    const int myvar = wave & 3, mydate = (wave >> 2) & 1;
    // (the transfers' immediate offset moves the memory address as well as the LDS address: the
    //  descriptor starts 4096 bytes early and every scalar offset carries 4096 minus the immediate)
    const float *myp = myvar == 0 ? g.c11 : (myvar == 1 ? g.c12r : (myvar == 2 ? g.c12i : g.c22));
    const ml_v4i myrs = ml_make_rsrc(myp - 1024);
    int rowoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int r = 4 * q + (lane >> 4);
        r = r < ROWS ? r : ROWS - 1;                             // (rows of the image beyond the staged ones)
        rowoff[q] = ml_reflect(y0 - HALO + r, ny) * (int)g.sy * 4 + (lane & 15) * 16;
    }

    // returns the number of transfers this wave issued
    auto stage = [&](const int s, const int Xi, const int slot) -> int {
        if (!stager) return 0;
        // planes 8 s .. 8 s + 7 = dates 2 s, 2 s + 1 (a date beyond the series repeats the last one)
        int t = 2 * s + mydate;
        t = t < k ? t : k - 1;
        const int so = t * sstep;
        const unsigned m0b = lds0 + 4u * (unsigned)(slot * M::SLOT + wave * PSZ);
        if (ml.x4 && Xi + 64 <= nx) {
            const int xb = Xi * 4;
            ml_dma16x4(myrs, m0b, rowoff[0] + xb, rowoff[1] + xb, rowoff[2] + xb, rowoff[3] + xb, so);
            return 4;
        }
        // (rare: the row offsets come from a table in LDS -- worked out per step they were hoisted out of
        //  the walk by the dozen and spilled: ~100 v_readlane per step)
        const int voff = ml_reflect(Xi + lane, nx) * 4;
        auto ro = [&](const int r) { return so + rowtab[r]; };
        ml_dma4x4<0>(myrs, m0b, voff, ro(0), ro(1), ro(2), ro(3));
        ml_dma4x4<4>(myrs, m0b, voff, ro(4), ro(5), ro(6), ro(7));
        ml_dma4x4<8>(myrs, m0b, voff, ro(8), ro(9), ro(10), ro(11));
        ml_dma4x4<12>(myrs, m0b, voff, ro(12), ro(13), ro(14), ro(15));
        return 16;
    };

    // ---- prologue: the carried columns of the segment's first tile, straight from memory ----
    {
        const int total = 8 * nstep_k * ROWS * 2 * HALO;
        for (int e = tid; e < total; e += NT) {
            const int c = e % (2 * HALO);
            const int r = (e / (2 * HALO)) % ROWS;
            const int q = e / (2 * HALO * ROWS);                 // plane = 4 * date + variable
            int t = q >> 2;
            t = t < k ? t : k - 1;
            const int xm = ml_reflect(Xs - 2 * HALO + c, nx);
            const int ym = ml_reflect(y0 - HALO + r, ny);
            carry[e] = vp[q & 3][(int64_t)t * g.st + (int64_t)ym * g.sy + xm];
        }
    }
    const int total_steps = ntiles * nstep_k;
    constexpr int PF = M::NSLOT - 1;                             // steps of transfers in flight
    // rowtab / tab_lds (written by wave 0 above) are read by the per-element form of stage() in the
    // other staging waves: one barrier in front of the first transfers
    ml_barrier();
    {
        int sp = 0, Xp = Xs, c1 = 0;
        for (int j = 0; j < PF; ++j) {
            if (j < total_steps) {
                const int c = stage(sp, Xp, j);
                if (j == 1) c1 = c;
            }
            if (++sp == nstep_k) {
                sp = 0;
                Xp += 64;
            }
        }
        ml_wait_vm(c1);                                          // step 0 has landed (this wave's part)
    }
    __syncthreads();

    // compute role: waves 0 .. 5 the patch rows of planes 0 .. 3, waves 6 .. 11 of planes 4 .. 7
    const int py = wave % (M::HT / 2), quad = wave / (M::HT / 2);
    const int cpl = quad * 4 + (lane >> 4), cpx = lane & 15;
    // the patch's window columns are the tile's new columns 4 px - 2h .. 4 px + 3: for px = 0 the
    // first 2h of them are the carried ones (read from the carry buffer, rows 2h floats apart)
    const int rd_main = cpl * PSZ + (2 * py) * 64 + 4 * cpx;                         // floats into a slot
    const int rd_c = (cpl * ROWS + 2 * py) * 2 * HALO;                               // floats into a step's carry
    const int wr_off = (2 * py) * (8 * 64) + cpl * 64 + 4 * cpx;                     // floats into a result buffer
    // carried columns: the wave that stages plane w also saves its last 2h new columns (lane e < ROWS 2h)
    const int co_rd = wave * PSZ + (lane / (2 * HALO)) * 64 + 64 - 2 * HALO + lane % (2 * HALO);
    const int co_wr = wave * ROWS * 2 * HALO + lane;
    const bool co_lane = stager && lane < ROWS * 2 * HALO;
    const double wt = ml.wt;

    int S = 0;                       // step counter of the block
    int slot_i = 0;                  // slot of step S
    int s_prev = 0;                  // plane group of step S - 1
    float cv = 0.f;                  // carried column element of step S - 1 (see the end of a step)
    float v[KMAX][4];

#ifdef ND_ML_TRACE
    // time stamps of steps 32 .. 47 of one block, kept in LDS (no memory traffic inside the loop) and
    // written out when the block ends
    unsigned *trc = reinterpret_cast<unsigned *>(rowtab + 16);
#define ML_STAMP(j)                                                                                  \
    do {                                                                                             \
        if (ml.trace && b == ml.trace_block && lane == 0 && S >= 32 && S < 48)                        \
            trc[(wave * 16 + (S - 32)) * 5 + (j)] = (unsigned)__builtin_amdgcn_s_memtime();          \
    } while (0)
#else
#define ML_STAMP(j)
#endif
    // One barrier per step.  Behind the barrier that opens step S everything of step S - 1 is visible:
    // its window sums (result buffer (S - 1) & 1), and every patch has read its planes.  Step S then
    //   A. saves the carried columns of step S - 1's planes and sends the transfers of step S + PF into
    //      that slot (the wave that stages a plane is the one that saves its columns: no other order
    //      is needed);
    //   B. forms the window sums of step S into result buffer S & 1;
    //   C. waits for its transfers of step S + 1, meets the others at the barrier and asks for its pixels'
    //      8 values of step S (the loads land in the retained registers while step S + 1 runs; behind
    //      the last step of a tile the tail consumes them at once).
    for (int i = 0; i < ntiles; ++i) {
        const int Xi = Xs + 64 * i;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s < nstep_k) {
                float *cur = slots + slot_i * M::SLOT;
                const int slot_p = slot_i == 0 ? M::NSLOT - 1 : slot_i - 1;      // slot of step S - 1 = of S + PF
                ML_STAMP(0);
                // ---- A ----
                if (S > 0 && co_lane) carry[s_prev * M::NCAR + co_wr] = cv;
                int cnt_new = 0;
#ifndef ND_ML_NO_DMA
                if (S + PF < total_steps) {
                    int s2 = s + PF, X2 = Xi;
                    while (s2 >= nstep_k) {
                        s2 -= nstep_k;
                        X2 += 64;
                    }
                    asm volatile("" : : "v"(cv) : "memory");                    // the columns above are in registers
                    cnt_new = stage(s2, X2, slot_p);
                }
#endif
#if ND_ML_PREFETCH > 0
                // ---- the waves that stage nothing (8 .. 11) ask for the planes of step S + PF + D: one
                //      4-byte LDS-DMA read per 128-byte line into a dump area nobody reads (no register
                //      to keep, nothing to wait for).  The lines then sit in the L2 (or the Infinity Cache)
                //      when the transfers of that step are issued D steps later ----
                if (!stager && S + PF + ND_ML_PREFETCH < total_steps) {
                    int s3 = s + PF + ND_ML_PREFETCH, X3 = Xi;
                    while (s3 >= nstep_k) {
                        s3 -= nstep_k;
                        X3 += 64;
                    }
                    constexpr int PFL = 2 * ROWS;                // lines of a plane's 64 new columns
                    const int hi = lane >= PFL ? 1 : 0;
                    const int pl = 2 * (wave - 8) + hi, lr = lane - hi * PFL;
                    if (lane < 2 * PFL && X3 < nx) {
                        int t = 2 * s3 + (pl >> 2);
                        t = t < k ? t : k - 1;
                        int xc = X3 + 32 * (lr & 1);
                        xc = xc < nx ? xc : nx - 1;
                        const int var = pl & 3;
                        const float *pb = var == 0 ? g.c11 : (var == 1 ? g.c12r : (var == 2 ? g.c12i : g.c22));
                        const float *pp = pb + (int64_t)t * g.st + (rowtab[lr >> 1] >> 2) + xc;
                        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\t"
                                     "global_load_lds_dword %1, off"
                                     :
                                     : "s"(__builtin_amdgcn_readfirstlane((int)pf_dump)), "v"(pp)
                                     : "memory");
                    }
                }
#endif
                // (a series of one step per tile: the columns saved above are the ones this very step
                //  reads -- the only case in which A and B of one step touch the same carry entries)
                if (nstep_k == 1) ml_barrier();
                ML_STAMP(1);
                // ---- B: window sums of this thread's patch ----
#ifdef ND_ML_NO_COMPUTE
                if (g.k < 0)
#endif
                {
                    double acc[2][4];
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) acc[oy][ii] = 0.0;
                    const float *P = cur + rd_main;
                    // first piece of a row: the 2h columns in front of the patch's own four
                    const float *F = cpx ? P - 2 * HALO : carry + s * M::NCAR + rd_c;
                    const int fstride = cpx ? 64 : 2 * HALO;
#pragma unroll
                    for (int r = 0; r < 2 + 2 * HALO; ++r) {
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
                        const float4 q = *reinterpret_cast<const float4 *>(P + r * 64);
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
                    float *W = res + (S & 1) * M::RES + wr_off;
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy) {
                        const float4 o = make_float4((float)acc[oy][0], (float)acc[oy][1], (float)acc[oy][2],
                                                     (float)acc[oy][3]);
                        *reinterpret_cast<float4 *>(W + oy * (8 * 64)) = o;
                    }
                }
                ML_STAMP(2);
                // ---- C: the transfers of the next step have landed (this wave's), then everybody's ----
                // (the waves that stage nothing have nothing to wait for: their prefetch reads go nowhere)
                if (stager || ND_ML_PREFETCH == 0) ml_wait_vm(PF >= 2 ? cnt_new : 0);
                ML_STAMP(3);
                ml_barrier();
                ML_STAMP(4);
                // the last 2h new columns of this step's planes, for the next tile (saved by the wave that
                // staged the plane, in front of its next transfers into this slot: part A of the next step)
                if (co_lane) cv = cur[co_rd];
                // this thread's pixel: its 8 values of the step (the loads land in the retained registers
                // while the next step's sums are formed; the tile's last step is followed by the tail)
                {
                    const float *R = res + (S & 1) * M::RES + wave * (8 * 64) + lane;
#pragma unroll
                    for (int pl = 0; pl < 8; ++pl) {
                        const int t = 2 * s + (pl >> 2);
                        if (t < KMAX) v[t][pl & 3] = R[pl * 64];
                    }
                }
                s_prev = s;
                S += 1;
                slot_i = slot_i + 1 == M::NSLOT ? 0 : slot_i + 1;
            } else {
                // dates beyond the series: a copy of a valid date (dense_chain masks them out)
#pragma unroll
                for (int pl = 0; pl < 8; ++pl) {
                    const int t = 2 * s + (pl >> 2);
                    if (t < KMAX) v[t][pl & 3] = v[0][pl & 3];
                }
            }
        }
        const int Sres = (S - 1) & 1;            // result buffer of the tile's last step: this wave's rows are free

        // ================= the series of this tile's pixels is complete =================
        const int y = y0 + wave;
        const int x = Xi - HALO + lane;
        const bool in = (y < ny) && (x >= out_lo) && (x < out_hi);
        // valid span of the wave: lanes lo .. lo + wnp - 1
        int xlo = Xi - HALO, xhi = Xi - HALO + 64;
        xlo = xlo < out_lo ? out_lo : xlo;
        xhi = xhi > out_hi ? out_hi : xhi;
        const int wnp = (y < ny && xhi > xlo) ? xhi - xlo : 0;
        const int lo = xlo - (Xi - HALO);
        uint8_t *wob = g.change + ((int64_t)y * nx + xlo) * (int64_t)k;
        // (a number no other wave of the launch has)
        const int64_t wid = ((int64_t)strip * ml.tmax + (Xi >> 6)) * (int64_t)M::NWAVE + wave;
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
            if (STATS) {
                // (round 6) the z / P rasters from the series the search just walked: the reference's forward fold
                // (nd/_change.pyx:53-77) and the chi-square pair (the same evaluation as every other form: the rasters are bit-identical whichever kernel writes them)
                Accum<float> A;
                A.reset();
#pragma unroll
                for (int t = 0; t < KMAX; ++t)
                    if (t < k) A.step(v[t][0], v[t][1], v[t][2], v[t][3]);
                const float z = z_stat<float>(A, k, g.nlooks, g.e);
                double zd[1] = {(double)z}, P1[1], P2[1];
                chisq_pair<1>(zd, 4 * (k - 1), g.e.lgam, P1, P2);
                const float P = combine_P<float>(P1[0], P2[0], g.e.omega2);
                if (in) {
                    if (g.z_out) g.z_out[pix] = z;
                    if (g.p_out) g.p_out[pix] = P;
                }
            }
            if (wnp > 0) {
                if ((k & 3) == 0) {
                    // (the wave's rows of the result area are its own until the next step's barrier)
                    ml_store_change_rows(wob, reinterpret_cast<uint32_t *>(res + Sres * M::RES + wave * (8 * 64)), k, mask,
                                         lane, lo, wnp);
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
        // ---- a sparse wave zero-fills its own slice of the change map (np.zeros, nd/_change.pyx:275) ----
        if (!dense && wnp > 0 && ml.list) zero_fill_span(wob, wnp * k, lane);
    }
#ifdef ND_ML_TRACE
    __syncthreads();
    if (ml.trace && b == ml.trace_block)
        for (int e = tid; e < 12 * 16 * 5; e += NT) ml.trace[e] = trc[e];
#endif
}

// -----------------------------------------------------------------------------------------
// host side
// -----------------------------------------------------------------------------------------
bool omni_ml_plan(int64_t ny, int64_t nx, int64_t k, int64_t sy, int64_t sx, int64_t st, int ml, int dtype,
                  OmniMlPlan *p)
{
    memset(p, 0, sizeof(*p));
    if (dtype != ND_AMD_F32 || (ml != 3 && ml != 5)) return false;
    if (k < 2 || k > 24) return false;
    const int halo = ml / 2;
    if (nx <= 2 * halo || ny <= 2 * halo || nx > 0x3fffffff || ny > 0x3fffffff) return false;
    if (sx != 1 || sy < nx || st < 0) return false;
    // buffer offsets are 32-bit: date + row + column, in bytes
    if (((int64_t)k * st + ny * sy + nx) * 4 + 8192 >= 0x7fffffffLL) return false;
    p->ml = ml;
    p->ny = ny;
    p->nx = nx;
    p->strips = (int)ceil_div(ny, (int64_t)kMlTileRows);
    // segments per strip: enough blocks to balance 256 CUs (one block per CU), at least 4 tiles each
    const int64_t tiles_x = ceil_div(nx, 64);
    int xsegs = 1;
    while ((int64_t)p->strips * xsegs < 8 * 256 && tiles_x / (xsegs * 2) >= 4) xsegs *= 2;
    p->segw = (int)(ceil_div(tiles_x, xsegs) * 64);
    p->xsegs = (int)ceil_div(nx, (int64_t)p->segw);
    p->nblocks = (int64_t)p->strips * p->xsegs;
    // waves that list pixels: tiles per segment <= segw / 64 + 1
    const int64_t waves = (int64_t)p->strips * (ceil_div(nx, 64) + 2) * kMlTileRows;
    p->seg = (uint32_t)(ceil_div(waves, (int64_t)kShards) * 64 + 64);
    return true;
}

template <int K, int KMAX>
static int launch_ml_k(const OmniGlobalArgs<float> &g, const OmniTab &tab, const OmniMlArgs &a,
                        const StreamScreen<32> *ss, bool stats, int64_t nblocks, hipStream_t stream)
{
    typedef MlGeom<K> M;
    const size_t lds = ((size_t)M::NSLOT * M::SLOT + 2 * M::RES + (size_t)(KMAX / 2) * M::NCAR) * sizeof(float) +
                       33 * sizeof(StreamEntry) + 16 * sizeof(int) + ND_ML_TRC_WORDS * 4 + (ND_ML_PREFETCH > 0 ? 256 : 0);
    const dim3 grid((unsigned)nblocks), block(M::NT);
    StreamScreen<32> none;
    if (!ss) memset(&none, 0, sizeof(none));
#define ND_ML_LAUNCH(STATS_, CHAIN_)                                                                          \
    do {                                                                                                      \
        /* on every launch: the attribute belongs to the CURRENT device's function object (one thread */     \
        /* per device in algorithm.parallel(devices=...)), and a failure must not go unnoticed */            \
        ND_HIP_CHECK(hipFuncSetAttribute(                                                                     \
            reinterpret_cast<const void *>(&omnibus_c2_ml_kernel<K, KMAX, STATS_, CHAIN_>),                   \
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                           \
        hipLaunchKernelGGL((omnibus_c2_ml_kernel<K, KMAX, STATS_, CHAIN_>), grid, block, lds, stream, g, tab, a, \
                           ss ? *ss : none);                                                                  \
    } while (0)
    if (ss && stats)
        ND_ML_LAUNCH(true, true);
    else if (ss)
        ND_ML_LAUNCH(false, true);
    else if (stats)
        ND_ML_LAUNCH(true, false);
    else
        ND_ML_LAUNCH(false, false);
#undef ND_ML_LAUNCH
    return ND_AMD_OK;
}

int launch_ml_pass_a(const OmniGlobalArgs<float> &g, const OmniTab &tab, const OmniMlPlan &p,
                     const StreamScreen<32> *ss, bool stats, bool list, hipStream_t stream)
{
    OmniMlArgs a;
    a.ny = p.ny;
    a.nx = p.nx;
    a.segw = p.segw;
    a.xsegs = p.xsegs;
    a.tmax = (int)ceil_div(p.nx, 64) + 2;
    a.x4 = ((((uintptr_t)g.c11 | (uintptr_t)g.c12r | (uintptr_t)g.c12i | (uintptr_t)g.c22) & 15) == 0 &&
            (g.sy & 3) == 0 && (g.st & 3) == 0) ? 1 : 0;
    {
        const char *e4 = getenv("ND_AMD_ML_X4");          // 0: per-element transfers everywhere (diagnostic)
        if (e4 && atoi(e4) == 0) a.x4 = 0;
    }
    a.wt = 1.0 / (double)(p.ml * p.ml);
    a.list = list ? 1 : 0;
    a.spx = a.nstrips = 0;
    a.trace = nullptr;
    a.trace_block = 1000;
#ifdef ND_ML_TRACE
    {
        const char *e = getenv("ND_AMD_ML_TRACE");       // device pointer (hex) of >= 147456 bytes, diagnostic builds
        a.trace = e ? reinterpret_cast<unsigned long long *>(strtoull(e, nullptr, 16)) : nullptr;
        const char *eb = getenv("ND_AMD_ML_TRACE_BLOCK");
        a.trace_block = eb ? atoi(eb) : 1000;
    }
#endif
    const int k = g.k;
#define ND_ML_K(KK)                                                                       \
    do {                                                                                  \
        if (k <= 8)                                                                       \
            return launch_ml_k<KK, 8>(g, tab, a, ss, stats, p.nblocks, stream);           \
        else if (k <= 16)                                                                 \
            return launch_ml_k<KK, 16>(g, tab, a, ss, stats, p.nblocks, stream);          \
        else                                                                              \
            return launch_ml_k<KK, 24>(g, tab, a, ss, stats, p.nblocks, stream);          \
    } while (0)
    if (p.ml == 3)
        ND_ML_K(3);
    else
        ND_ML_K(5);
#undef ND_ML_K
}

}  // namespace nd_amd
