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
//   * A block of 768 threads owns a strip of 12 rows and WALKS it along x in tiles of 64 columns.
//     Wave w owns tile row w, lane j the pixel in column j of the tile.
//   * A tile is consumed in steps of 8 planes (2 dates x 4 variables).  The planes of a step are
//     staged by LDS-DMA (buffer_load_dword ... lds: memory -> LDS, no registers): 12 + 2h rows of 64
//     NEW columns each -- every transfer is one aligned, fully coalesced 256-byte row piece, and no
//     column is ever fetched twice: the 2h columns a tile shares with its left neighbour are carried
//     over inside LDS (a tile's outputs are the 64 columns that end h columns before its last new
//     column).  Only the h rows above and below a strip are fetched by two blocks.
//   * Three staging slots: while step S is computed the transfers of S + 1 and S + 2 are in flight.
//   * Compute role: a thread forms the window sums of a patch of 4 columns x 2 rows of ONE plane,
//     walking the patch's 2 + 2h input rows once (two 16-byte LDS reads per row, bank-conflict free
//     by construction of the lane -> (plane, patch) map); every element is converted to double and
//     multiplied by 1 / w^2 once per patch, every output receives its w^2 terms in scipy's order.
//     The 8 results (float) go to a result area in LDS; after a barrier every thread picks up the 8
//     values of ITS pixel: v[2 s + ...][0..3].
//   * After the last step of a tile the series is complete and the kernel continues like the plain
//     forms: fold + screen + list + dump + zero-fill (sparse regime) or dense_chain (fused search).
//
// Traffic: 4 k planes x (12 + 2h) / 12 rows, nothing else.  Work: w^2 dependent double additions per
// value (scipy's order leaves no sharing between neighbouring windows) -- the kernel is bound by
// vector issue, not by memory, from 5 x 5 on.
#include "omnibus_c2_device.hpp"

namespace nd_amd {

typedef __attribute__((address_space(3))) float ml_lds_f32;

template <int K>
struct MlGeom {
    static constexpr int HALO = K / 2;
    static constexpr int W = 64, HT = kMlTileRows, NT = 64 * HT, NWAVE = HT, G = 8;
    static constexpr int ROWS = HT + 2 * HALO;
    static constexpr int NRD = (4 + 2 * HALO + 3) / 4;            // 16-byte reads per staged row and patch
    static constexpr int PITCH = 60 + 4 * NRD;                    // floats per staged row
    static constexpr int PSZ = (ROWS * PITCH + 63) / 64 * 64;     // floats per staged plane (64-dword multiple)
    static constexpr int SLOT = G * PSZ;
    static constexpr int RES = G * NT;
    static constexpr int NCAR = G * ROWS * 2 * HALO;              // carried elements per step
    static constexpr int NSLOT = 3;
};

struct OmniMlArgs {
    int64_t ny, nx;           // raster
    int segw;                 // output columns per segment (multiple of 64)
    int xsegs;                // segments per strip
    int tmax;                 // upper bound of the tiles of a segment (segw / 64 + 1): numbering of the waves
    double wt;                // 1 / ml^2 (nd/filters.py:297)
    int list;                 // 0: no candidate list (z / P rasters only)
};

// all LDS writes of this wave done, then the workgroup barrier (no wait for the LDS-DMA transfers in
// flight: those are waited for explicitly, by count)
__device__ __forceinline__ void ml_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int ml_reflect(int cc, const int len)     // scipy 'reflect': d c b a | a b c d | d c b a
{
    if (cc >= 0 && cc < len) return cc;
    const int sz2 = 2 * len;
    // (no division: the columns asked for lie within a tile of the raster, the loops run at most a
    //  few times, and only for rasters narrower than a tile)
    while (cc < -len) cc += sz2;
    while (cc >= sz2) cc -= sz2;
    if (cc < 0) return -cc - 1;
    if (cc >= len) return sz2 - cc - 1;
    return cc;
}

// LDS-DMA issued from inline assembly: 64 x 4 bytes, memory -> LDS at `lds_addr` + 4 * lane.  The
// compiler's wait-count pass does not see these transfers, which is the point: it would otherwise
// put `s_waitcnt vmcnt(0)` in front of EVERY LDS read that follows a transfer it cannot prove
// disjoint (it cannot, for dynamic LDS) -- i.e. wait for the planes of two steps ahead before the
// current step's first read.  The kernel waits for its transfers itself, by count.  (Wait counts
// the compiler emits for its own loads stay safe: transfers it does not know about only make
// `vmcnt(n)` wait for more than it had to.)
typedef int ml_v4i __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void ml_dma_row(const ml_v4i rsrc, const unsigned lds_base, const int voff, const int soff)
{
    // (one wait state between a write of M0 and the LDS-DMA that reads it: the s_nop)
    asm volatile("s_add_u32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :
                 : "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF)
                 : "memory", "scc");   // (M0 is reserved: the compiler keeps nothing in it on gfx9)
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

// The rows of the change map of `n` consecutive pixels, lanes lo .. lo + n - 1 of the wave, from the
// lanes' masks: through a wave-private LDS image (k / 4 words per lane), then 16-byte pieces of
// consecutive lanes wherever the destination allows (the span of a shifted tile starts on a 4-byte
// boundary only).  k a multiple of 4.
template <typename MT>
__device__ __forceinline__ void ml_store_change_rows(uint8_t *ob, uint32_t *img, const int k, const MT &mask,
                                                     const int lane, const int lo, const int n)
{
    const int kq = k >> 2;
    for (int q = 0; q < kq; ++q)
        img[lane * kq + q] = (mask_nibble(mask, q) * 0x00204081u) & 0x01010101u;     // bit i -> byte i
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint32_t *src = img + lo * kq;
    uint32_t *dst = reinterpret_cast<uint32_t *>(ob);
    const int nw = n * kq;                                       // words to write
    int head = (int)(((16 - ((uintptr_t)ob & 15)) & 15) >> 2);
    if (head > nw) head = nw;
    if (lane < head) dst[lane] = src[lane];
    const int nvec = (nw - head) >> 2;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    for (int c = lane; c < nvec; c += 64) {
        const uint32_t *s = src + head + 4 * c;
        const u4 q = {s[0], s[1], s[2], s[3]};
        __builtin_nontemporal_store(q, reinterpret_cast<u4 *>(dst + head) + c);
    }
    const int tail0 = head + 4 * nvec;
    if (tail0 + lane < nw) dst[tail0 + lane] = src[tail0 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int K, int KMAX, bool STATS, bool CHAIN>
__global__ void __launch_bounds__(64 * kMlTileRows)
omnibus_c2_ml_kernel(const OmniGlobalArgs<float> g, const OmniTab tab, const OmniMlArgs ml,
                     const StreamScreen<32> ss)
{
    typedef MlGeom<K> M;
    constexpr int HALO = M::HALO, ROWS = M::ROWS, PITCH = M::PITCH, PSZ = M::PSZ, NT = M::NT;
    constexpr int NSTEP = KMAX / 2;                 // steps per tile (2 dates x 4 variables each)
    extern __shared__ __align__(16) unsigned char nd_smem_ml[];
    float *slots = reinterpret_cast<float *>(nd_smem_ml);                 // [3][SLOT]
    float *res = slots + M::NSLOT * M::SLOT;                              // [wave][8][64]
    float *carry = res + M::RES;                                          // [NSTEP][NCAR]
    StreamEntry *tab_lds = reinterpret_cast<StreamEntry *>(carry + NSTEP * M::NCAR);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = g.k;
    const int nx = (int)ml.nx, ny = (int)ml.ny;
    const int64_t b = blockIdx.x;
    const int strip = (int)(b / ml.xsegs), xseg = (int)(b - (int64_t)strip * ml.xsegs);
    const int y0 = strip * M::HT;
    const int Xs = xseg * ml.segw;
    const int Xe = (Xs + ml.segw < nx) ? Xs + ml.segw : nx;      // outputs of this block: columns [Xs, Xe)
    const int ntiles = (Xe - Xs + HALO + 63) / 64;
    const int nstep_k = (k + 1) >> 1;               // steps that hold dates of the series

    if (CHAIN && tid <= 32) tab_lds[tid] = ss.e[tid];
    if (g.write_tab && b == 0) {
        for (int j = tid; j <= k; j += NT) g.tab_dev[j] = tab.e[j];
    }

    // ---- staging: wave `wave` moves rows wave and wave + 12 (if staged) of every plane of a step ----
    const float *vp[4] = {g.c11, g.c12r, g.c12i, g.c22};
    ml_v4i rs[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) rs[v] = ml_make_rsrc(vp[v]);
    const unsigned lds0 = (unsigned)(uintptr_t)(ml_lds_f32 *)slots;       // LDS byte address of the slots
    const bool two_rows = wave + M::NWAVE < ROWS;                // wave-uniform
    const int roffA = __builtin_amdgcn_readfirstlane(ml_reflect(y0 - HALO + wave, ny) * (int)g.sy * 4);
    const int roffB = __builtin_amdgcn_readfirstlane(
        ml_reflect(y0 - HALO + (two_rows ? wave + M::NWAVE : wave), ny) * (int)g.sy * 4);
    const int sstep = (int)g.st * 4;                             // bytes between dates (host: k * st * 4 < 2^31)

    auto stage = [&](const int s, const int Xi, const int slot) {
        // planes 8 s .. 8 s + 7 = dates 2 s, 2 s + 1 (a date beyond the series repeats the last one)
        const int xm = ml_reflect(Xi + lane, nx);
        const int voff = xm * 4;
        const int t0 = 2 * s < k ? 2 * s : k - 1, t1 = 2 * s + 1 < k ? 2 * s + 1 : k - 1;
        const int so0 = t0 * sstep, so1 = t1 * sstep;
        const unsigned sb = lds0 + 4u * (unsigned)(slot * M::SLOT + wave * PITCH + 2 * HALO);
#pragma unroll
        for (int pl = 0; pl < 8; ++pl) {
            if (pl == 0) ml_dma_row<0 * 4 * PSZ>(rs[0], sb, voff, so0 + roffA);
            if (pl == 1) ml_dma_row<1 * 4 * PSZ>(rs[1], sb, voff, so0 + roffA);
            if (pl == 2) ml_dma_row<2 * 4 * PSZ>(rs[2], sb, voff, so0 + roffA);
            if (pl == 3) ml_dma_row<3 * 4 * PSZ>(rs[3], sb, voff, so0 + roffA);
            if (pl == 4) ml_dma_row<4 * 4 * PSZ>(rs[0], sb, voff, so1 + roffA);
            if (pl == 5) ml_dma_row<5 * 4 * PSZ>(rs[1], sb, voff, so1 + roffA);
            if (pl == 6) ml_dma_row<6 * 4 * PSZ>(rs[2], sb, voff, so1 + roffA);
            if (pl == 7) ml_dma_row<7 * 4 * PSZ>(rs[3], sb, voff, so1 + roffA);
        }
        if (two_rows) {
            const unsigned sb2 = sb + 4u * (unsigned)(M::NWAVE * PITCH);
            ml_dma_row<0 * 4 * PSZ>(rs[0], sb2, voff, so0 + roffB);
            ml_dma_row<1 * 4 * PSZ>(rs[1], sb2, voff, so0 + roffB);
            ml_dma_row<2 * 4 * PSZ>(rs[2], sb2, voff, so0 + roffB);
            ml_dma_row<3 * 4 * PSZ>(rs[3], sb2, voff, so0 + roffB);
            ml_dma_row<4 * 4 * PSZ>(rs[0], sb2, voff, so1 + roffB);
            ml_dma_row<5 * 4 * PSZ>(rs[1], sb2, voff, so1 + roffB);
            ml_dma_row<6 * 4 * PSZ>(rs[2], sb2, voff, so1 + roffB);
            ml_dma_row<7 * 4 * PSZ>(rs[3], sb2, voff, so1 + roffB);
        }
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
    stage(0, Xs, 0);
    if (nstep_k > 1 || ntiles > 1) stage(nstep_k > 1 ? 1 : 0, nstep_k > 1 ? Xs : Xs + 64, 1);
    __syncthreads();
    // carry-in of the very first step
    if (tid < M::NCAR) {
        const int c = tid % (2 * HALO), r = (tid / (2 * HALO)) % ROWS, pl = tid / (2 * HALO * ROWS);
        slots[pl * PSZ + r * PITCH + c] = carry[tid];
    }

    // compute role
    const int py = wave % (M::HT / 2), quad = wave / (M::HT / 2);
    const int cpl = quad * 4 + (lane >> 4), cpx = lane & 15;
    const int rd_off = cpl * PSZ + (2 * py) * PITCH + 4 * cpx;                          // floats into a slot
    const int wr_off = (2 * py) * (8 * 64) + cpl * 64 + 4 * cpx;                        // floats into res
    const double wt = ml.wt;

    const int total_steps = ntiles * nstep_k;
    int S = 0;                       // global step counter of the block
    int slot_i = 0;                  // slot of step S
    float v[KMAX][4];

    for (int i = 0; i < ntiles; ++i) {
        const int Xi = Xs + 64 * i;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s < nstep_k) {
                float *cur = slots + slot_i * M::SLOT;
                const int slot_n = slot_i + 1 == M::NSLOT ? 0 : slot_i + 1;
                const int slot_nn = slot_n + 1 == M::NSLOT ? 0 : slot_n + 1;
                // ---- the transfers of this step have landed (this wave's), then everybody's ----
                if (S + 1 < total_steps) {
                    if (two_rows)
                        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                ml_barrier();
                // ---- two steps ahead ----
                if (S + 2 < total_steps) {
                    int s2 = s + 2, X2 = Xi;
                    while (s2 >= nstep_k) {
                        s2 -= nstep_k;
                        X2 += 64;
                    }
                    stage(s2, X2, slot_nn);
                }
                // ---- carried columns: this step's last 2h new columns for the next tile, and the
                //      next step's first 2h columns from what the previous tile left ----
                if (tid < M::NCAR) {
                    const int c = tid % (2 * HALO), r = (tid / (2 * HALO)) % ROWS, pl = tid / (2 * HALO * ROWS);
                    const int sn = s + 1 < nstep_k ? s + 1 : 0;
                    const float out_this = cur[pl * PSZ + r * PITCH + 64 + c];
                    // (a one-step tile: the next step is the next tile's, its carry is this step's)
                    const float in_next = sn == s ? out_this : carry[sn * M::NCAR + tid];
                    // (the next step of the LAST plane group belongs to the next tile: its carry is
                    //  what this tile's step 0 stored -- written a whole tile ago)
                    if (S + 1 < total_steps) slots[slot_n * M::SLOT + pl * PSZ + r * PITCH + c] = in_next;
                    carry[s * M::NCAR + tid] = out_this;
                }
                // ---- window sums of this thread's patch ----
                {
                    double acc[2][4];
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) acc[oy][ii] = 0.0;
                    const float *P = cur + rd_off;
#pragma unroll
                    for (int r = 0; r < 2 + 2 * HALO; ++r) {
                        float wv[4 * M::NRD];
                        const float4 *rp = reinterpret_cast<const float4 *>(P + r * PITCH);
#pragma unroll
                        for (int cc = 0; cc < M::NRD; ++cc) {
                            const float4 q = rp[cc];
                            wv[4 * cc + 0] = q.x;
                            wv[4 * cc + 1] = q.y;
                            wv[4 * cc + 2] = q.z;
                            wv[4 * cc + 3] = q.w;
                        }
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
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy) {
                        const float4 o = make_float4((float)acc[oy][0], (float)acc[oy][1], (float)acc[oy][2],
                                                     (float)acc[oy][3]);
                        *reinterpret_cast<float4 *>(res + wr_off + oy * (8 * 64)) = o;
                    }
                }
                ml_barrier();
                // ---- this thread's pixel: its 8 values of the step ----
                {
                    const float *R = res + wave * (8 * 64) + lane;
#pragma unroll
                    for (int pl = 0; pl < 8; ++pl) {
                        const int t = 2 * s + (pl >> 2);
                        if (t < KMAX) v[t][pl & 3] = R[pl * 64];
                    }
                }
                S += 1;
                slot_i = slot_n;
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
        const int y = y0 + wave;
        const int x = Xi - HALO + lane;
        const bool in = (y < ny) && (x >= Xs) && (x < Xe);
        // valid span of the wave: lanes lo .. lo + wnp - 1
        int xlo = Xi - HALO, xhi = Xi - HALO + 64;
        xlo = xlo < Xs ? Xs : xlo;
        xhi = xhi > Xe ? Xe : xhi;
        const int wnp = (y < ny && xhi > xlo) ? xhi - xlo : 0;
        const int lo = xlo - (Xi - HALO);
        uint8_t *wob = g.change + ((int64_t)y * nx + xlo) * (int64_t)k;
        const int64_t wid = (b * ml.tmax + i) * (int64_t)M::NWAVE + wave;
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
                    // (the wave's rows of the result area are its own until the next step's barrier)
                    ml_store_change_rows(wob, reinterpret_cast<uint32_t *>(res + wave * (8 * 64)), k, mask,
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
    if (((int64_t)k * st + ny * sy + nx) * 4 >= 0x7fffffffLL) return false;
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
    const int64_t waves = p->nblocks * (p->segw / 64 + 1) * kMlTileRows;
    p->seg = (uint32_t)(ceil_div(waves, (int64_t)kShards) * 64 + 64);
    return true;
}

template <int K, int KMAX>
static void launch_ml_k(const OmniGlobalArgs<float> &g, const OmniTab &tab, const OmniMlArgs &a,
                        const StreamScreen<32> *ss, bool stats, int64_t nblocks, hipStream_t stream)
{
    typedef MlGeom<K> M;
    const size_t lds = ((size_t)M::NSLOT * M::SLOT + M::RES + (size_t)(KMAX / 2) * M::NCAR) * sizeof(float) +
                       33 * sizeof(StreamEntry);
    const dim3 grid((unsigned)nblocks), block(M::NT);
    StreamScreen<32> none;
    if (!ss) memset(&none, 0, sizeof(none));
#define ND_ML_LAUNCH(STATS_, CHAIN_)                                                                          \
    do {                                                                                                      \
        static const hipError_t attr_ = hipFuncSetAttribute(                                                  \
            reinterpret_cast<const void *>(&omnibus_c2_ml_kernel<K, KMAX, STATS_, CHAIN_>),                   \
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                            \
        (void)attr_;                                                                                          \
        hipLaunchKernelGGL((omnibus_c2_ml_kernel<K, KMAX, STATS_, CHAIN_>), grid, block, lds, stream, g, tab, a, \
                           ss ? *ss : none);                                                                  \
    } while (0)
    if (ss)
        ND_ML_LAUNCH(false, true);
    else if (stats)
        ND_ML_LAUNCH(true, false);
    else
        ND_ML_LAUNCH(false, false);
#undef ND_ML_LAUNCH
}

void launch_ml_pass_a(const OmniGlobalArgs<float> &g, const OmniTab &tab, const OmniMlPlan &p,
                      const StreamScreen<32> *ss, bool stats, bool list, hipStream_t stream)
{
    OmniMlArgs a;
    a.ny = p.ny;
    a.nx = p.nx;
    a.segw = p.segw;
    a.xsegs = p.xsegs;
    a.tmax = p.segw / 64 + 1;
    a.wt = 1.0 / (double)(p.ml * p.ml);
    a.list = list ? 1 : 0;
    const int k = g.k;
#define ND_ML_K(KK)                                                                       \
    do {                                                                                  \
        if (k <= 8)                                                                       \
            launch_ml_k<KK, 8>(g, tab, a, ss, stats, p.nblocks, stream);                  \
        else if (k <= 16)                                                                 \
            launch_ml_k<KK, 16>(g, tab, a, ss, stats, p.nblocks, stream);                 \
        else                                                                              \
            launch_ml_k<KK, 24>(g, tab, a, ss, stats, p.nblocks, stream);                 \
    } while (0)
    if (p.ml == 3)
        ND_ML_K(3);
    else
        ND_ML_K(5);
#undef ND_ML_K
}

}  // namespace nd_amd
