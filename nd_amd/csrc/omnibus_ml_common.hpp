// nd_amd/csrc/omnibus_ml_common.hpp -- helpers of the fused multilooking kernel (omnibus_ml.hip: a block of
// twelve waves per strip, one barrier per step), shared with the wave-private form that was built and
// measured in round 5 (tools/experiments/omnibus_mlw.hip: every wave for itself, no barrier; same maps,
// 2.69 against 2.04 ms -- its rows above and below a wave's pair double the bytes staged through
// L2 -> LDS, and LDS-DMA ingest tops out near 6 - 7 TB/s chip-wide, tools/probe_tiles.hip).
#pragma once
#include "omnibus_c2_device.hpp"

namespace nd_amd {

typedef __attribute__((address_space(3))) float ml_lds_f32;

struct OmniMlArgs {
    int64_t ny, nx;           // raster
    int segw;                 // columns per segment (multiple of 64)
    int xsegs;                // segments per strip
    int tmax;                 // tile numbers per strip: ceil(nx / 64) + 2
    int x4;                   // rows and planes 16-byte aligned: 16-byte transfers allowed
    double wt;                // 1 / ml^2 (nd/filters.py:297)
    int list;                 // 0: no candidate list (z / P rasters only)
    int spx, nstrips;         // (wave form, tools/experiments: strips per XCD, strips in all)
    unsigned long long *trace;   // ND_ML_TRACE builds: time stamps of one block (tools/exp_ml_trace.py)
    int trace_block;
};

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


}  // namespace nd_amd
