// tools/probe_tiles.hip -- what does the memory side give a block that walks a strip of a stack of planes
// in tiles, as the fused multilooking kernel (omnibus_ml.hip) does?  Staging only: per step the block asks
// for ROWS rows x COLS columns of 8 planes by LDS-DMA (16 bytes per lane) into a ring of three slots and
// waits by count for the step two back; nothing is computed.  One block per CU (the LDS ring is sized like
// the kernel's).  Question: is the rate set by the SHAPE of the pieces (bytes that are contiguous in
// memory per row of a plane: 4 COLS), at equal bytes in flight?
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/probe_tiles tools/probe_tiles.hip && gpurun_out/probe_tiles
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(e)                                                                         \
    do {                                                                                 \
        hipError_t _e = (e);                                                             \
        if (_e != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));                      \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(1))) unsigned char glb_u8;
typedef __attribute__((address_space(3))) unsigned char lds_u8;

constexpr int NY = 4096, NX = 4096, NPL = 96;      // 24 dates x 4 variables, planar

// ROWS x COLS floats per plane and step, OUT rows of them are a strip's own (the rest is halo: fetched by
// the neighbouring strip as well).  256 threads; a step's 8 planes x ROWS x COLS / 4 sixteen-byte pieces
// are dealt out to the lanes in order (piece -> plane, row, column quad).
template <int ROWS, int COLS, int OUT>
__global__ void __launch_bounds__(256) probe_tiles(const float *base, float *sink, int xsegs, int segw)
{
    extern __shared__ __align__(16) float ring[];                 // [3][8][ROWS][COLS]
    constexpr int SLOT = 8 * ROWS * COLS;
    constexpr int PIECES = SLOT / 4;                              // 16-byte pieces per step
    constexpr int PER = (PIECES + 255) / 256;                     // per thread
    const int tid = threadIdx.x;
    const int strip = blockIdx.x / xsegs, xseg = blockIdx.x % xsegs;
    const int y0 = strip * OUT, Xs = xseg * segw;
    const int ntiles = segw / COLS;
    // per piece: offset (floats) inside a plane, relative to the tile's first column
    int64_t off[PER];
    int pl[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int pc = tid + 256 * u;
        const int cq = pc % (COLS / 4), row = (pc / (COLS / 4)) % ROWS, p = pc / (COLS / 4 * ROWS);
        int y = y0 - (ROWS - OUT) / 2 + row;
        y = y < 0 ? 0 : (y >= NY ? NY - 1 : y);
        off[u] = (int64_t)y * NX + 4 * cq;
        pl[u] = p;
    }
    const int total = ntiles * 12;
    int issued = 0;
    auto stage = [&](int S) {
        const int tile = S / 12, s = S % 12;
        const int X = Xs + tile * COLS;
        float *slot = ring + (S % 3) * SLOT;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int pc = tid + 256 * u;
            if (PIECES % 256 == 0 || pc < PIECES) {
                const float *src = base + ((int64_t)(8 * s + pl[u])) * NY * NX + off[u] + X;
                // the wave's 64 pieces land at consecutive 16-byte places from the wave's first piece on
                float *dst = slot + 4 * (pc & ~63);
                __builtin_amdgcn_global_load_lds((glb_u8 *)src, (lds_u8 *)dst, 16, 0, 0);
            }
        }
    };
    stage(0);
    if (total > 1) stage(1);
    issued = total > 1 ? 2 : 1;
    for (int S = 0; S < total; ++S) {
        if (issued < total) {
            stage(issued);
            ++issued;
        }
        // step S has landed: at most the steps behind it in flight
        const int behind = issued - 1 - S;
        if (behind >= 2) {
            if (PER == 7)
                asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            else if (PER == 8)
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (PER == 10)
                asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else if (PER == 12)
                asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    if (ring[tid] == 12345.678f) sink[0] = 1.f;
}

template <int ROWS, int COLS, int OUT>
static void run(const float *buf, float *sink, const char *name)
{
    const int strips = (NY + OUT - 1) / OUT;
    int xsegs = 1;
    while (strips * xsegs < 2048 && NX / (xsegs * 2) / COLS >= 4) xsegs *= 2;
    const int segw = NX / xsegs;
    const size_t lds = 3 * (size_t)8 * ROWS * COLS * sizeof(float);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_tiles<ROWS, COLS, OUT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe_tiles<ROWS, COLS, OUT>), dim3(strips * xsegs), dim3(256), lds, 0, buf, sink, xsegs, segw);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double useful = (double)NPL * NY * NX * 4;
    const double moved = useful * ROWS / OUT;
    printf("%-28s rows %2d (own %2d) x cols %3d: piece %4d B, lds %6zu B, blocks %5d: %.3f ms  useful %.2f TB/s  "
           "staged %.2f TB/s\n",
           name, ROWS, OUT, COLS, 4 * COLS, lds, strips * xsegs, best, useful / best / 1e9, moved / best / 1e9);
}

int main()
{
    const size_t bytes = (size_t)NPL * NY * NX * 4;
    float *buf, *sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    run<14, 64, 12>(buf, sink, "kernel's shape (3x3)");
    run<16, 64, 12>(buf, sink, "kernel's shape (5x5)");
    run<8, 128, 6>(buf, sink, "6 rows x 128");
    run<10, 128, 8>(buf, sink, "8 rows x 128");
    run<5, 256, 3>(buf, sink, "3 rows x 256");
    run<6, 256, 4>(buf, sink, "4 rows x 256");
    run<28, 32, 24>(buf, sink, "24 rows x 32");
    run<4, 256, 4>(buf, sink, "no halo, 4 x 256");
    run<16, 64, 16>(buf, sink, "no halo, 16 x 64");
    run<2, 512, 2>(buf, sink, "no halo, 2 x 512");
    return 0;
}
