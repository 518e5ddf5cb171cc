// tools/probe_fetch.hip -- calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns
// the nd_amd kernels use.  MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide
// coalesced streaming read (16 B per lane) ... other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern".  Every kernel below reads a KNOWN number of bytes of a
// 2 GiB buffer (well past the 256 MiB Infinity Cache) exactly once; tools/summarize_fetch_probe.py
// divides FETCH_SIZE by it.
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/probe_fetch tools/probe_fetch.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -o p --output-format csv -- gpurun_out/probe_fetch
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

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32;

// 1. 16 B per lane, coalesced (the planar pass A's loads)
__global__ void __launch_bounds__(256) probe_x4(const f4 *p, float *sink, size_t n16)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (; i < n16; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
// 1b. the same, non-temporal
__global__ void __launch_bounds__(256) probe_x4_nt(const f4 *p, float *sink, size_t n16)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (; i < n16; i += (size_t)gridDim.x * 256) acc += __builtin_nontemporal_load(p + i);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
// 2. 4 B per lane, coalesced (256 B per wave instruction)
__global__ void __launch_bounds__(256) probe_x1(const float *p, float *sink, size_t n4)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0;
    for (; i < n4; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 12345.678f) sink[0] = acc;
}
// 3. LDS-DMA, 4 B per lane (nlmeans_window_stream3_kernel, the ml kernel's edge form)
__global__ void __launch_bounds__(256) probe_lds_x1(const float *p, float *sink, size_t n4)
{
    __shared__ float img[256 * 4];
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 0x7fffffff, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // 2 GiB of offsets: the scalar offset carries the block's base (< 2^31)
    for (size_t i = (size_t)blockIdx.x * 256; i < n4; i += (size_t)gridDim.x * 256) {
        const int so = __builtin_amdgcn_readfirstlane((int)((i + wave * 64) * 4));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_f32 *)(img + wave * 64), 4, lane * 4, so, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (img[threadIdx.x] == 12345.678f) sink[0] = 1.f;
}
// 4. LDS-DMA, 16 B per lane (the ml kernel, the pixel-major pass A)
__global__ void __launch_bounds__(256) probe_lds_x4(const float *p, float *sink, size_t n4)
{
    __shared__ __align__(16) float img[256 * 4];
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 0x7fffffff, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (size_t i = (size_t)blockIdx.x * 1024; i < n4; i += (size_t)gridDim.x * 1024) {
        const int so = __builtin_amdgcn_readfirstlane((int)((i + wave * 256) * 4));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_f32 *)(img + wave * 256), 16, lane * 16, so, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (img[threadIdx.x] == 12345.678f) sink[0] = 1.f;
}
// 5. isolated 4-byte reads: one per `pitch` bytes (the C3 pass B's gather: 4 KiB apart = one per page
//    of a plane row; 256 B apart; 128 B apart).  bytes "used" = 4 per read.
__global__ void __launch_bounds__(256) probe_gather(const float *p, float *sink, size_t nreads, size_t pitch4,
                                                    unsigned mul)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0;
    for (; i < nreads; i += (size_t)gridDim.x * 256) {
        // a permutation of the read slots so that neighbouring lanes are far apart
        const size_t j = (i * (size_t)mul) % nreads;
        acc += p[j * pitch4];
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// 6. 16-byte pieces at a 96-byte pitch per lane, six instructions per 6 KiB span of a wave (the
//    DIRECT form of omnibus_c2_pm_dma_kernel: a lane reads its own 24-date series)
template <bool NT>
__global__ void __launch_bounds__(64) probe_pitch96(const float *p, float *sink, size_t nspans)
{
    const int lane = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (size_t s = blockIdx.x; s < nspans; s += gridDim.x) {
        const f4 *q = reinterpret_cast<const f4 *>(p + s * 1536 + lane * 24);
        f4 v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) v[u] = NT ? __builtin_nontemporal_load(q + u) : q[u];
#pragma unroll
        for (int u = 0; u < 6; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

int main()
{
    const size_t bytes = (size_t)2040 << 20;           // just under 2 GiB: 32-bit buffer offsets
    float *buf, *sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    const size_t n4 = bytes / 4, n16 = bytes / 16;
    const int grid = 256 * 8;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe_x4, dim3(grid), dim3(256), 0, 0, (const f4 *)buf, sink, n16);
        hipLaunchKernelGGL(probe_x4_nt, dim3(grid), dim3(256), 0, 0, (const f4 *)buf, sink, n16);
        hipLaunchKernelGGL(probe_x1, dim3(grid), dim3(256), 0, 0, buf, sink, n4);
        hipLaunchKernelGGL(probe_lds_x1, dim3(grid), dim3(256), 0, 0, buf, sink, n4);
        hipLaunchKernelGGL(probe_lds_x4, dim3(grid), dim3(256), 0, 0, buf, sink, n4);
        // gathers: 4 Mi reads each (16 MiB used)
        hipLaunchKernelGGL(probe_gather, dim3(grid), dim3(256), 0, 0, buf, sink, (size_t)(bytes / 4096), (size_t)1024, 7919u);
        hipLaunchKernelGGL(probe_gather, dim3(grid), dim3(256), 0, 0, buf, sink, (size_t)(bytes / 256), (size_t)64, 7919u);
        hipLaunchKernelGGL(probe_gather, dim3(grid), dim3(256), 0, 0, buf, sink, (size_t)(bytes / 128), (size_t)32, 7919u);
        hipLaunchKernelGGL(probe_gather, dim3(grid), dim3(256), 0, 0, buf, sink, (size_t)(bytes / 64), (size_t)16, 7919u);
        hipLaunchKernelGGL((probe_pitch96<true>), dim3(grid * 4), dim3(64), 0, 0, buf, sink, bytes / 6144);
        hipLaunchKernelGGL((probe_pitch96<false>), dim3(grid * 4), dim3(64), 0, 0, buf, sink, bytes / 6144);
        CHECK(hipDeviceSynchronize());
    }
    // the byte counts the summary divides by, in launch order
    printf("bytes %zu\n", bytes);
    printf("probe_x4 %zu\nprobe_x4_nt %zu\nprobe_x1 %zu\nprobe_lds_x1 %zu\nprobe_lds_x4 %zu\n", n16 * 16, n16 * 16, n4 * 4,
           n4 * 4, n4 / 1024 * 1024 * 4);
    printf("probe_gather_4096 %zu reads\nprobe_gather_256 %zu reads\nprobe_gather_128 %zu reads\nprobe_gather_64 %zu reads\n",
           bytes / 4096, bytes / 256, bytes / 128, bytes / 64);
    printf("probe_pitch96 %zu\n", bytes / 6144 * 6144);
    return 0;
}
