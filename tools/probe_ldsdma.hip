// tools/probe_ldsdma.hip -- where does `buffer_load_dword[x4] ... offset:N lds` read and write?
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probe_ldsdma.hip -o /tmp/probe_ldsdma && /tmp/probe_ldsdma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lf;

template <int OFF, bool X4>
__global__ void k(const float *src, float *dump, int soff)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = -1.f;
    __syncthreads();
    const uint64_t a = (uint64_t)(uintptr_t)src;
    v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));
    r.z = 0x7fffffff;
    r.w = 0x00020000;
    const unsigned base = (unsigned)(uintptr_t)(lf *)lds + 2048;     // LDS byte address 2048
    const int voff = threadIdx.x * (X4 ? 16 : 4);
    if (X4)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:%4 lds\n\ts_waitcnt vmcnt(0)"
                     :: "s"(base), "v"(voff), "s"(r), "s"(soff), "n"(OFF) : "memory");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen offset:%4 lds\n\ts_waitcnt vmcnt(0)"
                     :: "s"(base), "v"(voff), "s"(r), "s"(soff), "n"(OFF) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) dump[i] = lds[i];
}

template <int OFF, bool X4>
static void run(const float *src, float *dump)
{
    hipLaunchKernelGGL((k<OFF, X4>), dim3(1), dim3(64), 16384, 0, src, dump, 0);
    std::vector<float> h(4096);
    hipMemcpy(h.data(), dump, 16384, hipMemcpyDeviceToHost);
    int first = -1, last = -1, n = 0;
    for (int i = 0; i < 4096; ++i)
        if (h[i] >= 0) {
            if (first < 0) first = i;
            last = i;
            ++n;
        }
    printf("%s offset:%4d -> %d floats landed, LDS float index %d..%d (expected start %d), first values %g %g %g %g %g, value at lane 16's slot %g\n",
           X4 ? "x4" : "x1", OFF, n, first, last, 512 + OFF / 4, first >= 0 ? h[first] : -1, first >= 0 ? h[first + 1] : -1,
           first >= 0 ? h[first + 2] : -1, first >= 0 ? h[first + 3] : -1, first >= 0 ? h[first + 4] : -1,
           first >= 0 ? h[first + (X4 ? 64 : 16)] : -1);
}

int main()
{
    float *src, *dump;
    hipMalloc(&src, 1 << 20);
    hipMalloc(&dump, 16384);
    std::vector<float> h(1 << 18);
    for (int i = 0; i < (1 << 18); ++i) h[i] = (float)i;      // value = float index in memory
    hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
    run<0, false>(src, dump);
    run<256, false>(src, dump);
    run<1024, false>(src, dump);
    run<1280, false>(src, dump);
    run<3840, false>(src, dump);
    run<0, true>(src, dump);
    run<1024, true>(src, dump);
    run<2048, true>(src, dump);
    run<3072, true>(src, dump);
    return 0;
}
