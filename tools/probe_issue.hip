// Issue cost of single vector instructions on gfx950: one wave, 8 independent chains, 256
// instructions between two s_memtime reads.  hipcc --offload-arch=gfx950 -O3 tools/probe_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2_t __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(64) probe(long long *out, float seed)
{
    f2_t a[8];
    double d[8];
    float f[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = (f2_t){seed + i, seed - i};
        d[i] = (double)seed + i;
        f[i] = seed * (i + 1);
    }
    __shared__ float lds[1024];
    lds[threadIdx.x] = seed;
    lds[threadIdx.x + 64] = seed;
    __syncthreads();
    const f2_t c = (f2_t){seed, seed};
    const double dc = (double)seed;
    const unsigned la = threadIdx.x * 8;
    long long t0 = __builtin_readcyclecounter();
    asm volatile("s_nop 0" ::: "memory");
    t0 = clock64();
    for (int it = 0; it < 32; ++it) {
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(seed));
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(seed));
#define ADDDPP(i) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]) : "v"(seed));
#define MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(seed));
#define ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
#define MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(dc));
#define CVT64(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
#define CVT32(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
#define DSR64(i) asm volatile("ds_read_b64 %0, %1" : "=v"(a[i]) : "v"(la));
#define DSR32(i) asm volatile("ds_read_b32 %0, %1" : "=v"(f[i]) : "v"(la));
#define DSR2(i) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(a[i]) : "v"(la));
        if (OP == 0) { REP8(PKFMA) }
        if (OP == 1) { REP8(PKADD) }
        if (OP == 2) { REP8(PKMUL) }
        if (OP == 3) { REP8(FMA) }
        if (OP == 4) { REP8(ADD) }
        if (OP == 5) { REP8(ADDDPP) }
        if (OP == 6) { REP8(MOVDPP) }
        if (OP == 7) { REP8(EXP) }
        if (OP == 8) { REP8(MAXF) }
        if (OP == 9) { REP8(ADD64) }
        if (OP == 10) { REP8(MUL64) }
        if (OP == 11) { REP8(FMA64) }
        if (OP == 12) { REP8(CVT64) }
        if (OP == 13) { REP8(CVT32) }
        if (OP == 14) { REP8(DSR64) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 15) { REP8(DSR32) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 16) { REP8(DSR2) asm volatile("s_waitcnt lgkmcnt(0)"); }
    }
    asm volatile("s_nop 0" ::: "memory");
    const long long t1 = clock64();
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + (float)d[i] + f[i];
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (acc == 12345.678f) out[1] = 1;
}

template <int OP>
static void run(const char *name, long long *dout, int waves)
{
    // `waves` waves per SIMD: blocks of 64 threads on one CU cannot be forced; use one block per
    // wave and many blocks, report the slowest block's cycles / 256
    hipLaunchKernelGGL((probe<OP>), dim3(1), dim3(64), 0, 0, dout, 1.0f);
    hipDeviceSynchronize();
    long long best = 1ll << 60;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL((probe<OP>), dim3(1), dim3(64), 0, 0, dout, 1.0f);
        long long h[2];
        hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
        if (h[0] < best) best = h[0];
    }
    printf("%-22s %6.2f cycles per instruction (one wave, independent)\n", name, best / 256.0);
}

int main()
{
    long long *dout;
    hipMalloc(&dout, 4096);
    hipMemset(dout, 0, 4096);
    run<0>("v_pk_fma_f32", dout, 1);
    run<1>("v_pk_add_f32", dout, 1);
    run<2>("v_pk_mul_f32", dout, 1);
    run<3>("v_fma_f32", dout, 1);
    run<4>("v_add_f32", dout, 1);
    run<5>("v_add_f32_dpp wave_shr", dout, 1);
    run<6>("v_mov_b32_dpp wave_shr", dout, 1);
    run<7>("v_exp_f32", dout, 1);
    run<8>("v_max_f32", dout, 1);
    run<9>("v_add_f64", dout, 1);
    run<10>("v_mul_f64", dout, 1);
    run<11>("v_fma_f64", dout, 1);
    run<12>("v_cvt_f64_f32", dout, 1);
    run<13>("v_cvt_f32_f64", dout, 1);
    run<14>("ds_read_b64", dout, 1);
    run<15>("ds_read_b32", dout, 1);
    run<16>("ds_read2_b32", dout, 1);
    return 0;
}
