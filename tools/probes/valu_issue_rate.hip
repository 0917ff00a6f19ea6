// Probe: issue cost (shader clocks per wave64 instruction, from s_memtime) of the instruction
// forms the step kernels are made of, at 1 / 2 / 4 waves per SIMD.  16 independent
// destinations, 128 instructions per loop iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(256) k(double* out, long long* clk, int iters, double sval)
{
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = 1.0 + i + threadIdx.x * 1e-3;
    double b = 1.0000001 + threadIdx.x * 1e-9, c = 0.25;
    unsigned u[16];
    for (int i = 0; i < 16; ++i) u[i] = threadIdx.x * 2654435761u + i;
    double s = sval;  // wave-uniform -> SGPR pair
    unsigned su = (unsigned)__builtin_amdgcn_readfirstlane((int)blockIdx.x);
    unsigned long long sm = __builtin_amdgcn_ballot_w64((threadIdx.x & 1) != 0);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "s"(s), "v"(c));
                if (KIND == 2) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(c));
                if (KIND == 3) asm volatile("v_cmp_le_f64 vcc, %1, %0" : "+v"(a[i]) : "s"(s) : "vcc");
                if (KIND == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (KIND == 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(a[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]) : "vcc");
                if (KIND == 7) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
                if (KIND == 8) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "s"(sm));
                if (KIND == 9) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (KIND == 10) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "s"(s));
                if (KIND == 11) asm volatile("v_fma_f64 %0, %2, %3, %0\n\ts_and_b64 %1, %1, %1" : "+v"(a[i]), "+s"(sm) : "s"(s), "v"(c) : "scc");
                if (KIND == 12) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (KIND == 13) asm volatile("v_fma_f64 %0, %3, %4, %0\n\ts_and_b64 %1, %1, %1\n\ts_add_u32 %2, %2, 1" : "+v"(a[i]), "+s"(sm), "+s"(su) : "s"(s), "v"(c) : "scc");
            }
    }
    const long long t1 = clock64();
    double acc = 0;
    for (int i = 0; i < 16; ++i) acc += a[i] + u[i];
    asm volatile("; sink" :: "s"(sm), "s"(su));
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, double* out, long long* clk)
{
    const int iters = 20000;
    printf("%-34s", name);
    for (int wg : {256, 512, 1024}) {
        k<KIND><<<wg, 256>>>(out, clk, iters, 1.0000001);
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); k<KIND><<<wg, 256>>>(out, clk, iters, 1.0000001); hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long h[4096 * 4];
        hipMemcpy(h, clk, sizeof(long long) * wg * 4, hipMemcpyDeviceToHost);
        double m = 0;
        for (int i = 0; i < wg * 4; ++i) m += (double)h[i];
        m /= wg * 4;
        const double per_wave = m / ((double)iters * 128);           // clocks per instr of a wave
        const double per_simd = per_wave / (wg / 256);               // with n waves sharing a SIMD
        printf("  %dw/SIMD: %6.2f clk/instr/SIMD (%.3f ms)", wg / 256, per_simd, ms);
    }
    printf("\n");
    fflush(stdout);
}

int main()
{
    double* out; long long* clk;
    hipMalloc(&out, sizeof(double) * 256 * 4096);
    hipMalloc(&clk, sizeof(long long) * 4 * 4096);
    run<0>("v_fma_f64 v,v,v,v", out, clk);
    run<1>("v_fma_f64 v,s,v,v", out, clk);
    run<2>("v_fmac_f64 v,s,v", out, clk);
    run<3>("v_cmp_le_f64 vcc,s,v", out, clk);
    run<4>("v_add_f64 v,v,v", out, clk);
    run<10>("v_mul_f64 v,v,s", out, clk);
    run<5>("v_mul_lo_u32", out, clk);
    run<12>("v_mul_hi_u32", out, clk);
    run<6>("v_mad_u64_u32", out, clk);
    run<7>("v_rcp_f64", out, clk);
    run<8>("v_cndmask_b32 (sgpr mask)", out, clk);
    run<9>("v_xor_b32", out, clk);
    run<11>("v_fma_f64 + s_and_b64", out, clk);
    run<13>("v_fma_f64 + s_and_b64 + s_add_u32", out, clk);
    return 0;
}
